/*
 * mpc_oracle.c -- plain-C float64 CPU oracle for the MPC hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call this file; the product
 * (neo_mpc_planner2_amd/, libneo_mpc.so) never does.
 *
 * Part 1 restates the REFERENCE's arithmetic (py: = neo_mpc_planner2/
 * mpc_optimization_server.py of neobotix/neo_mpc_planner2): yaw extraction, objective,
 * disc constraint, collision check and the optimizer() wrapper.  It is pinned against
 * outputs of the reference itself through tests/golden/ (oracle/gen_golden.py).
 * The costmap lookup and SciPy's SLSQP are NOT in the reference (un-vendored
 * dependencies): the costmap contract here is this build's own ("parity unpinned"
 * at that boundary), SLSQP is exercised through oracle/mpc_oracle.py + SciPy.
 *
 * Part 2 is a scalar CPU mirror of the build's batched solver (the algorithm of
 * neo_mpc_planner2_amd/csrc/neo_mpc_kernels.hip: L-BFGS / proximal-gradient
 * projected arc search with 64 candidates per iteration), written independently of
 * the HIP source so that GPU results can be checked against it on the same inputs.
 *
 * POD layouts come from include/neo_mpc.h (declarations only).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/neo_mpc.h"
/* the rule book of the build's search (constants + derived tolerances): shared with the product so that the mirror
 * (part 2) cannot drift from the device code; parts 1 and 3 (the restatement of the REFERENCE) use nothing of it */
#include "../neo_mpc_planner2_amd/csrc/solver_rules.h"

#define ORC_LANES 64
#define ORC_STALL_ITERATIONS NEO_RULE_STALL_ITERATIONS
#define ORC_MAXN NEO_MPC_MAX_CONTROL_STEPS
#define ORC_MAXV (3 * ORC_MAXN)

typedef struct orc_map {
  const uint8_t* cells;
  int32_t size_x, size_y;
  double resolution, origin_x, origin_y;
} orc_map;

/* ------------------------------------------------------------------ costmap contract */
/* nav2 Costmap2DPublisher translation: raw u8 -> occupancy [-1, 100]. */
static int orc_occupancy(int raw) {
  if (raw == 0) return 0;
  if (raw == 253) return 99;
  if (raw == 254) return 100;
  if (raw == 255) return -1;
  return 1 + (97 * (raw - 1)) / 251;
}

static void orc_world_to_map(const orc_map* m, double wx, double wy, int64_t* mx, int64_t* my) {
  *mx = (int64_t)floor((wx - m->origin_x) / m->resolution);
  *my = (int64_t)floor((wy - m->origin_y) / m->resolution);
}

static double orc_cost(const orc_map* m, int64_t mx, int64_t my) {
  if (mx < 0 || my < 0 || mx >= m->size_x || my >= m->size_y) return 1.0;
  return (double)orc_occupancy(m->cells[my * (int64_t)m->size_x + mx]) / 100.0;
}

/* max cell cost along the closed polygon outline (Bresenham, end points inclusive). */
double orc_footprint_cost(const orc_map* m, const double* pts, int npts) {
  if (npts <= 0) return 0.0;
  double worst = -1.0;
  for (int i = 0; i < npts; ++i) {
    int64_t x0, y0, x1, y1;
    int j = (i + 1) % npts;
    orc_world_to_map(m, pts[2 * i], pts[2 * i + 1], &x0, &y0);
    orc_world_to_map(m, pts[2 * j], pts[2 * j + 1], &x1, &y1);
    int64_t dx = llabs(x1 - x0), dy = llabs(y1 - y0);
    int64_t sx = x1 >= x0 ? 1 : -1, sy = y1 >= y0 ? 1 : -1;
    int64_t err = dx - dy, x = x0, y = y0;
    for (;;) {
      double c = orc_cost(m, x, y);
      if (c > worst) worst = c;
      if (x == x1 && y == y1) break;
      int64_t e2 = 2 * err;
      if (e2 > -dy) { err -= dy; x += sx; }
      if (e2 < dx) { err += dx; y += sy; }
    }
  }
  return worst;
}

/* ------------------------------------------------------------------ Part 1: reference restatement */
/* py:176-178 */
double orc_yaw(double x, double y, double z, double w) {
  double t3 = 2.0 * (w * z + x * y);
  double t4 = 1.0 - 2.0 * (y * y + z * z);
  return atan2(t3, t4);
}

/* py:157-158 */
double orc_constraint(const neo_mpc_params* p, const double* u, int index) {
  double vx = u[3 * index], vy = u[3 * index + 1];
  return p->max_vel_trans - sqrt(vx * vx + vy * vy);
}

static double orc_odom_yaw(const neo_mpc_params* p, const neo_mpc_problem* q) {
  /* py:213: w taken from the goal pose unless the compat flag is cleared */
  double w = (p->compat_flags & NEO_MPC_COMPAT_ODOM_YAW_GOAL_W) ? q->goal_q[3] : q->cur_q[3];
  return orc_yaw(q->cur_q[0], q->cur_q[1], q->cur_q[2], w);
}

/* py:204-269 */
double orc_objective(const neo_mpc_params* p, const orc_map* m, const neo_mpc_problem* q,
                     const double* u, double footprint_cost) {
  const int n = p->control_steps;
  const double dt = p->prediction_horizon / n; /* py:137 */
  double target_yaw = orc_yaw(q->carrot_q[0], q->carrot_q[1], q->carrot_q[2], q->carrot_q[3]);
  double final_yaw = orc_yaw(q->goal_q[0], q->goal_q[1], q->goal_q[2], q->goal_q[3]);
  double odom_yaw = orc_odom_yaw(p, q);
  double total = 0.0, x = 0.0, y = 0.0, z = 0.0;
  double pos_x = q->cur_xy[0], pos_y = q->cur_xy[1];
  for (int i = 0; i < n; ++i) {
    double vx = u[3 * i], vy = u[3 * i + 1], wz = u[3 * i + 2];
    z += wz * dt;                                              /* py:230 */
    x += (vx * cos(z) * dt - vy * sin(z) * dt);               /* py:231 */
    y += (vx * sin(z) * dt + vy * cos(z) * dt);               /* py:232 */
    odom_yaw += wz * dt;                                       /* py:234 */
    pos_x += vx * cos(odom_yaw) * dt - vy * sin(odom_yaw) * dt; /* py:235 */
    pos_y += vx * sin(odom_yaw) * dt + vy * cos(odom_yaw) * dt; /* py:236 */
    int64_t mx, my;
    orc_world_to_map(m, pos_x, pos_y, &mx, &my);              /* py:246 */
    double c = orc_cost(m, mx, my);
    double costmap_cost = c * c;                               /* py:247 */
    double ddx = q->carrot_xy[0] - x, ddy = q->carrot_xy[1] - y;
    double dist = sqrt(ddx * ddx + ddy * ddy);                 /* py:250 */
    double eth = target_yaw - z;                               /* py:251 */
    total += ((p->w_trans * (dist * dist)) + (p->w_orient * (eth * eth))) / n; /* py:252 */
    double e0 = q->cur_vel[0] - vx, e1 = q->cur_vel[1] - vy, e2 = q->cur_vel[2] - wz;
    total += p->w_control * sqrt(e0 * e0 + e1 * e1 + e2 * e2) / n; /* py:253-254 */
    if (c == 1.0) total += costmap_cost * 1000 / n;           /* py:257-258 */
    else total += p->w_costmap * costmap_cost / n;            /* py:260 */
    if (footprint_cost == 1.0)                                 /* py:262-263 */
      total += (footprint_cost * footprint_cost) * p->w_footprint / n;
  }
  double gdx = q->carrot_xy[0] - q->goal_xyz[0], gdy = q->carrot_xy[1] - q->goal_xyz[1];
  double gdist = sqrt(gdx * gdx + gdy * gdy);                  /* py:266 */
  double eth = final_yaw - z;                                  /* py:267 */
  total += ((p->w_trans * (gdist * gdist)) + (p->w_orient * (eth * eth))) * p->w_terminal;
  return total;
}

/* py:312-341: returns 1 when a predicted cell costs >= 0.99 */
static int orc_collision_check(const neo_mpc_params* p, const orc_map* m, const neo_mpc_problem* q,
                               const double* x, double* path /* optional [n][3] */) {
  const int n = p->control_steps;
  const double dt = p->prediction_horizon / n;
  double pos_x = q->cur_xy[0], pos_y = q->cur_xy[1];
  double yaw = orc_yaw(q->cur_q[0], q->cur_q[1], q->cur_q[2], q->cur_q[3]); /* py:317 */
  int hit = 0;
  for (int i = 0; i < n; ++i) {
    yaw += x[3 * i + 2] * dt;
    pos_x += x[3 * i] * cos(yaw) * dt - x[3 * i + 1] * sin(yaw) * dt;
    pos_y += x[3 * i] * sin(yaw) * dt + x[3 * i + 1] * cos(yaw) * dt;
    if (path) { path[3 * i] = pos_x; path[3 * i + 1] = pos_y; path[3 * i + 2] = yaw; }
    int64_t mx, my;
    orc_world_to_map(m, pos_x, pos_y, &mx, &my);
    if (!hit && orc_cost(m, mx, my) >= 0.99) { hit = 1; if (!path) break; }
  }
  return hit;
}

/* py:358-361: returns 1 when the reset was taken */
static int orc_reset_if_new_goal(const neo_mpc_params* p, const neo_mpc_problem* q,
                                 neo_mpc_state* st, double* warm) {
  int same = st->has_old_goal;
  for (int k = 0; k < 3 && same; ++k) same = (st->old_goal[k] == q->goal_xyz[k]);
  for (int k = 0; k < 4 && same; ++k) same = (st->old_goal[3 + k] == q->goal_q[k]);
  if (same) return 0;
  for (int k = 0; k < 3 * p->control_steps; ++k) warm[k] = 0.0;
  st->last_control[0] = st->last_control[1] = st->last_control[2] = 0.0;
  st->waiting_time = 0.0;
  return 1;
}

/* py:365-403 given the raw solver output x (modified in place like x.x) */
static void orc_postprocess(const neo_mpc_params* p, const orc_map* m, const neo_mpc_problem* q,
                            neo_mpc_state* st, double* warm, double* x, int success,
                            double footprint_cost, neo_mpc_command* out, double* path) {
  const int n = p->control_steps;
  const double g = p->low_pass_gain;
  if (path) { /* py:293-306 rolls out the UNFILTERED solution (publishLocalPlan precedes py:366) */
    orc_collision_check(p, m, q, x, path);
  }
  for (int i = 0; i < 3; ++i) x[i] = x[i] * g + st->last_control[i] * (1 - g); /* py:366-367 */
  if (orc_collision_check(p, m, q, x, NULL)) st->collision = 1;               /* py:338-341 */
  st->collision_footprint = (footprint_cost == 1.0);                          /* py:343-347 */
  if (st->collision || st->collision_footprint) {                             /* py:374-382 */
    out->vel[0] = out->vel[1] = out->vel[2] = 0.0;
    out->flags |= NEO_MPC_FLAG_STOPPED;
    st->waiting_time += q->delta_t;
    if (st->waiting_time >= 3.0) { st->collision = 0; st->waiting_time = 0.0; }
  } else {                                                                    /* py:385-391 */
    const double acc[3] = {p->acc_x_limit, p->acc_y_limit, p->acc_theta_limit};
    for (int i = 0; i < 3; ++i) {
      double t = fmin(x[i], st->last_control[i] + acc[i] * q->control_interval);
      out->vel[i] = fmax(t, st->last_control[i] - acc[i] * q->control_interval);
    }
  }
  for (int i = 0; i < 3; ++i) st->last_control[i] = out->vel[i];              /* py:393-395 */
  if (success) {                                                              /* py:397-398, 198-202 */
    for (int i = 0; i < n - 1; ++i)
      for (int k = 0; k < 3; ++k) warm[3 * i + k] = x[3 * i + 3 + k];
    for (int k = 0; k < 3; ++k) warm[3 * (n - 1) + k] = x[k];
  } else {                                                                    /* py:399-400 */
    for (int k = 0; k < 3 * n; ++k) warm[k] = x[k];
  }
  for (int k = 0; k < 3; ++k) st->old_goal[k] = q->goal_xyz[k];               /* py:402 */
  for (int k = 0; k < 4; ++k) st->old_goal[3 + k] = q->goal_q[k];
  st->has_old_goal = 1;
}

/* ------------------------------------------------------------------ Part 2: CPU mirror of the build's solver */
typedef struct orc_ctx {
  int n;
  double dt, wt_n, wo_n, wc_n, wterm_o;
  double cx, cy, tyaw, fyaw, c0, s0, X0, Y0, v[3];
  double konst;          /* terms constant in u: terminal distance + footprint */
  double term[256];      /* per-step costmap term by raw cell value */
  double lo[3], hi[3], r;
  double kink_radius;
  const orc_map* map;
} orc_ctx;

static void orc_ctx_init(orc_ctx* c, const neo_mpc_params* p, const orc_map* m,
                         const neo_mpc_problem* q, double footprint_cost) {
  const int n = p->control_steps;
  c->n = n;
  c->dt = p->prediction_horizon / n;
  c->wt_n = p->w_trans / n;
  c->wo_n = p->w_orient / n;
  c->wc_n = p->w_control / n;
  c->wterm_o = p->w_terminal * p->w_orient;
  c->cx = q->carrot_xy[0];
  c->cy = q->carrot_xy[1];
  c->tyaw = orc_yaw(q->carrot_q[0], q->carrot_q[1], q->carrot_q[2], q->carrot_q[3]);
  c->fyaw = orc_yaw(q->goal_q[0], q->goal_q[1], q->goal_q[2], q->goal_q[3]);
  double psi0 = orc_odom_yaw(p, q);
  c->c0 = cos(psi0);
  c->s0 = sin(psi0);
  c->X0 = q->cur_xy[0];
  c->Y0 = q->cur_xy[1];
  for (int k = 0; k < 3; ++k) c->v[k] = q->cur_vel[k];
  double gdx = c->cx - q->goal_xyz[0], gdy = c->cy - q->goal_xyz[1];
  c->konst = p->w_terminal * p->w_trans * (gdx * gdx + gdy * gdy);
  if (footprint_cost == 1.0) c->konst += p->w_footprint;
  for (int raw = 0; raw < 256; ++raw) {
    double cc = (double)orc_occupancy(raw) / 100.0;
    c->term[raw] = (cc == 1.0) ? cc * cc * 1000 / n : p->w_costmap * (cc * cc) / n;
  }
  c->lo[0] = p->min_vel_x; c->hi[0] = p->max_vel_x;
  c->lo[1] = p->min_vel_y; c->hi[1] = p->max_vel_y;
  c->lo[2] = p->min_vel_theta; c->hi[2] = p->max_vel_theta;
  c->r = p->max_vel_trans;
  {
    /* (the stage-wise direction predicts landings on the kink inside its sweep: its prox-only zone is small) */
    neo_rules r;
    neo_rules_derive(p, &r);
    c->kink_radius = r.kink_radius;
  }
  c->map = m;
}

static double orc_clamp(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Euclidean projection of (vx, vy) onto box ∩ disc, omega clamped (py:125-134 feasible set). */
static void orc_project(const orc_ctx* c, double* b) {
  b[2] = orc_clamp(b[2], c->lo[2], c->hi[2]);
  const double zx = b[0], zy = b[1], r = c->r;
  double px = orc_clamp(zx, c->lo[0], c->hi[0]), py = orc_clamp(zy, c->lo[1], c->hi[1]);
  if (px * px + py * py <= r * r) { b[0] = px; b[1] = py; return; }
  double nz = sqrt(zx * zx + zy * zy);
  double qx = zx * (r / nz), qy = zy * (r / nz);
  if (qx >= c->lo[0] && qx <= c->hi[0] && qy >= c->lo[1] && qy <= c->hi[1]) { b[0] = qx; b[1] = qy; return; }
  /* both constraints bind: closest circle/box-edge intersection point */
  double best = INFINITY, bx = px, by = py;
  for (int e = 0; e < 4; ++e) {
    double fixed = (e == 0) ? c->lo[0] : (e == 1) ? c->hi[0] : (e == 2) ? c->lo[1] : c->hi[1];
    if (fabs(fixed) > r) continue;
    double o = sqrt(r * r - fixed * fixed);
    for (int s = -1; s <= 1; s += 2) {
      double ex = (e < 2) ? fixed : s * o, ey = (e < 2) ? s * o : fixed;
      if (ex < c->lo[0] || ex > c->hi[0] || ey < c->lo[1] || ey > c->hi[1]) continue;
      double d = (ex - zx) * (ex - zx) + (ey - zy) * (ey - zy);
      if (d < best) { best = d; bx = ex; by = ey; }
    }
  }
  b[0] = bx; b[1] = by;
}

static double orc_step_term(const orc_ctx* c, double x, double y) {
  double X = c->X0 + (c->c0 * x - c->s0 * y), Y = c->Y0 + (c->s0 * x + c->c0 * y);
  int64_t mx, my;
  orc_world_to_map(c->map, X, Y, &mx, &my);
  int raw = 254;
  if (mx >= 0 && my >= 0 && mx < c->map->size_x && my < c->map->size_y)
    raw = c->map->cells[my * (int64_t)c->map->size_x + mx];
  return c->term[raw];
}

/* the solver's objective (same value as orc_objective up to rounding) */
static double orc_eval(const orc_ctx* c, const double* u) {
  double f = 0.0, x = 0.0, y = 0.0, th = 0.0;
  for (int i = 0; i < c->n; ++i) {
    double vx = u[3 * i], vy = u[3 * i + 1], w = u[3 * i + 2];
    th += w * c->dt;
    double cs = cos(th), sn = sin(th);
    x += (vx * cs - vy * sn) * c->dt;
    y += (vx * sn + vy * cs) * c->dt;
    double dx = c->cx - x, dy = c->cy - y, et = c->tyaw - th;
    double e0 = c->v[0] - vx, e1 = c->v[1] - vy, e2 = c->v[2] - w;
    f += c->wt_n * (dx * dx + dy * dy) + c->wo_n * (et * et);
    f += c->wc_n * sqrt(e0 * e0 + e1 * e1 + e2 * e2);
    f += orc_step_term(c, x, y);
  }
  double et = c->fyaw - th;
  return f + c->wterm_o * (et * et) + c->konst;
}

/* does every stage of the rollout sit in a free cell (raw cost 0, inside the map)? */
static int orc_free_path(const orc_ctx* c, const double* u) {
  double x = 0.0, y = 0.0, th = 0.0;
  for (int i = 0; i < c->n; ++i) {
    th += u[3 * i + 2] * c->dt;
    const double cs = cos(th), sn = sin(th);
    x += (u[3 * i] * cs - u[3 * i + 1] * sn) * c->dt;
    y += (u[3 * i] * sn + u[3 * i + 1] * cs) * c->dt;
    const double X = c->X0 + (c->c0 * x - c->s0 * y), Y = c->Y0 + (c->s0 * x + c->c0 * y);
    int64_t mx, my;
    orc_world_to_map(c->map, X, Y, &mx, &my);
    if (mx < 0 || my < 0 || mx >= c->map->size_x || my >= c->map->size_y) return 0;
    if (c->map->cells[my * (int64_t)c->map->size_x + mx] != 0) return 0;
  }
  return 1;
}

/* gradient of the smooth (tracking + terminal) part, adjoint sweep */
static void orc_grad_smooth(const orc_ctx* c, const double* u, double* g) {
  const int n = c->n;
  double cs[ORC_MAXN], sn[ORC_MAXN], dxs[ORC_MAXN], dys[ORC_MAXN], rx[ORC_MAXN], ry[ORC_MAXN], rt[ORC_MAXN] = {0.0};
  double x = 0.0, y = 0.0, th = 0.0;
  for (int i = 0; i < n; ++i) {
    double vx = u[3 * i], vy = u[3 * i + 1], w = u[3 * i + 2];
    th += w * c->dt;
    cs[i] = cos(th); sn[i] = sin(th);
    dxs[i] = (vx * cs[i] - vy * sn[i]) * c->dt;
    dys[i] = (vx * sn[i] + vy * cs[i]) * c->dt;
    x += dxs[i]; y += dys[i];
    rx[i] = -2.0 * c->wt_n * (c->cx - x);
    ry[i] = -2.0 * c->wt_n * (c->cy - y);
    rt[i] = -2.0 * c->wo_n * (c->tyaw - th);
  }
  rt[n - 1] += -2.0 * c->wterm_o * (c->fyaw - th);
  double SX = 0.0, SY = 0.0, ST = 0.0;
  for (int k = n - 1; k >= 0; --k) {
    SX += rx[k]; SY += ry[k];
    ST += rt[k] - dys[k] * SX + dxs[k] * SY;
    g[3 * k] = c->dt * (cs[k] * SX + sn[k] * SY);
    g[3 * k + 1] = c->dt * (-sn[k] * SX + cs[k] * SY);
    g[3 * k + 2] = c->dt * ST;
  }
}

/* total gradient gt (smooth + control-norm, minimal-norm subgradient at the kink), reduced
 * gradient gr = minus the projection of -gt onto the tangent cone of the feasible set at u,
 * and the active-set description used to restrict the quasi-Newton direction. */
typedef struct orc_active {
  uint8_t wfroz[ORC_MAXN]; /* omega at a bound, gradient pushing outward */
  uint8_t mode[ORC_MAXN];  /* (vx, vy): 0 free, 1 slide along the constraint with normal n, 2 pinned */
  uint8_t near[ORC_MAXN];  /* block within kink_radius of the control-norm kink u_i = v_cur: it is
                              moved by the proximal step in every candidate and kept out of the
                              quasi-Newton model (the norm's curvature wc/|u_i - v| is unbounded there) */
  uint8_t disc[ORC_MAXN];  /* mode 1 and the binding constraint is the disc (curved) */
  double nx[ORC_MAXN], ny[ORC_MAXN];
  double lambda[ORC_MAXN]; /* mode 1: n . (-g) > 0, the multiplier estimate of the binding constraint */
  uint8_t tokink[ORC_MAXN]; /* Riccati direction: the stage model's minimiser is the kink itself (d_i = v - u_i) */
  int riccati;              /* the Riccati kernel's candidate rules apply (per-block prox step) */
  int longshots;            /* lanes 61-63 are step lengths 8, 16, 32 (Newton directions) */
  uint8_t rest[ORC_MAXN];   /* near block exactly on the kink whose smooth gradient keeps it there */
} orc_active;


/* Tangent-cone reduction of a block's gradient g at u_i: omega frozen at a bound the descent direction -g pushes into;
 * (vx, vy) free, sliding along one active constraint (longest slide that keeps the others satisfied) or pinned.
 * Out: the reduced gradient gr[3] and the face (wfroz, mode 0/1/2, outward normal, disc flag, multiplier). */
static int orc_near_reduced = 1;
void orc_set_near_reduced(int m) { orc_near_reduced = m; }
static void orc_tangent(const orc_ctx* c, const double* ui, const double* gi, double* gr, uint8_t* wfroz, uint8_t* mode,
                        double* nxo, double* nyo, uint8_t* disc, double* lambda) {
  gr[0] = gi[0]; gr[1] = gi[1]; gr[2] = gi[2];
  /* omega: plain bound */
  *wfroz = (ui[2] <= c->lo[2] && gi[2] > 0.0) || (ui[2] >= c->hi[2] && gi[2] < 0.0);
  if (*wfroz) gr[2] = 0.0;
  /* (vx, vy): outward normals of the constraints active at u */
  double nx[3], ny[3];
  int isdisc[3] = {0, 0, 0};
  int na = 0;
  /* (the disc first: where a bound touches the disc with the same normal -- max_vel_x = max_vel_trans, README -- the slide
   * is the disc's, with its curvature and without the corner stop of a bound slide) */
  double nv = sqrt(ui[0] * ui[0] + ui[1] * ui[1]);
  if (nv > 0.0 && nv >= c->r * (1.0 - 1e-12)) { nx[na] = ui[0] / nv; ny[na] = ui[1] / nv; isdisc[na] = 1; ++na; }
  /* (a bound is active within NEO_RULE_CORNER_ROOM of it: the projection's radial rescale leaves a block that sat on a bound
   * a rounding error inside it -- seen as free, a block in the corner between the disc and a bound slid along the disc INTO the
   * bound, was pinned by the corner re-pin, and never tried the slide along the bound that led on: round-5 review, seed 62022) */
  const double room = NEO_RULE_CORNER_ROOM;
  if (ui[0] <= c->lo[0] + room) { nx[na] = -1.0; ny[na] = 0.0; ++na; }
  else if (ui[0] >= c->hi[0] - room) { nx[na] = 1.0; ny[na] = 0.0; ++na; }
  if (ui[1] <= c->lo[1] + room) { nx[na] = 0.0; ny[na] = -1.0; ++na; }
  else if (ui[1] >= c->hi[1] - room) { nx[na] = 0.0; ny[na] = 1.0; ++na; }
  const double dx = -gi[0], dy = -gi[1]; /* steepest descent */
  *mode = 0; *nxo = 0.0; *nyo = 0.0; *disc = 0; *lambda = 0.0;
  int violated = 0;
  for (int k = 0; k < na; ++k) if (nx[k] * dx + ny[k] * dy > 0.0) violated = 1;
  if (violated) {
    double bestn = -1.0;
    int bestk = -1;
    for (int k = 0; k < na; ++k) {
      double dn = nx[k] * dx + ny[k] * dy;
      if (!(dn > 0.0)) continue;
      double px = dx - dn * nx[k], py = dy - dn * ny[k];
      int ok = 1;
      for (int j = 0; j < na; ++j)
        if (j != k && nx[j] * px + ny[j] * py > 1e-14 * (fabs(px) + fabs(py))) ok = 0;
      double pn = px * px + py * py;
      if (ok && pn > bestn) { bestn = pn; bestk = k; }
    }
    if (bestk >= 0) {
      double dn = nx[bestk] * dx + ny[bestk] * dy;
      *mode = 1; *nxo = nx[bestk]; *nyo = ny[bestk];
      *disc = (uint8_t)isdisc[bestk]; *lambda = dn;
      gr[0] = -(dx - dn * nx[bestk]);
      gr[1] = -(dy - dn * ny[bestk]);
    } else {
      *mode = 2;
      gr[0] = 0.0; gr[1] = 0.0;
    }
  }
}

/* (round 4) gs is rewritten for blocks next to the kink: their SMOOTH gradient reduced on the tangent cone -- the
 * proximal step of such a block is taken on its face (a component that pushes omega into its bound, or the velocity out
 * of the disc, used to dominate the step's shrink factor and keep the block from landing on the kink; projection and
 * prox do not commute), and a block ON the kink is at rest when the REDUCED smooth gradient lies inside the norm's
 * subdifferential. */
static void orc_reduce(const orc_ctx* c, const double* u, double* gs, double* gt, double* gr,
                       orc_active* a) {
  for (int i = 0; i < c->n; ++i) {
    const double* ui = u + 3 * i;
    double* gsi = gs + 3 * i;
    a->tokink[i] = 0;
    double e[3] = {ui[0] - c->v[0], ui[1] - c->v[1], ui[2] - c->v[2]};
    double ne = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    a->near[i] = ne < c->kink_radius;
    if (a->near[i]) {
      if (orc_near_reduced) {
        double gsr[3], nxo, nyo, lam;
        uint8_t wf, md, dc;
        orc_tangent(c, ui, gsi, gsr, &wf, &md, &nxo, &nyo, &dc, &lam);
        gsi[0] = gsr[0]; gsi[1] = gsr[1]; gsi[2] = gsr[2];
      }
      /* a block sitting exactly ON the kink with a smooth gradient inside the norm's subdifferential
       * (|g_s| <= w_control/N) stays there under every proximal step: it is at rest */
      a->rest[i] = ne == 0.0 && gsi[0] * gsi[0] + gsi[1] * gsi[1] + gsi[2] * gsi[2] <= c->wc_n * c->wc_n;
      for (int k = 0; k < 3; ++k) { gt[3 * i + k] = 0.0; gr[3 * i + k] = 0.0; }
      a->wfroz[i] = 0; a->mode[i] = 0; a->nx[i] = 0.0; a->ny[i] = 0.0; a->disc[i] = 0; a->lambda[i] = 0.0;
      continue;
    }
    a->rest[i] = 0;
    double gi[3];
    if (ne > 0.0) {
      for (int k = 0; k < 3; ++k) gi[k] = gsi[k] + c->wc_n * (e[k] / ne);
    } else {
      double ng = sqrt(gsi[0] * gsi[0] + gsi[1] * gsi[1] + gsi[2] * gsi[2]);
      double sh = (ng > c->wc_n) ? 1.0 - c->wc_n / ng : 0.0;
      for (int k = 0; k < 3; ++k) gi[k] = gsi[k] * sh;
    }
    for (int k = 0; k < 3; ++k) gt[3 * i + k] = gi[k];
    orc_tangent(c, ui, gi, gr + 3 * i, &a->wfroz[i], &a->mode[i], &a->nx[i], &a->ny[i], &a->disc[i], &a->lambda[i]);
  }
}

static void orc_apply_active(const orc_ctx* c, const orc_active* a, double* d) {
  for (int i = 0; i < c->n; ++i) {
    if (a->tokink[i]) continue;
    if (a->near[i]) { d[3 * i] = 0.0; d[3 * i + 1] = 0.0; d[3 * i + 2] = 0.0; continue; }
    if (a->wfroz[i]) d[3 * i + 2] = 0.0;
    if (a->mode[i] == 1) {
      double dot = d[3 * i] * a->nx[i] + d[3 * i + 1] * a->ny[i];
      d[3 * i] -= dot * a->nx[i]; d[3 * i + 1] -= dot * a->ny[i];
    } else if (a->mode[i] == 2) {
      d[3 * i] = 0.0; d[3 * i + 1] = 0.0;
    }
  }
}

/* Projected Newton direction (method NEWTON): Hessian of the smooth part by forward differences
 * of the analytic gradient (one column per perturbed coordinate -- one lane each on the GPU),
 * plus the control norm's Hessian on blocks away from the kink and the curvature lambda/r of a
 * binding disc; restricted to the tangent cone's face with the projector P (H_r = P H P + I - P)
 * and solved by Gaussian elimination without pivoting, non-positive pivots replaced. */
#define ORC_NEWTON_MAXV 24
static int orc_newton_f64 = 0;
static void orc_newton_direction(const orc_ctx* c, const double* u, const double* gs, const double* gr,
                                 const orc_active* a, double* d) {
  const int n = c->n, nv = 3 * n;
  const double h = 1e-6;
  double H[ORC_NEWTON_MAXV][ORC_NEWTON_MAXV], rhs[ORC_NEWTON_MAXV], up[ORC_NEWTON_MAXV], gk[ORC_NEWTON_MAXV];
  for (int k = 0; k < nv; ++k) {
    memcpy(up, u, sizeof(double) * nv);
    up[k] += h;
    orc_grad_smooth(c, up, gk);
    for (int j = 0; j < nv; ++j) H[j][k] = (gk[j] - gs[j]) / h;
  }
  double P00[ORC_MAXN], P01[ORC_MAXN], P11[ORC_MAXN], PW[ORC_MAXN];
  for (int i = 0; i < n; ++i) {
    const double* ui = u + 3 * i;
    if (a->near[i]) { P00[i] = P01[i] = P11[i] = PW[i] = 0.0; continue; }
    double e[3] = {ui[0] - c->v[0], ui[1] - c->v[1], ui[2] - c->v[2]};
    double ne = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    if (ne > 0.0) { /* Hessian of (w_control/N) |u_i - v|: (w/|e|)(I - e e^T/|e|^2) */
      const double s = c->wc_n / ne;
      for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q)
          H[3 * i + r][3 * i + q] += s * ((r == q ? 1.0 : 0.0) - (e[r] / ne) * (e[q] / ne));
    }
    PW[i] = a->wfroz[i] ? 0.0 : 1.0;
    if (a->mode[i] == 0) { P00[i] = 1.0; P01[i] = 0.0; P11[i] = 1.0; }
    else if (a->mode[i] == 1) {
      const double nx = a->nx[i], ny = a->ny[i];
      P00[i] = 1.0 - nx * nx; P01[i] = -nx * ny; P11[i] = 1.0 - ny * ny;
      if (a->disc[i]) { /* moving along the circle: second-order term lambda/r on the tangent */
        const double k2 = a->lambda[i] / c->r, tx = -ny, ty = nx;
        H[3 * i][3 * i] += k2 * tx * tx; H[3 * i][3 * i + 1] += k2 * tx * ty;
        H[3 * i + 1][3 * i] += k2 * ty * tx; H[3 * i + 1][3 * i + 1] += k2 * ty * ty;
      }
    } else { P00[i] = P01[i] = P11[i] = 0.0; }
  }
  for (int j = 0; j < nv; ++j) /* H <- H P (row j) */
    for (int i = 0; i < n; ++i) {
      const double hx = H[j][3 * i], hy = H[j][3 * i + 1];
      H[j][3 * i] = hx * P00[i] + hy * P01[i];
      H[j][3 * i + 1] = hx * P01[i] + hy * P11[i];
      H[j][3 * i + 2] *= PW[i];
    }
  for (int k = 0; k < nv; ++k) /* H <- P H (column k) */
    for (int i = 0; i < n; ++i) {
      const double hx = H[3 * i][k], hy = H[3 * i + 1][k];
      H[3 * i][k] = P00[i] * hx + P01[i] * hy;
      H[3 * i + 1][k] = P01[i] * hx + P11[i] * hy;
      H[3 * i + 2][k] *= PW[i];
    }
  double dmax = 0.0;
  for (int i = 0; i < n; ++i) { /* + (I - P) */
    H[3 * i][3 * i] += 1.0 - P00[i]; H[3 * i][3 * i + 1] -= P01[i];
    H[3 * i + 1][3 * i] -= P01[i]; H[3 * i + 1][3 * i + 1] += 1.0 - P11[i];
    H[3 * i + 2][3 * i + 2] += 1.0 - PW[i];
  }
  for (int j = 0; j < nv; ++j) { rhs[j] = -gr[j]; dmax = fmax(dmax, fabs(H[j][j])); }
  if (orc_newton_f64) { /* test hook: the same elimination in float64 */
    const double delta = fmax(1e-6 * dmax, 1e-30);
    for (int p = 0; p < nv; ++p) {
      double piv = H[p][p];
      if (!(piv > delta)) piv = fmax(fabs(piv), delta);
      H[p][p] = piv;
      for (int j = p + 1; j < nv; ++j) {
        const double fac = H[j][p] / piv;
        for (int q = p + 1; q < nv; ++q) H[j][q] -= fac * H[p][q];
        rhs[j] -= fac * rhs[p];
      }
    }
    for (int p = nv - 1; p >= 0; --p) {
      double acc = rhs[p];
      for (int q = p + 1; q < nv; ++q) acc -= H[p][q] * d[q];
      d[p] = acc / H[p][p];
    }
    return;
  }
  /* the reduced system is solved in float32 (as the kernel does): it only yields a search direction */
  float Hf[ORC_NEWTON_MAXV][ORC_NEWTON_MAXV], rf[ORC_NEWTON_MAXV], df[ORC_NEWTON_MAXV];
  for (int j = 0; j < nv; ++j) { rf[j] = (float)rhs[j]; for (int q = 0; q < nv; ++q) Hf[j][q] = (float)H[j][q]; }
  const float delta = fmaxf(1e-6f * (float)dmax, 1e-30f);
  for (int p = 0; p < nv; ++p) {
    float piv = Hf[p][p];
    if (!(piv > delta)) piv = fmaxf(fabsf(piv), delta);
    Hf[p][p] = piv;
    const float inv = 1.0f / piv;
    for (int j = p + 1; j < nv; ++j) {
      const float fac = Hf[j][p] * inv;
      for (int q = p + 1; q < nv; ++q) Hf[j][q] -= fac * Hf[p][q];
      rf[j] -= fac * rf[p];
    }
  }
  for (int p = nv - 1; p >= 0; --p) {
    float acc = rf[p];
    for (int q = p + 1; q < nv; ++q) acc -= Hf[p][q] * df[q];
    df[p] = acc * (1.0f / Hf[p][p]);
    d[p] = (double)df[p];
  }
}

/* Projected Newton direction by the stage-wise (Riccati) recursion (method RICCATI), any control_steps.
 * The same system as orc_newton_direction -- H_r d = -g_r with H the exact Hessian of the smooth part
 * plus the control norm's and a binding disc's curvature, restricted to the tangent cone's face --
 * but solved as the linear-quadratic problem it is: the rollout (py:230-232) is a chain
 * z_i = F(z_{i-1}, u_i), z = (x, y, theta), so
 *     H = sum_i J_i^T W_i J_i + sum_i lambda_i . d2F_i + blockdiag(R_i),
 * W_i the stage cost's Hessian, lambda_i = (SX_i, SY_i) the costate of the adjoint sweep, J_i the
 * sensitivity of z_i.  One backward sweep over the stages with 3x3 value-function Hessians and one
 * forward sweep give d in O(control_steps) -- no (3N)^2 matrix, no finite differences. */
static int orc_piv_replaced;
static void orc_sym3_solve_prepare(double Q[3][3], double L[3][3], double delta) {
  /* LDL^T-free Cholesky-like elimination without pivoting, non-positive pivots replaced:
   * L holds the eliminated upper triangle rows (as orc_newton_direction does for the dense system) */
  for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) L[r][q] = Q[r][q];
  for (int p = 0; p < 3; ++p) {
    double piv = L[p][p];
    if (!(piv > delta)) { piv = fmax(fabs(piv), delta); ++orc_piv_replaced; }
    L[p][p] = piv;
    for (int j = p + 1; j < 3; ++j) {
      const double fac = L[j][p] / piv;
      L[j][p] = fac; /* multiplier kept below the diagonal */
      for (int q = p + 1; q < 3; ++q) L[j][q] -= fac * L[p][q];
    }
  }
}
static void orc_sym3_solve(double L[3][3], const double* b, double* x) {
  double y[3] = {b[0], b[1], b[2]};
  for (int p = 0; p < 3; ++p)
    for (int j = p + 1; j < 3; ++j) y[j] -= L[j][p] * y[p];
  for (int p = 2; p >= 0; --p) {
    double acc = y[p];
    for (int q = p + 1; q < 3; ++q) acc -= L[p][q] * x[q];
    x[p] = acc / L[p][p];
  }
}

#define ORC_STICKY NEO_RULE_STICKY
#define ORC_STICKY_DIST NEO_RULE_STICKY_DIST
static double orc_term_at(const orc_ctx* c, int64_t mx, int64_t my) {
  int raw = 254;
  if (mx >= 0 && my >= 0 && mx < c->map->size_x && my < c->map->size_y)
    raw = c->map->cells[my * (int64_t)c->map->size_x + mx];
  return c->term[raw];
}
/* Wall model of the stage-wise direction.  A stage whose position (x, y: rollout frame) sits within ORC_STICKY_DIST
 * cells of a cell edge behind which the costmap term is higher gets motion along that edge's normal penalised in the
 * stage model, W += rho n n^T (n = the edge normal, a world axis, in the rollout's frame): the Newton direction then
 * slides along the cost step instead of running into it at every step length (searches used to die creeping towards
 * such an edge).  rho = ORC_STICKY x the tracking curvature for an ordinary cost step.  Behind a LETHAL cell (or the
 * map's border) the edge is a wall no candidate will ever cross: within ORC_WALL_DIST cells of it rho = ORC_WALL x the
 * tracking curvature and the penalty is centred ORC_WALL_DIST cells inside, 1/2 rho (n . dz - pb)^2 -- its linear term
 * l = -rho pb n pushes the stage back to that stand-off, so that a finite step along the wall does not end inside it
 * (the stage's path along a straight wall is curved in the controls: with a stand-off of 2 % of a cell a slide advances
 * a centimetre per iteration, 50 iterations in the case below; with 10 % it takes 17).  (With the soft penalty
 * alone a search blocked by a lethal cell crept up to the wall -- 1e-6 cells -- and ended there with every candidate
 * lethal: 0.79 above the reference's SLSQP value in one case of the G8 fixtures; with the wall model it ends ON the
 * reference's optimum.) */
#define ORC_WALL NEO_RULE_WALL
#define ORC_WALL_DIST NEO_RULE_WALL_DIST
static void orc_wall_model(const orc_ctx* c, double x, double y, double* W, double* l) {
  const double X = c->X0 + (c->c0 * x - c->s0 * y), Y = c->Y0 + (c->s0 * x + c->c0 * y);
  const orc_map* m = c->map;
  int64_t mx, my;
  orc_world_to_map(m, X, Y, &mx, &my);
  const double fx = (X - m->origin_x) * (1.0 / m->resolution) - (double)mx, fy = (Y - m->origin_y) * (1.0 / m->resolution) - (double)my;
  const double here = orc_term_at(c, mx, my), lethal = c->term[254];
  const int64_t nbx[4] = {mx - 1, mx + 1, mx, mx}, nby[4] = {my, my, my - 1, my + 1};
  const double dist[4] = {fx, 1.0 - fx, fy, 1.0 - fy}, sign[4] = {1.0, -1.0, 1.0, -1.0};
  W[0] = W[1] = W[2] = 0.0; l[0] = l[1] = 0.0;
  for (int k = 0; k < 4; ++k) {
    const double there = orc_term_at(c, nbx[k], nby[k]);
    const int wall = there >= lethal && here < lethal;
    const double zone = wall ? ORC_WALL_DIST : ORC_STICKY_DIST;
    if (!(dist[k] < zone && there > here)) continue;
    const double rho = (wall ? ORC_WALL : ORC_STICKY) * 2.0 * c->wt_n;
    /* the world axis the edge is normal to, in the rollout's frame: x -> (c0, -s0), y -> (s0, c0) */
    const double nlx = k < 2 ? c->c0 : c->s0, nly = k < 2 ? -c->s0 : c->c0;
    W[0] += rho * nlx * nlx; W[1] += rho * nlx * nly; W[2] += rho * nly * nly;
    if (wall) { /* push back along +axis (wall on the low side) or -axis (wall on the high side), metres */
      const double pb = sign[k] * (zone - dist[k]) * m->resolution;
      l[0] -= rho * pb * nlx; l[1] -= rho * pb * nly;
    }
  }
}
static int orc_kink_predict = 1; /* (the debug hook switches it off to compare with the dense direction) */
static void orc_riccati_direction(const orc_ctx* c, const double* u, const double* gs, const double* gt, orc_active* a,
                                  double* d) {
  const int n = c->n;
  const double dt = c->dt;
  double cs[ORC_MAXN], sn[ORC_MAXN], px[ORC_MAXN], py[ORC_MAXN], SX[ORC_MAXN], SY[ORC_MAXN], xs_[ORC_MAXN], ys_[ORC_MAXN];
  { /* nominal rollout and position costates */
    double x = 0.0, y = 0.0, th = 0.0, rx[ORC_MAXN], ry[ORC_MAXN];
    for (int i = 0; i < n; ++i) {
      th += u[3 * i + 2] * dt;
      cs[i] = cos(th); sn[i] = sin(th);
      px[i] = (u[3 * i] * cs[i] - u[3 * i + 1] * sn[i]) * dt;
      py[i] = (u[3 * i] * sn[i] + u[3 * i + 1] * cs[i]) * dt;
      x += px[i]; y += py[i];
      xs_[i] = x; ys_[i] = y;
      rx[i] = -2.0 * c->wt_n * (c->cx - x);
      ry[i] = -2.0 * c->wt_n * (c->cy - y);
    }
    double ax = 0.0, ay = 0.0;
    for (int i = n - 1; i >= 0; --i) { ax += rx[i]; ay += ry[i]; SX[i] = ax; SY[i] = ay; }
  }
  static _Thread_local double Kf[ORC_MAXN][3][3], kf[ORC_MAXN][3];
  double V[3][3] = {{0}}, v[3] = {0.0, 0.0, 0.0};
  for (int i = n - 1; i >= 0; --i) {
    /* S = W_i + V: Hessian of (stage cost at z_i + cost to go) */
    double S[3][3];
    for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) S[r][q] = V[r][q];
    S[0][0] += 2.0 * c->wt_n; S[1][1] += 2.0 * c->wt_n;
    S[2][2] += 2.0 * c->wo_n + (i == n - 1 ? 2.0 * c->wterm_o : 0.0);
    { /* wall model (orc_wall_model): curvature on the stage position, linear push-back from a lethal wall */
      double W[3], l[2];
      orc_wall_model(c, xs_[i], ys_[i], W, l);
      S[0][0] += W[0]; S[0][1] += W[1]; S[1][0] += W[1]; S[1][1] += W[2];
      v[0] += l[0]; v[1] += l[1];
    }
    const double A[3][3] = {{1, 0, -py[i]}, {0, 1, px[i]}, {0, 0, 1}};
    const double B[3][3] = {{dt * cs[i], -dt * sn[i], -py[i] * dt}, {dt * sn[i], dt * cs[i], px[i] * dt}, {0, 0, dt}};
    double SA[3][3], SB[3][3], Qzz[3][3], Quz[3][3], Quu[3][3], Qz[3], Qu[3];
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 3; ++q) {
        SA[r][q] = 0.0; SB[r][q] = 0.0;
        for (int k = 0; k < 3; ++k) { SA[r][q] += S[r][k] * A[k][q]; SB[r][q] += S[r][k] * B[k][q]; }
      }
    for (int r = 0; r < 3; ++r) {
      Qz[r] = 0.0; Qu[r] = gt[3 * i + r];
      for (int k = 0; k < 3; ++k) { Qz[r] += A[k][r] * v[k]; Qu[r] += B[k][r] * v[k]; }
      for (int q = 0; q < 3; ++q) {
        Qzz[r][q] = 0.0; Quz[r][q] = 0.0; Quu[r][q] = 0.0;
        for (int k = 0; k < 3; ++k) {
          Qzz[r][q] += A[k][r] * SA[k][q]; Quz[r][q] += B[k][r] * SA[k][q]; Quu[r][q] += B[k][r] * SB[k][q];
        }
      }
    }
    /* second-order terms of the rollout step, weighted by the costate (SX_i, SY_i) of its result:
     * with xi = theta_{i-1} + w dt, d2(lambda . p)/d(vx, vy, xi)^2 = [[0 0 ax], [0 0 ay], [ax ay kappa]] */
    const double ax = dt * (-SX[i] * sn[i] + SY[i] * cs[i]), ay = dt * (-SX[i] * cs[i] - SY[i] * sn[i]);
    const double kappa = -(SX[i] * px[i] + SY[i] * py[i]);
    Qzz[2][2] += kappa;
    Quz[0][2] += ax; Quz[1][2] += ay; Quz[2][2] += dt * kappa;
    Quu[0][2] += dt * ax; Quu[2][0] += dt * ax; Quu[1][2] += dt * ay; Quu[2][1] += dt * ay; Quu[2][2] += dt * dt * kappa;
    /* block curvature R_i and the face projector P_i (as in orc_newton_direction) */
    double P[3][3] = {{0}};
    a->tokink[i] = 0;
    if (!a->near[i] && orc_kink_predict) {
      /* does the stage model put this block ON the kink u_i = v_cur?  0 in Qu_s + Quu_s k + w d|u_i + k - v|
       * at k = v - u_i  <=>  |Qu_s + Quu_s (v - u_i)| <= w  (smooth parts only) */
      const double* ui = u + 3 * i;
      double kk[3] = {c->v[0] - ui[0], c->v[1] - ui[1], c->v[2] - ui[2]}, r[3];
      for (int q = 0; q < 3; ++q) {
        r[q] = gs[3 * i + q];
        for (int k = 0; k < 3; ++k) r[q] += B[k][q] * v[k] + Quu[q][k] * kk[k];
      }
      double vv[3] = {c->v[0], c->v[1], c->v[2]}, vp[3] = {c->v[0], c->v[1], c->v[2]};
      orc_project(c, vp);
      const int feasible = vp[0] == vv[0] && vp[1] == vv[1] && vp[2] == vv[2];
      if (feasible && r[0] * r[0] + r[1] * r[1] + r[2] * r[2] <= c->wc_n * c->wc_n) a->tokink[i] = 1;
    }
    if (a->tokink[i]) {
      const double* ui = u + 3 * i;
      for (int r = 0; r < 3; ++r) {
        kf[i][r] = c->v[r] - ui[r];
        for (int q = 0; q < 3; ++q) Kf[i][r][q] = 0.0;
      }
      /* fixed step, no feedback: v = Qz + Quz^T k, V = Qzz */
      for (int r = 0; r < 3; ++r) {
        v[r] = Qz[r];
        for (int k = 0; k < 3; ++k) v[r] += Quz[k][r] * kf[i][k];
        for (int q = 0; q < 3; ++q) V[r][q] = Qzz[r][q];
      }
      continue;
    }
    if (!a->near[i]) {
      const double* ui = u + 3 * i;
      double e[3] = {ui[0] - c->v[0], ui[1] - c->v[1], ui[2] - c->v[2]};
      double ne = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
      if (ne > 0.0) {
        const double s = c->wc_n / ne;
        for (int r = 0; r < 3; ++r)
          for (int q = 0; q < 3; ++q) Quu[r][q] += s * ((r == q ? 1.0 : 0.0) - (e[r] / ne) * (e[q] / ne));
      }
      P[2][2] = a->wfroz[i] ? 0.0 : 1.0;
      if (a->mode[i] == 0) { P[0][0] = 1.0; P[1][1] = 1.0; }
      else if (a->mode[i] == 1) {
        const double nx = a->nx[i], ny = a->ny[i];
        P[0][0] = 1.0 - nx * nx; P[0][1] = P[1][0] = -nx * ny; P[1][1] = 1.0 - ny * ny;
        if (a->disc[i]) {
          const double k2 = a->lambda[i] / c->r, tx = -ny, ty = nx;
          Quu[0][0] += k2 * tx * tx; Quu[0][1] += k2 * tx * ty; Quu[1][0] += k2 * ty * tx; Quu[1][1] += k2 * ty * ty;
        }
      }
    }
    double T[3][3], Qur[3], Quzr[3][3], Quur[3][3];
    for (int r = 0; r < 3; ++r) {
      Qur[r] = 0.0;
      for (int k = 0; k < 3; ++k) Qur[r] += P[r][k] * Qu[k];
      for (int q = 0; q < 3; ++q) {
        T[r][q] = 0.0; Quzr[r][q] = 0.0;
        for (int k = 0; k < 3; ++k) { T[r][q] += P[r][k] * Quu[k][q]; Quzr[r][q] += P[r][k] * Quz[k][q]; }
      }
    }
    double dmax = 0.0;
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 3; ++q) {
        Quur[r][q] = (r == q ? 1.0 : 0.0) - P[r][q];
        for (int k = 0; k < 3; ++k) Quur[r][q] += T[r][k] * P[k][q];
      }
    for (int r = 0; r < 3; ++r) dmax = fmax(dmax, fabs(Quur[r][r]));
    double L[3][3];
    orc_sym3_solve_prepare(Quur, L, fmax(1e-6 * dmax, 1e-30));
    double rhs[3] = {-Qur[0], -Qur[1], -Qur[2]};
    orc_sym3_solve(L, rhs, kf[i]);
    for (int q = 0; q < 3; ++q) {
      double col[3] = {-Quzr[0][q], -Quzr[1][q], -Quzr[2][q]}, sol[3];
      orc_sym3_solve(L, col, sol);
      for (int r = 0; r < 3; ++r) Kf[i][r][q] = sol[r];
    }
    for (int r = 0; r < 3; ++r) {
      v[r] = Qz[r];
      for (int k = 0; k < 3; ++k) v[r] += Quzr[k][r] * kf[i][k];
      for (int q = 0; q < 3; ++q) {
        V[r][q] = Qzz[r][q];
        for (int k = 0; k < 3; ++k) V[r][q] += Quzr[k][r] * Kf[i][k][q];
      }
    }
    for (int r = 0; r < 3; ++r) for (int q = r + 1; q < 3; ++q) { V[r][q] = V[q][r] = 0.5 * (V[r][q] + V[q][r]); }
  }
  double dz[3] = {0.0, 0.0, 0.0};
  for (int i = 0; i < n; ++i) {
    double du[3];
    for (int r = 0; r < 3; ++r) {
      du[r] = kf[i][r];
      for (int q = 0; q < 3; ++q) du[r] += Kf[i][r][q] * dz[q];
      d[3 * i + r] = du[r];
    }
    const double nx0 = dz[0] - py[i] * dz[2] + dt * (cs[i] * du[0] - sn[i] * du[1]) - py[i] * dt * du[2];
    const double ny0 = dz[1] + px[i] * dz[2] + dt * (sn[i] * du[0] + cs[i] * du[1]) + px[i] * dt * du[2];
    dz[2] += dt * du[2]; dz[0] = nx0; dz[1] = ny0;
  }
}

/* The same direction computed the way the device does (riccati.h): in the DISPLACEMENT coordinates of every
 * stage, w = B0 du with B0 = dt diag(Rot(theta_i), 1).  The linearised step is then dz_i = A_i (dz_{i-1} + w_i),
 * the Gauss-Newton part of Quu, Quz and Qzz is one matrix M = A^T S A, the second-order terms of the step are
 * T = [[0 0 SY] [0 0 -SX] [SY -SX kappa]], and every stage is solved in the coordinates of its face (0-3 free
 * directions) instead of through projector products.  float64 here, float32 on the device. */
static int orc_disp = 1;
static _Thread_local double orc_mu = 0.0;   /* damping of the displacement-form direction (set per call) */
void orc_set_disp(int on) { orc_disp = on; }
static int orc_piv_replaced = 0;
static double orc_piv(double p, double delta) { if (!(p > delta)) ++orc_piv_replaced; return p > delta ? p : fmax(fabs(p), delta); }
int orc_get_piv_replaced(void) { int r = orc_piv_replaced; orc_piv_replaced = 0; return r; }
static void orc_riccati_direction_disp_tau(const orc_ctx* c, const double* u, const double* gs, const double* gt,
                                       orc_active* a, double* d, double tau);
/* What the solver uses: the GAUSS-NEWTON part of the Hessian (tau = 0: the lambda . d2F terms of the rollout
 * step are left out).  Every stage system is then positive definite by construction and the recursion is
 * safe in float32 (the device's precision); with the second-order terms (tau = 1, kept for the equivalence
 * test against the dense Newton direction) stage systems turn indefinite far from the minimiser, pivots
 * get replaced and the forward sweep can blow up (control_steps 64: |d| ~ 1e53) -- for 1-3 % fewer
 * iterations (12.07 against 12.21 at control_steps 32, 7.01 / 7.06 at 8). */
static void orc_riccati_direction_disp(const orc_ctx* c, const double* u, const double* gs, const double* gt,
                                       orc_active* a, double* d) {
  orc_riccati_direction_disp_tau(c, u, gs, gt, a, d, 0.0);
}
static void orc_riccati_direction_disp_tau(const orc_ctx* c, const double* u, const double* gs, const double* gt,
                                       orc_active* a, double* d, double tau) {
  const int n = c->n;
  const double dt = c->dt;
  double cs[ORC_MAXN], sn[ORC_MAXN], px[ORC_MAXN], py[ORC_MAXN], SX[ORC_MAXN], SY[ORC_MAXN], xs_[ORC_MAXN], ys_[ORC_MAXN];
  {
    double x = 0.0, y = 0.0, th = 0.0, rx[ORC_MAXN], ry[ORC_MAXN];
    for (int i = 0; i < n; ++i) {
      th += u[3 * i + 2] * dt;
      cs[i] = cos(th); sn[i] = sin(th);
      px[i] = (u[3 * i] * cs[i] - u[3 * i + 1] * sn[i]) * dt;
      py[i] = (u[3 * i] * sn[i] + u[3 * i + 1] * cs[i]) * dt;
      x += px[i]; y += py[i];
      xs_[i] = x; ys_[i] = y;
      rx[i] = -2.0 * c->wt_n * (c->cx - x);
      ry[i] = -2.0 * c->wt_n * (c->cy - y);
    }
    double ax = 0.0, ay = 0.0;
    for (int i = n - 1; i >= 0; --i) { ax += rx[i]; ay += ry[i]; SX[i] = ax; SY[i] = ay; }
  }
  double vv[3] = {c->v[0], c->v[1], c->v[2]}, vp[3] = {c->v[0], c->v[1], c->v[2]};
  orc_project(c, vp);
  const int v_feasible = vp[0] == vv[0] && vp[1] == vv[1] && vp[2] == vv[2];
  static _Thread_local double Kf[ORC_MAXN][3][3], kf[ORC_MAXN][3];
  double V00 = 0, V01 = 0, V02 = 0, V11 = 0, V12 = 0, V22 = 0, v0 = 0, v1 = 0, v2 = 0;
  const double w2 = 2.0 * c->wt_n, wc2 = (c->wc_n * c->wc_n) / (dt * dt), idt = 1.0 / dt;
  for (int i = n - 1; i >= 0; --i) {
    /* wall model (orc_wall_model) */
    double Ww[3], lw[2];
    orc_wall_model(c, xs_[i], ys_[i], Ww, lw);
    const double wxx = Ww[0], wxy = Ww[1], wyy = Ww[2];
    v0 += lw[0]; v1 += lw[1];
    const double S00 = V00 + w2 + wxx, S01 = V01 + wxy, S11 = V11 + w2 + wyy;
    const double S22 = V22 + 2.0 * c->wo_n + (i == n - 1 ? 2.0 * c->wterm_o : 0.0);
    const double M02 = -py[i] * S00 + px[i] * S01 + V02, M12 = -py[i] * S01 + px[i] * S11 + V12;
    const double M22 = -py[i] * M02 + px[i] * M12 + (-py[i] * V02 + px[i] * V12 + S22);
    const double kap = -tau * (SX[i] * px[i] + SY[i] * py[i]);
    const double Z02 = M02 + tau * SY[i], Z12 = M12 - tau * SX[i], Z22 = M22 + kap;
    const double z0 = v0, z1 = v1, z2 = -py[i] * v0 + px[i] * v1 + v2;
    /* gradients and the step onto the kink in displacement coordinates */
    const double* ui = u + 3 * i;
    const double gt0 = (cs[i] * gt[3 * i] - sn[i] * gt[3 * i + 1]) * idt, gt1 = (sn[i] * gt[3 * i] + cs[i] * gt[3 * i + 1]) * idt,
                 gt2 = gt[3 * i + 2] * idt;
    const double gs0 = (cs[i] * gs[3 * i] - sn[i] * gs[3 * i + 1]) * idt, gs1 = (sn[i] * gs[3 * i] + cs[i] * gs[3 * i + 1]) * idt,
                 gs2 = gs[3 * i + 2] * idt;
    const double e0 = ui[0] - c->v[0], e1 = ui[1] - c->v[1], e2 = ui[2] - c->v[2];
    const double f0 = cs[i] * e0 - sn[i] * e1, f1 = sn[i] * e0 + cs[i] * e1, f2 = e2;
    double k[3] = {0, 0, 0}, K[3][3] = {{0}};
    a->tokink[i] = 0;
    if (!a->near[i] && v_feasible && orc_kink_predict) {
      const double w0 = -f0 * dt, w1 = -f1 * dt, w2k = -f2 * dt;
      const double r0 = gs0 + z0 + S00 * w0 + S01 * w1 + Z02 * w2k, r1 = gs1 + z1 + S01 * w0 + S11 * w1 + Z12 * w2k,
                   r2 = gs2 + z2 + Z02 * w0 + Z12 * w1 + Z22 * w2k;
      if (r0 * r0 + r1 * r1 + r2 * r2 <= wc2) { a->tokink[i] = 1; k[0] = w0; k[1] = w1; k[2] = w2k; }
    }
    if (!a->tokink[i]) {
      double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0, tx = 0, ty = 0;
      if (!a->near[i]) {
        const double fn = sqrt(f0 * f0 + f1 * f1 + f2 * f2), ine = fn > 0.0 ? 1.0 / fn : 0.0;
        const double sN = c->wc_n * ine * idt * idt, h0 = f0 * ine, h1 = f1 * ine, h2 = f2 * ine;
        const double utx = -a->ny[i], uty = a->nx[i];
        tx = cs[i] * utx - sn[i] * uty; ty = sn[i] * utx + cs[i] * uty;
        const double k2 = (a->mode[i] == 1 && a->disc[i]) ? a->lambda[i] / c->r * idt * idt : 0.0;
        c00 = sN * (1 - h0 * h0) + k2 * tx * tx; c01 = -sN * h0 * h1 + k2 * tx * ty; c02 = -sN * h0 * h2;
        c11 = sN * (1 - h1 * h1) + k2 * ty * ty; c12 = -sN * h1 * h2; c22 = sN * (1 - h2 * h2);
        /* Levenberg-Marquardt damping, in units of one stage's tracking weights (orc_pg_solve adapts it) */
        c00 += orc_mu * w2; c11 += orc_mu * w2; c22 += orc_mu * 2.0 * c->wo_n;
      }
      const double Q00 = S00 + c00, Q01 = S01 + c01, Q02 = Z02 + c02, Q11 = S11 + c11, Q12 = Z12 + c12, Q22 = Z22 + c22;
      const double q0 = gt0 + z0, q1 = gt1 + z1, q2 = gt2 + z2;
      const int wfree = !(a->near[i] || a->wfroz[i]);
      const int xy = a->near[i] ? 2 : a->mode[i];
      /* Quz rows: (S00 S01 Z02), (S01 S11 Z12), (M02 M12 Z22) */
      const double Zr[3][3] = {{S00, S01, Z02}, {S01, S11, Z12}, {M02, M12, Z22}};
      if (xy == 0 && wfree) {
        double Q[3][3] = {{Q00, Q01, Q02}, {Q01, Q11, Q12}, {Q02, Q12, Q22}}, Lm[3][3];
        orc_sym3_solve_prepare(Q, Lm, fmax(1e-6 * fmax(fabs(Q00), fmax(fabs(Q11), fabs(Q22))), 1e-30));
        double rhs[3] = {-q0, -q1, -q2};
        orc_sym3_solve(Lm, rhs, k);
        for (int q = 0; q < 3; ++q) {
          double col[3] = {-Zr[0][q], -Zr[1][q], -Zr[2][q]}, sol[3];
          orc_sym3_solve(Lm, col, sol);
          for (int r = 0; r < 3; ++r) K[r][q] = sol[r];
        }
      } else {
        double ax = 0, ay = 0;
        int has_a = 0, has_b = 0, b_is_w = 0;
        if (xy == 1) { ax = tx; ay = ty; has_a = 1; has_b = wfree; b_is_w = 1; }
        else if (xy == 0) { ax = 1.0; has_a = 1; has_b = 1; b_is_w = 0; }
        else { has_b = wfree; b_is_w = 1; }
        const double Qa0 = ax * Q00 + ay * Q01, Qa1 = ax * Q01 + ay * Q11, Qa2 = ax * Q02 + ay * Q12;
        const double haa = Qa0 * ax + Qa1 * ay, hab = b_is_w ? Qa2 : Qa1, hbb = b_is_w ? Q22 : Q11;
        const double ga = ax * q0 + ay * q1, gb = b_is_w ? q2 : q1;
        double Za[3], Zb[3], ka = 0, kb = 0, Ka[3] = {0, 0, 0}, Kb[3] = {0, 0, 0};
        for (int q = 0; q < 3; ++q) { Za[q] = ax * Zr[0][q] + ay * Zr[1][q]; Zb[q] = b_is_w ? Zr[2][q] : Zr[1][q]; }
        if (has_a && has_b) {
          const double delta = fmax(1e-6 * fmax(fabs(haa), fabs(hbb)), 1e-30);
          const double d0 = orc_piv(haa, delta), l = hab / d0, d1 = orc_piv(hbb - l * hab, delta);
          { const double y0 = -ga, y1 = -gb - l * y0; kb = y1 / d1; ka = y0 / d0 - l * kb; }
          for (int q = 0; q < 3; ++q) { const double y0 = -Za[q], y1 = -Zb[q] - l * y0; Kb[q] = y1 / d1; Ka[q] = y0 / d0 - l * Kb[q]; }
        } else if (has_a) {
          const double i0 = -1.0 / orc_piv(haa, fmax(1e-6 * fabs(haa), 1e-30));
          ka = ga * i0; for (int q = 0; q < 3; ++q) Ka[q] = Za[q] * i0;
        } else if (has_b) {
          const double i0 = -1.0 / orc_piv(hbb, fmax(1e-6 * fabs(hbb), 1e-30));
          kb = gb * i0; for (int q = 0; q < 3; ++q) Kb[q] = Zb[q] * i0;
        }
        k[0] = ax * ka; k[1] = ay * ka;
        for (int q = 0; q < 3; ++q) { K[0][q] = ax * Ka[q]; K[1][q] = ay * Ka[q]; }
        if (b_is_w) { k[2] = kb; for (int q = 0; q < 3; ++q) K[2][q] = Kb[q]; }
        else { k[1] += kb; for (int q = 0; q < 3; ++q) K[1][q] += Kb[q]; }
      }
      v0 = z0 + S00 * k[0] + S01 * k[1] + M02 * k[2];
      v1 = z1 + S01 * k[0] + S11 * k[1] + M12 * k[2];
      v2 = z2 + Z02 * k[0] + Z12 * k[1] + Z22 * k[2];
      V00 = S00 + S00 * K[0][0] + S01 * K[1][0] + M02 * K[2][0];
      V01 = S01 + S00 * K[0][1] + S01 * K[1][1] + M02 * K[2][1];
      V02 = M02 + S00 * K[0][2] + S01 * K[1][2] + M02 * K[2][2];
      V11 = S11 + S01 * K[0][1] + S11 * K[1][1] + M12 * K[2][1];
      V12 = M12 + S01 * K[0][2] + S11 * K[1][2] + M12 * K[2][2];
      V22 = Z22 + Z02 * K[0][2] + Z12 * K[1][2] + Z22 * K[2][2];
    } else {
      v0 = z0 + S00 * k[0] + S01 * k[1] + M02 * k[2];
      v1 = z1 + S01 * k[0] + S11 * k[1] + M12 * k[2];
      v2 = z2 + Z02 * k[0] + Z12 * k[1] + Z22 * k[2];
      V00 = S00; V01 = S01; V02 = M02; V11 = S11; V12 = M12; V22 = Z22;
    }
    for (int r = 0; r < 3; ++r) { kf[i][r] = k[r]; for (int q = 0; q < 3; ++q) Kf[i][r][q] = K[r][q]; }
  }
  double z0 = 0, z1 = 0, z2 = 0;
  for (int i = 0; i < n; ++i) {
    const double w0 = kf[i][0] + Kf[i][0][0] * z0 + Kf[i][0][1] * z1 + Kf[i][0][2] * z2;
    const double w1 = kf[i][1] + Kf[i][1][0] * z0 + Kf[i][1][1] * z1 + Kf[i][1][2] * z2;
    const double w2f = kf[i][2] + Kf[i][2][0] * z0 + Kf[i][2][1] * z1 + Kf[i][2][2] * z2;
    if (a->tokink[i]) { for (int r = 0; r < 3; ++r) d[3 * i + r] = c->v[r] - u[3 * i + r]; }
    else { d[3 * i] = (cs[i] * w0 + sn[i] * w1) * idt; d[3 * i + 1] = (-sn[i] * w0 + cs[i] * w1) * idt; d[3 * i + 2] = w2f * idt; }
    const double e0 = z0 + w0, e1 = z1 + w1, e2 = z2 + w2f;
    z0 = e0 - py[i] * e2; z1 = e1 + px[i] * e2; z2 = e2;
  }
}

/* debug/test hook: both Newton directions at a feasible point u (the dense one in float64) */
void orc_debug_newton_directions(const neo_mpc_params* p, const uint8_t* cells, int32_t sx, int32_t sy, double res,
                                 double ox, double oy, const neo_mpc_problem* q, const double* u, double* d_dense,
                                 double* d_stage, double* d_disp);

static double orc_dot(const double* a, const double* b, int n) {
  double s = 0.0;
  for (int k = 0; k < n; ++k) s += a[k] * b[k];
  return s;
}

/* candidate step multipliers: lanes 0..31 scale the proximal-gradient step alpha by
 * 2^(-12 + l/2); lanes 32..63 are step lengths along the L-BFGS direction. */
static const double ORC_QN_T[32] = {
    1.0, 0.84, 1.19, 0.71, 1.41, 0.59, 1.68, 0.5, 2.0, 0.42, 2.38, 0.35, 2.83, 0.25, 4.0, 0.177,
    0.125, 0.088, 0.0625, 0.044, 0.03125, 0.0156, 0.0078, 0.0039, 0.00195, 0.00098, 4.9e-4, 2.4e-4,
    1.2e-4, 6e-5, 3e-5, 1.5e-5};

/* long: the Newton directions trade the three shortest step lengths (L-BFGS's last resort) for long shots,
 * 8, 16 and 32 -- they find the way out of a lethal cell (measured on 8192 cold starts at control_steps 32:
 * 39 objectives more than 1 % better against 31 worse; same iteration count) */
static double orc_lane_scale(int lane, int longshots) {
  if (longshots && lane >= 61) return ldexp(1.0, lane - 58);
  if (lane >= 32) return ORC_QN_T[lane - 32];
  double s = ldexp(1.0, -12 + (lane >> 1));
  return (lane & 1) ? s * 1.4142135623730951 : s;
}

/* (round 4) A block sliding along a box bound stops where the bound meets the speed disc: the Euclidean projection of a
 * point beyond that corner slides DOWN the disc, away from the bound, so a Newton step along the bound was cut to the
 * fraction that reaches the corner and every other block's step with it (held-out set "a", control_steps 12: searches
 * jammed at the corner for 20 iterations).  A/B hook: orc_set_corner_stop(0). */
static int orc_final_confirm = 1;
void orc_set_final_confirm(int m) { orc_final_confirm = m; }
static int orc_closing_need = NEO_RULE_CLOSING_RUN;
void orc_set_closing_need(int m) { orc_closing_need = m; }
static int orc_repin = 1;
void orc_set_repin(int m) { orc_repin = m; }
static long orc_repin_count = 0;
long orc_get_repin_count(void) { return orc_repin_count; }
static int orc_corner_stop = 1;
void orc_set_corner_stop(int m) { orc_corner_stop = m; }
static int orc_land_mode = 0;
void orc_set_land_mode(int m) { orc_land_mode = m; }
static int orc_near_mode = 1;
void orc_set_near_mode(int m) { orc_near_mode = m; }
static void orc_candidate(const orc_ctx* c, const orc_active* act, int lane, double alpha, const double* u,
                          const double* gs, const double* d, double* cand) {
  const double sc = orc_lane_scale(lane, act->longshots);
  for (int i = 0; i < c->n; ++i) {
    double b[3];
    if ((orc_near_mode == 1 || orc_near_mode == 3) && lane >= 32 && (lane & 1) && act->near[i]) {
      for (int k = 0; k < 3; ++k) b[k] = u[3 * i + k];
    } else if (orc_near_mode == 2 && lane >= 32 && act->near[i]) {
      for (int k = 0; k < 3; ++k) b[k] = u[3 * i + k];
    } else if (lane < 32 || act->near[i]) {
      /* per-block step: the curvature of block i's own tracking terms is proportional to the number
       * of stages it still moves, N - i (diagonal of the Gauss-Newton Hessian: 2 w_trans/N dt^2 (N - i)) */
      double a = alpha * sc * (act->riccati ? (double)c->n / (double)(c->n - i) : 1.0);
      if (orc_near_mode == 3 && lane >= 32) {
        /* the block's own curvature bound: 2 dt^2 ((N - i)/N (max(w_trans, w_orient) + w_trans (r H)^2) + w_terminal w_orient) */
        const double wt = c->wt_n * c->n, wo = c->wo_n * c->n, reach = c->r * c->dt * c->n;
        const double L = 2.0 * c->dt * c->dt * ((double)(c->n - i) / c->n * (fmax(wt, wo) + wt * reach * reach) + c->wterm_o);
        a = 1.0 / L;
      }
      double e[3], ne2 = 0.0;
      for (int k = 0; k < 3; ++k) { e[k] = (u[3 * i + k] - a * gs[3 * i + k]) - c->v[k]; ne2 += e[k] * e[k]; }
      double ne = sqrt(ne2);
      double sh = (ne > 0.0) ? fmax(0.0, 1.0 - a * c->wc_n / ne) : 0.0;
      for (int k = 0; k < 3; ++k) b[k] = c->v[k] + sh * e[k];
    } else {
      const double t = act->tokink[i] ? fmin(sc, 1.0) : sc;
      for (int k = 0; k < 3; ++k) b[k] = u[3 * i + k] + t * d[3 * i + k];
      if (act->tokink[i] && t == 1.0) for (int k = 0; k < 3; ++k) b[k] = c->v[k];
      if (orc_land_mode && !act->riccati && !act->tokink[i]) {
        /* a Newton step that carries the block THROUGH the kink (radially inward by more than its distance) ends on it */
        double e[3] = {u[3 * i] - c->v[0], u[3 * i + 1] - c->v[1], u[3 * i + 2] - c->v[2]};
        const double rho2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
        const double inward = -(t * (d[3 * i] * e[0] + d[3 * i + 1] * e[1] + d[3 * i + 2] * e[2]));
        if (rho2 > 0.0 && inward >= rho2) for (int k = 0; k < 3; ++k) b[k] = c->v[k];
      }
      const double rl = c->r * (1.0 - 1e-12);
      if (orc_corner_stop && act->mode[i] == 1 && !act->disc[i] && !act->tokink[i] &&
          u[3 * i] * u[3 * i] + u[3 * i + 1] * u[3 * i + 1] < rl * rl) {   /* (not AT the corner already: from there the projection slides it along the disc) */
        /* a block sliding along a box bound stops where the bound meets the speed disc (the Euclidean projection of a
         * point beyond the corner slides down the disc instead, away from the bound) */
        const int free_axis = act->nx[i] != 0.0 ? 1 : 0, fixed_axis = 1 - free_axis;
        const double fixed = u[3 * i + fixed_axis];
        const double lim2 = c->r * c->r - fixed * fixed, lim = lim2 > 0.0 ? sqrt(lim2) * (1.0 - 1e-15) : 0.0;
        b[fixed_axis] = fixed;
        b[free_axis] = orc_clamp(b[free_axis], -lim, lim);
      }
    }
    orc_project(c, b);
    for (int k = 0; k < 3; ++k) cand[3 * i + k] = b[k];
  }
}

/* Hop candidates of the stage-wise direction (mirror of costmap.h edge_hop + feasible_set.h).  The costmap term is
 * piecewise constant: a stage that sits within ORC_HOP_DIST cells of a cell edge behind which the term is LOWER can
 * gain that step for next to nothing -- a displacement of a few millimetres -- but no descent direction says so (the
 * term has no gradient), and at a heavy costmap weight one such step is worth more than the whole 1e-3 budget
 * (found with the G9 fixtures: the node's own defaults, every weight 0.5; SLSQP's line search lands across such
 * edges by chance).  Lanes 1..ORC_HOP_LANES of the search therefore try: the current point with ONE block changed so
 * that its stage lands ORC_HOP_MARGIN cells inside the cheaper neighbour cell (all later stages shift with it).
 * hop[i] = the change of block i's (vx, vy), has[i] = stage i has a cheaper neighbour in range. */
#define ORC_HOP_DIST NEO_RULE_HOP_DIST
#define ORC_HOP_MARGIN NEO_RULE_HOP_MARGIN
#define ORC_HOP_MAX_DV NEO_RULE_HOP_MAX_DV
#define ORC_HOP_LANES NEO_RULE_HOP_LANES
static int orc_hops_on = 1;
void orc_set_hops(int on) { orc_hops_on = on; }
static int orc_hops(const orc_ctx* c, const double* u, double min_drop, double hop[][2], uint8_t* has) {
  const orc_map* m = c->map;
  const double range = fmin(ORC_HOP_DIST, ORC_HOP_MAX_DV * c->dt / m->resolution);
  double x = 0.0, y = 0.0, th = 0.0;
  int any = 0;
  for (int i = 0; i < c->n; ++i) {
    th += u[3 * i + 2] * c->dt;
    const double cs = cos(th), sn = sin(th);
    x += (u[3 * i] * cs - u[3 * i + 1] * sn) * c->dt;
    y += (u[3 * i] * sn + u[3 * i + 1] * cs) * c->dt;
    const double X = c->X0 + (c->c0 * x - c->s0 * y), Y = c->Y0 + (c->s0 * x + c->c0 * y);
    int64_t mx, my;
    orc_world_to_map(m, X, Y, &mx, &my);
    const double fx = (X - m->origin_x) * (1.0 / m->resolution) - (double)mx, fy = (Y - m->origin_y) * (1.0 / m->resolution) - (double)my;
    const double here = orc_term_at(c, mx, my);
    const int64_t nbx[4] = {mx - 1, mx + 1, mx, mx}, nby[4] = {my, my, my - 1, my + 1};
    const double dist[4] = {fx, 1.0 - fx, fy, 1.0 - fy}, sign[4] = {-1.0, 1.0, -1.0, 1.0};
    has[i] = 0; hop[i][0] = 0.0; hop[i][1] = 0.0;
    double best = min_drop;   /* (a step worth less than a tenth of opt_tolerance is not worth a hop) */
    for (int k = 0; k < 4; ++k) {
      const double drop = here - orc_term_at(c, nbx[k], nby[k]);
      if (!(dist[k] < range && drop > best)) continue;
      best = drop;
      const double len = sign[k] * (dist[k] + ORC_HOP_MARGIN) * m->resolution;   /* metres along the world axis */
      const double wx = k < 2 ? len : 0.0, wy = k < 2 ? 0.0 : len;
      const double rx = c->c0 * wx + c->s0 * wy, ry = -c->s0 * wx + c->c0 * wy;      /* rollout frame */
      hop[i][0] = (cs * rx + sn * ry) / c->dt; hop[i][1] = (-sn * rx + cs * ry) / c->dt;  /* block i's frame */
      has[i] = 1;
    }
    any |= has[i];
  }
  return any;
}

/* (round 5) Cell scan -- mirror of csrc/cell_scan.h.  The costmap term is piecewise constant: no gradient, and the hop
 * candidates look one cell edge and a quarter of a cell ahead; SLSQP's line search samples the term cells away from its
 * iterate (py:246-260 through py:363-364) and now and then lands in a cheaper cell one to three cells from where a descent
 * method ends (random parameter sets against the reference: 12 of the 14 objective misses in 5568 costmap cases).  A search
 * that has ENDED therefore looks once at the cells around every stage: "lane" L of 64 takes stage L / lps (lps = 64 /
 * control_steps lanes per stage) and every lps-th of the 48 cells within NEO_RULE_SCAN_CELLS cells of that stage's cell --
 * none further than the reach tile's radius from the robot's own cell --, keeps the one with the best estimate among those
 * whose term is lower by more than min_drop (term drop minus the tracking cost of moving that stage alone), and evaluates
 * the current point with the stage displaced to land ORC_HOP_MARGIN cells inside that cell, exactly, twice: (A) block i
 * changed alone (every later stage shifts with it), (B) block i changed and block i + 1 changed back (only stage i moves).
 * Candidates compete by objective value; ties go to the lowest lane, A before B.  Returns 1 when one lowered f (u, *f
 * updated).  A/B hook: orc_set_scan(0) switches it off. */
static int orc_scan_on = 1;
void orc_set_scan(int on) { orc_scan_on = on; }
static long orc_scan_calls = 0, orc_scan_cands = 0, orc_scan_wins = 0;   /* (development statistics; not exact under OpenMP) */
long orc_get_scan_stat(int k) { return k == 0 ? orc_scan_calls : k == 1 ? orc_scan_cands : orc_scan_wins; }
static int orc_cell_scan(const orc_ctx* c, int reach, double* u, double* f, double min_drop, int* nfev) {
  const orc_map* m = c->map;
  const int n = c->n, nv = 3 * n, R = NEO_RULE_SCAN_CELLS, W = 2 * R + 1, ncell = W * W - 1;
  double px[ORC_MAXN], py[ORC_MAXN], pcs[ORC_MAXN], psn[ORC_MAXN];
  double x = 0.0, y = 0.0, th = 0.0;
  for (int i = 0; i < n; ++i) {
    th += u[3 * i + 2] * c->dt;
    pcs[i] = cos(th); psn[i] = sin(th);
    x += (u[3 * i] * pcs[i] - u[3 * i + 1] * psn[i]) * c->dt;
    y += (u[3 * i] * psn[i] + u[3 * i + 1] * pcs[i]) * c->dt;
    px[i] = x; py[i] = y;
  }
  int64_t mx0, my0;
  orc_world_to_map(m, c->X0, c->Y0, &mx0, &my0);
  double fbest = *f, best_u[ORC_MAXV], cand[ORC_MAXV];
  int won = 0;
  ++orc_scan_calls;
  *nfev += 2;
  const int lps = n < ORC_LANES ? ORC_LANES / n : 1;
  const double idt = 1.0 / c->dt;
  for (int lane = 0; lane < ORC_LANES; ++lane) {
    const int i = lane / lps, sct = lane % lps;
    if (i >= n) continue;
    const double X = c->X0 + (c->c0 * px[i] - c->s0 * py[i]), Y = c->Y0 + (c->s0 * px[i] + c->c0 * py[i]);
    int64_t mx, my;
    orc_world_to_map(m, X, Y, &mx, &my);
    const double fx = (X - m->origin_x) * (1.0 / m->resolution) - (double)mx, fy = (Y - m->origin_y) * (1.0 / m->resolution) - (double)my;
    const double here = orc_term_at(c, mx, my);
    const double ex = c->cx - px[i], ey = c->cy - py[i];
    double best_score = 0.0, brx = 0.0, bry = 0.0;
    int have = 0;
    for (int cc = sct; cc < ncell; cc += lps) {
      const int c2 = cc < ncell / 2 ? cc : cc + 1, dy = c2 / W - R, dx = c2 - (dy + R) * W - R;
      if (llabs(mx + dx - mx0) > reach || llabs(my + dy - my0) > reach) continue;
      const double there = orc_term_at(c, mx + dx, my + dy);
      if (!(here - there > min_drop)) continue;
      /* the nearest point of that cell, ORC_HOP_MARGIN cells inside it (cells, relative to the stage) */
      const double gx = dx < 0 ? (double)(dx + 1) - fx - ORC_HOP_MARGIN : dx > 0 ? (double)dx - fx + ORC_HOP_MARGIN : 0.0;
      const double gy = dy < 0 ? (double)(dy + 1) - fy - ORC_HOP_MARGIN : dy > 0 ? (double)dy - fy + ORC_HOP_MARGIN : 0.0;
      const double wx = gx * m->resolution, wy = gy * m->resolution;
      const double rx = c->c0 * wx + c->s0 * wy, ry = -c->s0 * wx + c->c0 * wy;      /* rollout frame */
      const double score = (here - there) - c->wt_n * (rx * rx + ry * ry - 2.0 * (rx * ex + ry * ey));
      if (!have || score > best_score) { have = 1; best_score = score; brx = rx; bry = ry; }
    }
    if (!have) continue;
    for (int type = 0; type < 2; ++type) {
      if (type == 1 && i == n - 1) continue;
      memcpy(cand, u, sizeof(double) * nv);
      double b[3] = {u[3 * i] + (pcs[i] * brx + psn[i] * bry) * idt, u[3 * i + 1] + (-psn[i] * brx + pcs[i] * bry) * idt, u[3 * i + 2]};
      orc_project(c, b);
      cand[3 * i] = b[0]; cand[3 * i + 1] = b[1];
      if (type == 1) {
        const int j = i + 1;
        double b2[3] = {u[3 * j] - (pcs[j] * brx + psn[j] * bry) * idt, u[3 * j + 1] - (-psn[j] * brx + pcs[j] * bry) * idt, u[3 * j + 2]};
        orc_project(c, b2);
        cand[3 * j] = b2[0]; cand[3 * j + 1] = b2[1];
      }
      const double fc = orc_eval(c, cand);
      ++orc_scan_cands;
      if (fc < fbest) { fbest = fc; won = 1; memcpy(best_u, cand, sizeof(double) * nv); }
    }
  }
  if (won) { memcpy(u, best_u, sizeof(double) * nv); *f = fbest; ++orc_scan_wins; }
  return won;
}

/* test hook: the direction of one iteration of the next orc_pg_solve call (single-threaded use) */
static int orc_capture_it = -1;
static double* orc_capture_d = NULL;
void orc_capture_direction(int it, double* d_out) { orc_capture_it = it; orc_capture_d = d_out; }
#define ORC_TRIAL_RATIO NEO_RULE_TRIAL_RATIO
#define ORC_BLOCKED_STEP NEO_RULE_BLOCKED_STEP
#define ORC_LATE_ITERATION NEO_RULE_LATE_ITERATION
static int orc_trial = 1;
void orc_set_trial(int on) { orc_trial = on; }
#define ORC_ALT_LANE 5   /* = kAltLane (solver_context.h) */
static int orc_unshift = 1;
void orc_set_unshift(int on) { orc_unshift = on; }
/* sum of the costmap terms of the rollout of u: 0.0 exactly when every stage sits in a cell whose term is zero (what the
 * kernels read off the winner's rollout: rollout.h term_sum) */
static double orc_term_sum(const orc_ctx* c, const double* u) {
  double x = 0.0, y = 0.0, th = 0.0, ts = 0.0;
  for (int i = 0; i < c->n; ++i) {
    th += u[3 * i + 2] * c->dt;
    const double cs = cos(th), sn = sin(th);
    x += (u[3 * i] * cs - u[3 * i + 1] * sn) * c->dt;
    y += (u[3 * i] * sn + u[3 * i + 1] * cs) * c->dt;
    ts += orc_step_term(c, x, y);
  }
  return ts;
}
static int orc_rest_rule = 1;
void orc_set_rest_rule(int on) { orc_rest_rule = on; }
static int orc_trace = 0;
void orc_set_trace(int on) { orc_trace = on; }
/* Blocked-run stop rule (dense Newton direction).  ORC_BLOCKED_RUN consecutive iterations NOT won by a decent Newton
 * step -- a proximal lane, or a Newton step cut below ORC_BLOCKED_STEP -- that together gain less than
 * ORC_BLOCKED_TOL_MAP * opt_tolerance (ORC_BLOCKED_TOL_FREE * opt_tolerance when no stage of the rollout has a costmap
 * term under it) end the search: something the quadratic model does not see is in the way -- a costmap cell edge, or
 * blocks hovering next to the control norm's kink -- and the search advances 1e-6 of f per iteration.  (SLSQP stops
 * on ONE iteration gaining less than opt_tolerance.)  Absolute, not scaled by |f|: f is dominated by the lethal term
 * while a search is on its way out of a lethal cell.  Closed 30 Hz loop of 4096 robots (warm ticks, where such
 * searches set a launch's duration): per-tick maximum 25 -> 13 iterations in the median, 100 -> 16 at worst; 4096 cold
 * C2 solves: no objective more than 9e-6 higher; 8192 zero-map problems against solves run to the end: unchanged (max
 * 5.8e-4).  Part of the window rule: off with it.  The stage-wise direction does not take it (its wall model and hop
 * candidates deal with cell edges, and its long shots need their blocked iterations to get out of lethal cells).
 * Beyond 3 control steps (run-time-sized dense kernel) the thresholds shrink with (3/N)^2 like the stall threshold. */
#define ORC_BLOCKED_RUN NEO_RULE_BLOCKED_RUN
static int orc_blocked_rule = 1;
void orc_set_blocked_rule(int on) { orc_blocked_rule = on; }

/* Wall in reach: a LETHAL cell (raw 254) -- or the outside of the map, which reads lethal -- among the cells of the reach tile
 * as K1 stages it (rollout.h load_tile: rows [my0 - R, my0 + R], columns from (mx0 - R) & ~3, neo_rules_tile_width(R) of them).
 * No tile (a reach beyond NEO_RULE_MAX_TILE_REACH cells): the kernels cannot look, every instance counts as next to a wall. */
static int orc_wall_in_reach(const orc_ctx* c, int reach) {
  const orc_map* m = c->map;
  const int w = neo_rules_tile_width(reach);
  if (!w) return 1;
  int64_t mx0, my0;
  orc_world_to_map(m, c->X0, c->Y0, &mx0, &my0);
  const int64_t x0 = (mx0 - reach) & ~(int64_t)3, y0 = my0 - reach;
  if (x0 < 0 || y0 < 0 || x0 + w > m->size_x || y0 + 2 * reach + 1 > m->size_y) return 1;
  for (int64_t y = y0; y <= y0 + 2 * reach; ++y)
    for (int64_t x = x0; x < x0 + w; ++x)
      if (m->cells[y * (int64_t)m->size_x + x] == 254) return 1;
  return 0;
}
static int orc_route = 1;   /* A/B hook: 0 = round 5's AUTO (the dense direction for every instance at control_steps 3) */
void orc_set_route(int on) { orc_route = on; }

/* Returns status; x_out = minimiser estimate, *f_out its objective. */
/* (round 4) A/B hooks.  orc_tau_mode 1 (default): the stage-wise direction carries the second-order terms of the rollout
 * step (lambda . d2F: the exact Hessian, quadratic convergence); 0: Gauss-Newton (rounds 2-3: linear convergence wherever
 * the tracking residuals are large -- held-out parameter sets "a" and "c": first controls 1.4e-3 ... 3.2e-3 from SLSQP's
 * converged ones when the window rule cut in).  orc_rule_mode 1 (default): the gain thresholds are relative to the part
 * of the objective that depends on u (the constant terminal distance term, py:266, can be 20x that), and with the
 * stage-wise direction the three-iteration window and the closing-in rule only judge runs of BLOCKED iterations
 * (iterations won by a decent Newton step end through the Newton step test); 0: rounds 2-3. */
static double orc_final_frac = NEO_RULE_FINAL_FRAC_GN;
void orc_set_final_frac(double f) { orc_final_frac = f; }
static int orc_tau_mode = 1, orc_rule_mode = 1;
void orc_set_tau_mode(int m) { orc_tau_mode = m; }
void orc_set_rule_mode(int m) { orc_rule_mode = m; }
int orc_pg_solve(const neo_mpc_params* p, const orc_map* m, const neo_mpc_problem* q,
                 double footprint_cost, const double* x0, const double* prev_u0, double* x_out, double* f_out,
                 int32_t* nit_out, int32_t* nfev_out) {
  orc_ctx c;
  orc_ctx_init(&c, p, m, q, footprint_cost);
  const int n = c.n, nv = 3 * n;
  /* every tolerance and the choice of the search direction: solver_rules.h, the rule book shared with the device code
   * (neo_mpc_capi.cpp derive()) */
  neo_rules rules;
  neo_rules_derive(p, &rules);
  /* (round 6) direction by neighbourhood: AUTO at control_steps 3 sends an instance with a lethal cell in its reach tile to
   * the stage-wise direction -- with that direction's rules (solver_rules.h neo_rules_routes_by_neighbourhood) */
  int routed = 0;   /* the stage-wise direction with the control_steps-3 stop rules (window rule on every run, blocked-run rule,
                     * closing-in behind two blocked iterations): what ends a search depends on the horizon, not on how d is computed */
  if (orc_route && neo_rules_routes_by_neighbourhood(p) && orc_wall_in_reach(&c, neo_rules_reach_cells(p, m->resolution))) {
    neo_rules_derive_routed(p, &rules);
    c.kink_radius = rules.kink_radius;
    routed = 1;
  }
  const int max_it = rules.max_iterations;
  int mem = rules.lbfgs_memory;
  if (mem > NEO_MPC_MAX_LBFGS_MEMORY) mem = NEO_MPC_MAX_LBFGS_MEMORY;
  const double xtol = rules.xtol, stall_step = rules.stall_step;
  /* search direction of lanes 32-63: stage-wise (Riccati) Newton, dense Newton (control_steps <= 8) or L-BFGS */
  const int riccati = rules.direction == NEO_DIRECTION_STAGEWISE;
  const int newton = riccati || (rules.direction == NEO_DIRECTION_DENSE && 3 * p->control_steps <= ORC_NEWTON_MAXV);
  const double ftol = rules.ftol;

  double u[ORC_MAXV], gs[ORC_MAXV], gt[ORC_MAXV], gr[ORC_MAXV], d[ORC_MAXV];
  double u_prev[ORC_MAXV], gt_prev[ORC_MAXV], cand[ORC_MAXV], best_c[ORC_MAXV];
  static _Thread_local double S[NEO_MPC_MAX_LBFGS_MEMORY][ORC_MAXV], Y[NEO_MPC_MAX_LBFGS_MEMORY][ORC_MAXV];
  double rho[NEO_MPC_MAX_LBFGS_MEMORY];
  int npairs = 0, head = 0; /* ring: newest at (head-1) mod mem */
  orc_active act;
  memset(&act, 0, sizeof(act));
  act.riccati = riccati;
  act.longshots = newton;
  uint8_t near_prev[ORC_MAXN];
  memset(near_prev, 0, sizeof(near_prev));

  for (int i = 0; i < n; ++i) {
    double b[3] = {x0[3 * i], x0[3 * i + 1], x0[3 * i + 2]};
    orc_project(&c, b);
    u[3 * i] = b[0]; u[3 * i + 1] = b[1]; u[3 * i + 2] = b[2];
  }
  double f = orc_eval(&c, u);
  int cold = 1;
  for (int k = 0; k < nv; ++k) cold = cold && (u[k] == 0.0);
  int alt_lane = 0;
  double alt_u[ORC_MAXV];
  if (orc_unshift && !cold && n > 1) {
    /* The warm start is the previous solution shifted by a WHOLE control step (py:198-202: block i <- block i + 1, the
     * FILTERED first control last, py:366-367) although only one control interval -- an eighth of a step at 30 Hz and the
     * README's horizon -- has passed: the previous solution itself, i.e. the shift undone with the first block as the solver
     * left it (`prev_u0`: kept in the state record by the batch entry points; the filtered one without it), is usually much
     * closer to this tick's minimiser -- and IS the minimiser for a robot the collision latch has stopped.  In free space
     * (no costmap term under either rollout: one basin) the search starts from whichever of the two has the lower
     * objective; on the costmap it starts where the reference starts and the un-shifted point, when it has the lower
     * objective, is one CANDIDATE of the first iteration (lane ORC_ALT_LANE): starting from it outright ended one recorded
     * call of the reference 1.9e-3 above it (P3w).  The warm start handed BACK is the reference's shift as ever (K2).
     * Closed loop of 4096 robots: 6.2 -> 4.45 -> 3.6 iterations per warm tick at control_steps 3, per-tick maximum 13 -> 10. */
    double alt[ORC_MAXV];
    for (int i = 0; i < n; ++i) {
      const int src = (i + n - 1) % n;
      alt[3 * i] = u[3 * src]; alt[3 * i + 1] = u[3 * src + 1]; alt[3 * i + 2] = u[3 * src + 2];
    }
    if (prev_u0) {   /* the previous solution's own first block (the shift carries the FILTERED one, py:366-367) */
      double b[3] = {prev_u0[0], prev_u0[1], prev_u0[2]};
      orc_project(&c, b);
      alt[0] = b[0]; alt[1] = b[1]; alt[2] = b[2];
    }
    const double fa = orc_eval(&c, alt);
    if (fa < f && orc_term_sum(&c, u) == 0.0 && orc_term_sum(&c, alt) == 0.0) { memcpy(u, alt, sizeof(double) * nv); f = fa; }
    else if (fa < f) { alt_lane = 1; memcpy(alt_u, alt, sizeof(double) * nv); }
  }
  /* Long horizons (Riccati direction): the curvature of a block falls with 1/N^2, so the proximal step starts
   * longer (one iteration of growing it saved); and two neighbouring blocks can trade displacement at almost no
   * cost, so the Newton step is long along such valleys and leaves the region where the model holds (constraints,
   * the control norm's kink, costmap cells): Levenberg-Marquardt damping mu (in units of one stage's tracking
   * weights, added to the block curvature in displacement coordinates), relaxed x1/4 after an iteration won by
   * the (nearly) full Newton step, tightened x4 after one won by a proximal step or a short Newton step.
   * Measured on 8192 cold starts: 12.6 -> 9.0 iterations at control_steps 32, 20.2 -> 10.3 at 64, 9.1 -> 8.1 at
   * 16; no gain at 8 and below (no damping there). */
  double alpha = riccati ? fmax(1.0, n / 8.0) : 1.0;
  const double mu0 = (riccati && n > 8) ? (n - 8) / 8.0 : 0.0, mu_lo = mu0 / 16.0, mu_hi = 16.0 * mu0;
  double mu = mu0;
  int nfev = 1, it = 0, status = NEO_MPC_STATUS_MAX_ITER, stall = 0;
  /* three iterations in a row that together gain less than wtol end the search (Newton only by default) */
  const double wtol = rules.wtol, wtol_late = rules.wtol_late;
  double gain1 = INFINITY, gain2 = INFINITY;
  const double final_tol = rules.final_tol;
  int final = 0;
  int blocked_run = 0;   /* consecutive iterations not won by a decent Newton step */
  int exact_step = 0;    /* this iteration's stage-wise direction carries the second-order terms */
  int nblocked = 1;      /* consecutive iterations not won by a Newton step of at least half its length */
  int scanned = 0;   /* the cell scan has had its turn */
  it = 0;
resume_search:
  for (; it < max_it; ++it) {
    orc_grad_smooth(&c, u, gs);
    orc_reduce(&c, u, gs, gt, gr, &act);
    double hop[ORC_MAXN][2];
    uint8_t has_hop[ORC_MAXN];
    int hop_stage[ORC_HOP_LANES], nhops = 0;
    if (riccati && orc_hops_on && orc_hops(&c, u, rules.hop_min_drop, hop, has_hop))
      for (int i = 0; i < n && nhops < ORC_HOP_LANES; ++i) if (has_hop[i]) hop_stage[nhops++] = i;
    if (newton) {
      /* a cold start (x0 = 0, the reference's reset state py:359) is far from the minimiser and the
       * Newton step almost never wins there (9 % of the cases): lanes 32-63 walk the reduced
       * steepest-descent direction in that first iteration instead */
      if (it == 0 && cold) for (int k = 0; k < nv; ++k) d[k] = -gr[k];
      else if (riccati && orc_disp) {
        orc_mu = mu;
        /* second-order terms only behind an iteration won by a decent Newton step (the model held there): far from the
         * minimiser -- a search blocked by a wall, the first steps of a cold start -- the exact Hessian is indefinite
         * and the Gauss-Newton direction is the safer one */
        const double tau = orc_tau_mode == 1 ? (nblocked == 0 ? 1.0 : 0.0) : orc_tau_mode == 2 ? 1.0 : 0.0;
        exact_step = tau != 0.0;
        orc_riccati_direction_disp_tau(&c, u, gs, gt, &act, d, tau);
        orc_mu = 0.0;
      }
      else if (riccati) orc_riccati_direction(&c, u, gs, gt, &act, d);
      else orc_newton_direction(&c, u, gs, gr, &act, d);
      orc_apply_active(&c, &act, d);
      if (orc_repin && (riccati || orc_repin == 1) && !(it == 0 && cold) &&
          !(c.lo[0] <= -c.r && c.hi[0] >= c.r && c.lo[1] <= -c.r && c.hi[1] >= c.r) &&   /* (the box cuts the disc: there are corners) */
          (orc_repin == 3 || (riccati ? orc_free_path(&c, u) : (it > 0 && orc_term_sum(&c, u) == 0.0)))) {
        /* (round 4) one-sided slides: a block in a corner of the feasible set (two constraints active) that slides along
         * one of them can only slide AWAY from the other -- a Newton step that sends it the other way is stopped by the
         * projection while every other block takes the step that counted on it.  Such blocks are pinned and the direction
         * is computed once more (the active-set step of a QP solver, one round of it).  In free space only (no costmap term
         * under the iterate's rollout): next to a cost step the Newton model is off either way and what finds the way on is
         * the spread of the candidates -- pinning changed which basin one recorded episode of the reference ended in. */
        int redo = 0;
        for (int i = 0; i < n; ++i) {
          if (act.mode[i] != 1 || act.near[i] || act.tokink[i]) continue;
          /* the OTHER constraints of the block (it slides along the disc, or along the vx or the vy bound): does the step
           * approach one of them with no room left (NEO_RULE_CORNER_ROOM: the projection's rounding leaves a block 1e-16
           * inside a bound it sat on)? */
          const double u0 = u[3 * i], u1 = u[3 * i + 1], d0 = d[3 * i], d1 = d[3 * i + 1];
          const int on_x = !act.disc[i] && act.nx[i] != 0.0, on_y = !act.disc[i] && act.nx[i] == 0.0;
          int blocked = 0;
          if (!on_x) {
            const double rate = fabs(d0), room = d0 > 0.0 ? c.hi[0] - u0 : u0 - c.lo[0];
            blocked |= rate > 0.0 && room <= NEO_RULE_CORNER_ROOM;
          }
          if (!on_y) {
            const double rate = fabs(d1), room = d1 > 0.0 ? c.hi[1] - u1 : u1 - c.lo[1];
            blocked |= rate > 0.0 && room <= NEO_RULE_CORNER_ROOM;
          }
          if (!act.disc[i]) {
            const double rin = c.r - NEO_RULE_CORNER_ROOM;
            blocked |= d0 * u0 + d1 * u1 > 0.0 && u0 * u0 + u1 * u1 >= rin * rin;
          }
          if (blocked) {
            if (orc_trace) fprintf(stderr, "      repin block %d: d %.3e %.3e\n", i, d0, d1);
            act.mode[i] = 2; gr[3 * i] = 0.0; gr[3 * i + 1] = 0.0; redo = 1;
          }
        }
        if (redo) {
          ++orc_repin_count;
          if (riccati && orc_disp) { orc_mu = mu; orc_riccati_direction_disp_tau(&c, u, gs, gt, &act, d, exact_step ? 1.0 : 0.0); orc_mu = 0.0; }
          else if (riccati) orc_riccati_direction(&c, u, gs, gt, &act, d);
          else orc_newton_direction(&c, u, gs, gr, &act, d);
          orc_apply_active(&c, &act, d);
        }
      }
      if (it > 0) { /* the full Newton step is already below the step tolerance: u is the answer
                     * (the prox-moved blocks next to the kink are not covered by d) */
        double dm = 0.0;
        int anynear = 0;
        for (int k = 0; k < nv; ++k) dm = fmax(dm, fabs(d[k]));
        /* (blocks next to the kink are moved by the prox step, which d does not describe -- unless they are at
         * rest on it) */
        for (int i = 0; i < n; ++i) anynear |= act.near[i] && !(orc_rest_rule && act.rest[i]);
        /* (with a cheaper cell a hop away the search runs once more: its hop lanes decide) */
        if (dm < xtol && !anynear && nhops == 0) { status = NEO_MPC_STATUS_CONVERGED; goto exit_check; }
        /* a full Newton step below opt_tolerance (SLSQP's own step test) is the last one: it is
         * searched and taken like any other, but nothing re-checks the point it lands on (the
         * error left is of the order of the step squared) */
        /* (a Gauss-Newton step converges linearly: it has to be shorter to be the last) */
        if (dm < (riccati && !exact_step ? orc_final_frac : 1.0) * final_tol && !anynear) final = 1;
      }
    }
    if (!newton && it > 0) {
      double* s = S[head];
      double* y = Y[head];
      for (int k = 0; k < nv; ++k) {
        const int skip = act.near[k / 3] || near_prev[k / 3];
        s[k] = skip ? 0.0 : u[k] - u_prev[k];
        y[k] = skip ? 0.0 : gt[k] - gt_prev[k];
      }
      double sy = orc_dot(s, y, nv), ss = orc_dot(s, s, nv), yy = orc_dot(y, y, nv);
      if (ss > 0.0 && sy > 1e-10 * sqrt(ss * yy)) {
        rho[head] = 1.0 / sy;
        head = (head + 1) % mem;
        if (npairs < mem) ++npairs;
      }
    }
    /* two-loop recursion on the reduced gradient */
    if (!newton) {
      double al[NEO_MPC_MAX_LBFGS_MEMORY];
      for (int k = 0; k < nv; ++k) d[k] = gr[k];
      for (int j = 0; j < npairs; ++j) {
        int idx = (head - 1 - j + 2 * mem) % mem;
        al[j] = rho[idx] * orc_dot(S[idx], d, nv);
        for (int k = 0; k < nv; ++k) d[k] -= al[j] * Y[idx][k];
      }
      double gamma = 1.0;
      if (npairs > 0) {
        int idx = (head - 1 + mem) % mem;
        gamma = 1.0 / (rho[idx] * orc_dot(Y[idx], Y[idx], nv));
      }
      for (int k = 0; k < nv; ++k) d[k] *= gamma;
      for (int j = npairs - 1; j >= 0; --j) {
        int idx = (head - 1 - j + 2 * mem) % mem;
        double be = rho[idx] * orc_dot(Y[idx], d, nv);
        for (int k = 0; k < nv; ++k) d[k] += S[idx][k] * (al[j] - be);
      }
      for (int k = 0; k < nv; ++k) d[k] = -d[k];
      orc_apply_active(&c, &act, d);
    }
    if (orc_capture_it == it && orc_capture_d) memcpy(orc_capture_d, d, sizeof(double) * nv);
    /* 64 candidates, lowest objective wins (ties: lowest lane) */
    double fb = INFINITY, fb_qn = INFINITY;
    int best = -1, best_qn = -1;
    /* Riccati direction, rollout in free space (no costmap term at any stage: the objective is smooth up to the
     * control norm's kink): the full Newton step (lane 32's candidate) is tried on its own first -- one
     * objective evaluation, lane = stage on the device -- and taken without the 64-candidate search when it
     * achieves ORC_TRIAL_RATIO of the decrease the quadratic model promises (-1/2 g_r . step).  Measured on 8192
     * cold starts at control_steps 32: 71 % of such trials succeed, 3.9 searches saved per solve for 0.3
     * iterations more, same results.  (With a costmap term under the rollout the search's spread of candidates
     * is what steps over cost edges and out of lethal cells: trying the step alone there loses 0.1 % of the
     * solves to worse minima and one of the reference-anchored P3 cases.) */
    int trial_ok = 0;
    const int free_before = riccati ? orc_free_path(&c, u) : 0;
    if (riccati && orc_trial && it > 0 && free_before) {
      orc_candidate(&c, &act, 32, alpha, u, gs, d, cand);
      const double ft = orc_eval(&c, cand);
      double pred = 0.0;
      for (int k = 0; k < nv; ++k) pred -= 0.5 * gr[k] * (cand[k] - u[k]);
      if (ft < f && f - ft >= ORC_TRIAL_RATIO * pred) { trial_ok = 1; fb = ft; best = 32; memcpy(best_c, cand, sizeof(double) * nv); }
    }
    for (int lane = 0; lane < ORC_LANES && !trial_ok; ++lane) {
      orc_candidate(&c, &act, lane, alpha, u, gs, d, cand);
      if (it == 0 && lane == 0) memcpy(cand, u, sizeof(double) * nv); /* the kernel gets f(x0) from this lane */
      if (lane >= 1 && lane <= nhops) {   /* hop candidate: u with one block moved across a cheaper cell edge */
        const int i = hop_stage[lane - 1];
        memcpy(cand, u, sizeof(double) * nv);
        double b[3] = {u[3 * i] + hop[i][0], u[3 * i + 1] + hop[i][1], u[3 * i + 2]};
        orc_project(&c, b);
        cand[3 * i] = b[0]; cand[3 * i + 1] = b[1];
      }
      if (it == 0 && lane == ORC_ALT_LANE && alt_lane) memcpy(cand, alt_u, sizeof(double) * nv);
      double fc = orc_eval(&c, cand);
      if (fc < fb) { fb = fc; best = lane; memcpy(best_c, cand, sizeof(double) * nv); }
      if (lane >= 32 && fc < fb_qn) { fb_qn = fc; best_qn = lane; }
      if (orc_trace > 1 && it == orc_trace) {
        fprintf(stderr, "  lane %2d sc %.3e fc-f %.3e cand", lane, orc_lane_scale(lane, act.longshots), fc - f);
        for (int k = 0; k < nv; ++k) fprintf(stderr, " %.7f", cand[k] - u[k]);
        fprintf(stderr, "\n");
      }
    }
    if (orc_trace > 1 && it == orc_trace) {
      fprintf(stderr, "  u "); for (int k = 0; k < nv; ++k) fprintf(stderr, " %.7f", u[k]);
      fprintf(stderr, "\n  gs"); for (int k = 0; k < nv; ++k) fprintf(stderr, " %.3e", gs[k]);
      fprintf(stderr, "\n  gr"); for (int k = 0; k < nv; ++k) fprintf(stderr, " %.3e", gr[k]);
      fprintf(stderr, "\n  d "); for (int k = 0; k < nv; ++k) fprintf(stderr, " %.3e", d[k]);
      fprintf(stderr, "\n");
    }
    ++nfev;
    if (orc_trace) {
      double gn = 0.0;
      for (int k = 0; k < nv; ++k) gn = fmax(gn, fabs(gr[k]));
      int nact = 0, nnear = 0;
      for (int i = 0; i < n; ++i) { nact += act.mode[i] != 0 || act.wfroz[i]; nnear += act.near[i]; }
      fprintf(stderr, "it %3d f %.15g fb-f %.3e best %2d alpha %.3e |gr|inf %.3e npairs %d | qn best %2d df %.3e active %d near %d |",
              it, f, fb - f, best, alpha, gn, npairs, best_qn, fb_qn - f, nact, nnear);
      for (int i = 0; i < n; ++i) fprintf(stderr, " %d%s%s", act.mode[i], act.wfroz[i] ? "w" : "", act.near[i] ? "k" : "");
      fprintf(stderr, "\n");
      if (orc_trace > 2) { for (int i = 0; i < n; ++i) fprintf(stderr, "      u[%d] % .6f % .6f % .6f  d % .3e % .3e % .3e  gt % .3e % .3e % .3e\n", i, u[3*i], u[3*i+1], u[3*i+2], d[3*i], d[3*i+1], d[3*i+2], gt[3*i], gt[3*i+1], gt[3*i+2]); }
    }
    if (!(fb < f)) { status = NEO_MPC_STATUS_CONVERGED; ++it; goto exit_check; }
    double step = 0.0;
    for (int k = 0; k < nv; ++k) {
      double ad = fabs(best_c[k] - u[k]);
      if (ad > step) step = ad;
      u_prev[k] = u[k]; gt_prev[k] = gt[k]; u[k] = best_c[k];
    }
    memcpy(near_prev, act.near, sizeof(near_prev));
    const double decrease = f - fb;
    f = fb;
    blocked_run = (best < 32 || orc_lane_scale(best, act.longshots) < ORC_BLOCKED_STEP) ? blocked_run + 1 : 0;
    /* (a hop that won says nothing about step lengths: damping and proximal step stay as they are) */
    const int hop_won = (best >= 1 && best <= nhops) || (it == 0 && best == ORC_ALT_LANE && alt_lane);
    if (!(it == 0 && cold) && !hop_won) {   /* (an iteration that had a Newton direction) */
      if (best >= 32 && orc_lane_scale(best, act.longshots) >= 0.8) mu = fmax(0.25 * mu, mu_lo);
      else if (best < 32 || orc_lane_scale(best, act.longshots) < 0.3) mu = fmin(4.0 * mu, mu_hi);
    }
    if (best < 32 && !hop_won) {
      alpha *= orc_lane_scale(best, act.longshots);
      alpha = orc_clamp(alpha, 1e-6, 1e6);
    }
    /* iterations that gain next to nothing or barely move (creeping along a costmap cell edge, the
     * slow tail next to the control-norm kink) end the search once ORC_STALL_ITERATIONS of them
     * are in a row */
    /* (gain thresholds are relative to the u-dependent part of the objective: f without the constant terms) */
    const double fsc = orc_rule_mode >= 1 ? fmax(1.0, fabs(fb - c.konst)) : fmax(1.0, fabs(fb));
    stall = (decrease <= ftol * fsc || step <= stall_step) ? stall + 1 : 0;
    /* (from ORC_LATE_ITERATION on the window is the control_steps-3 one again: a long-horizon search that has run
     * twice its usual length is creeping, gaining 1e-8 of f per iteration up to the iteration cap -- a handful per
     * 65 536 solves, but a launch lasts as long as its slowest wave) */
    const double wnow = it >= ORC_LATE_ITERATION ? wtol_late : wtol;
    /* stage-wise direction: the window and closing-in rules only judge runs of BLOCKED iterations (none of the three won
     * by a Newton step of at least half its length); iterations won by the Newton step end through the step test */
    nblocked = (best < 32 || orc_lane_scale(best, act.longshots) < NEO_RULE_WINDOW_STEP || hop_won) ? nblocked + 1 : 0;
    const int creeping = wnow > 0.0 && decrease + gain1 + gain2 <= wnow * fsc && (orc_rule_mode < 1 || !riccati || routed || nblocked >= 3);
    /* ... and so does a step below stall_step whose gain halved twice in a row: the search is closing in on a
     * costmap cell edge (or the kink) geometrically, what is left to gain is less than the last gain (part of the
     * window rule: off with it).  -3 % iterations at control_steps 3 and 32, no command moves by 1e-3.
     * (round 5) In free space three real gains it takes (the INFINITY the two older ones start at used to pass for one), and the
     * geometric series they start has to be worth less than the stall threshold -- gain r / (1 - r) <= ftol f~ with r =
     * gain / gain1 --: the rule ended warm searches whose gains fell by a sixth per iteration with 3e-6 left to gain, which
     * along a flat direction (curvature 1) is 2.5e-3 in the first control. */
    /* (in free space -- where the first control is gated, not only the objective: dense direction: no costmap term under the
     * NEW iterate's rollout; stage-wise: under the rollout the iteration started from) */
    const int free_now = riccati ? free_before : (newton ? orc_term_sum(&c, u) == 0.0 : 0);
    const int closing_in = wtol > 0.0 && step <= stall_step && decrease <= 0.5 * gain1 && gain1 <= 0.5 * gain2 &&
                           (!free_now || (gain2 < INFINITY && decrease * decrease <= ftol * fsc * (gain1 - decrease))) && (orc_rule_mode < 1 || ((riccati && !routed) ? nblocked >= 3 : nblocked >= orc_closing_need));
    int blocked_stop = 0;
    if (newton && (!riccati || routed) && orc_blocked_rule && wtol > 0.0 && blocked_run >= ORC_BLOCKED_RUN)
      blocked_stop = decrease + gain1 + gain2 <= (orc_term_sum(&c, u) == 0.0 ? rules.btol_free : rules.btol_map);
    gain2 = gain1; gain1 = decrease;
    /* (round 4) the last-step rule rests on the Newton model having held: an iteration announced as the last but WON by a
     * proximal step or a short Newton step (the model was off: a bound about to become active, the kink) is not the last */
    if (orc_final_confirm && final && !(best >= 32 && orc_lane_scale(best, act.longshots) >= NEO_RULE_WINDOW_STEP)) final = 0;
    if (orc_trace) fprintf(stderr, "      rules: step %.3e (xtol %.1e stall_step %.1e) decrease %.3e stall %d creeping %d closing_in %d final %d blocked_stop %d (run %d) nblocked %d\n",
                           step, xtol, stall_step, decrease, stall, creeping, closing_in, final, blocked_stop, blocked_run, nblocked);
    if (step < xtol || stall >= ORC_STALL_ITERATIONS || creeping || closing_in || final || blocked_stop) { status = NEO_MPC_STATUS_CONVERGED; ++it; goto exit_check; }
    continue;
  exit_check:
    break;
  }
  /* (round 5) Second-order directions: a search that has ENDED looks once at the costmap cells around every stage
   * (orc_cell_scan; it takes the place of round 4's exit hop of the dense direction, whose candidates are among its own).
   * Skipped when no stage of the iterate has a costmap term under it.  Behind a scan that gained more than opt_tolerance the
   * search is taken up again: the other blocks have a new neighbour to adjust to. */
  if (orc_scan_on && newton && status == NEO_MPC_STATUS_CONVERGED && !scanned && orc_term_sum(&c, u) != 0.0) {
    scanned = 1;
    const double f_before = f;
    /* (round 6: NEO_RULE_SCAN_REPEATS scans in a row -- a scan that found a cheaper cell may be followed by another from the new
     * point; the rule book keeps it at one and says what more would buy and cost) */
    int won = orc_cell_scan(&c, neo_rules_reach_cells(p, m->resolution), u, &f, rules.hop_min_drop, &nfev);
    for (int k = 1; k < NEO_RULE_SCAN_REPEATS && won && orc_term_sum(&c, u) != 0.0; ++k)
      if (!orc_cell_scan(&c, neo_rules_reach_cells(p, m->resolution), u, &f, rules.hop_min_drop, &nfev)) break;
    if (won && f_before - f > rules.scan_resume_gain && it < max_it) {
      status = NEO_MPC_STATUS_MAX_ITER; stall = 0; final = 0; blocked_run = 0; nblocked = 1; gain1 = INFINITY; gain2 = INFINITY;
      goto resume_search;
    }
  }
  /* (a search taken up again behind a scan that runs into the iteration cap HAD converged, and the scan only improved its point) */
  if (scanned && status == NEO_MPC_STATUS_MAX_ITER) status = NEO_MPC_STATUS_CONVERGED;
  memcpy(x_out, u, sizeof(double) * nv);
  *f_out = f;
  if (nit_out) *nit_out = it;
  if (nfev_out) *nfev_out = nfev;
  return status;
}

/* NEO_MPC_FLAG_WALL_IN_REACH of a request (K1's set-up: a lethal cell in the reach tile) */
static int orc_wall_flag(const neo_mpc_params* p, const orc_map* m, const neo_mpc_problem* q) {
  orc_ctx c;
  orc_ctx_init(&c, p, m, q, 0.0);
  return orc_wall_in_reach(&c, neo_rules_reach_cells(p, m->resolution));
}

/* ------------------------------------------------------------------ batch entry points (ctypes) */
static orc_map orc_make_map(const uint8_t* cells, int32_t sx, int32_t sy, double res, double ox, double oy) {
  orc_map m = {cells, sx, sy, res, ox, oy};
  return m;
}

static double orc_batch_footprint(const orc_map* m, const neo_mpc_batch* b, size_t i) {
  if (b->footprints && b->footprint_points > 0)
    return orc_footprint_cost(m, b->footprints + i * 2 * b->footprint_points, (int)b->footprint_points);
  return b->problems[i].footprint_cost;
}

void orc_objective_batch(const neo_mpc_params* p, const uint8_t* cells, int32_t sx, int32_t sy,
                         double res, double ox, double oy, const neo_mpc_problem* probs,
                         const double* u, double* f_out, size_t count) {
  orc_map m = orc_make_map(cells, sx, sy, res, ox, oy);
  for (size_t i = 0; i < count; ++i)
    f_out[i] = orc_objective(p, &m, &probs[i], u + i * 3 * p->control_steps, probs[i].footprint_cost);
}

void orc_footprint_cost_batch(const uint8_t* cells, int32_t sx, int32_t sy, double res, double ox,
                              double oy, const double* pts, int npts, double* out, size_t count) {
  orc_map m = orc_make_map(cells, sx, sy, res, ox, oy);
  for (size_t i = 0; i < count; ++i) out[i] = orc_footprint_cost(&m, pts + i * 2 * npts, npts);
}

/* wrapper only (P5): batch->solution supplies x.x, success[] supplies x.success */
void orc_postprocess_batch(const neo_mpc_params* p, const uint8_t* cells, int32_t sx, int32_t sy,
                           double res, double ox, double oy, const neo_mpc_batch* b,
                           const int32_t* success) {
  orc_map m = orc_make_map(cells, sx, sy, res, ox, oy);
  const int nv = 3 * p->control_steps;
  for (size_t i = 0; i < b->count; ++i) {
    neo_mpc_command* out = &b->commands[i];
    if (b->problems[i].skip) {   /* (no request this tick: see orc_solve_batch) */
      memset(out, 0, sizeof(*out));
      out->flags = NEO_MPC_FLAG_SKIPPED;
      if (b->predicted_path) memset(b->predicted_path + i * nv, 0, sizeof(double) * nv);
      continue;
    }
    memset(out, 0, sizeof(*out));
    double* warm = b->warm_start + i * nv;
    if (orc_reset_if_new_goal(p, &b->problems[i], &b->states[i], warm)) out->flags |= NEO_MPC_FLAG_RESET;
    if (orc_wall_flag(p, &m, &b->problems[i])) out->flags |= NEO_MPC_FLAG_WALL_IN_REACH;
    double x[ORC_MAXV];
    memcpy(x, b->solution + i * nv, sizeof(double) * nv);
    double fc = orc_batch_footprint(&m, b, i);
    out->cost = orc_objective(p, &m, &b->problems[i], x, fc);
    const double raw_u0[3] = {x[0], x[1], x[2]};
    orc_postprocess(p, &m, &b->problems[i], &b->states[i], warm, x, success ? success[i] : 1, fc, out,
                    b->predicted_path ? b->predicted_path + i * nv : NULL);
    b->states[i].has_prev_u0 = success ? success[i] != 0 : 1;   /* (the build's own hint: mirror of K2) */
    for (int k = 0; k < 3; ++k) b->states[i].prev_u0[k] = raw_u0[k];
  }
}

/* number of OpenMP threads of orc_solve_batch (bench.py's cpu_mirror: the CPUs the process may actually use -- the
 * runtime's default is every hardware thread of the host, 256 on a box whose container owns 16) */
#ifdef _OPENMP
#include <omp.h>
void orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
#else
void orc_set_threads(int n) { (void)n; }
#endif

/* full path with the build's solver: reset -> pg_solve -> postprocess */
void orc_solve_batch(const neo_mpc_params* p, const uint8_t* cells, int32_t sx, int32_t sy,
                     double res, double ox, double oy, const neo_mpc_batch* b) {
  orc_map m = orc_make_map(cells, sx, sy, res, ox, oy);
  const int nv = 3 * p->control_steps;
#pragma omp parallel for schedule(dynamic, 16)
  for (long long ii = 0; ii < (long long)b->count; ++ii) {
    size_t i = (size_t)ii;
    neo_mpc_command* out = &b->commands[i];
    /* no request for this robot this tick (the plugin threw before its service call, cpp:234-236): the node's state does
     * not advance -- state and warm start untouched; zero twist with the flag, zero rows in the optional outputs */
    if (b->problems[i].skip) {
      memset(out, 0, sizeof(*out));
      out->flags = NEO_MPC_FLAG_SKIPPED;
      if (b->solution) memset(b->solution + i * nv, 0, sizeof(double) * nv);
      if (b->predicted_path) memset(b->predicted_path + i * nv, 0, sizeof(double) * nv);
      if (b->velocities) memset(b->velocities + 3 * i, 0, sizeof(double) * 3);
      continue;
    }
    memset(out, 0, sizeof(*out));
    double* warm = b->warm_start + i * nv;
    if (orc_reset_if_new_goal(p, &b->problems[i], &b->states[i], warm)) out->flags |= NEO_MPC_FLAG_RESET;
    if (orc_wall_flag(p, &m, &b->problems[i])) out->flags |= NEO_MPC_FLAG_WALL_IN_REACH;
    double fc = orc_batch_footprint(&m, b, i);
    double x[ORC_MAXV], f;
    neo_mpc_state* st = &b->states[i];
    const int has_prev = st->has_prev_u0 == 1 && !(out->flags & NEO_MPC_FLAG_RESET);
    out->status = orc_pg_solve(p, &m, &b->problems[i], fc, warm, has_prev ? st->prev_u0 : NULL, x, &f, &out->iterations, &out->evaluations);
    out->cost = f;
    if (b->solution) memcpy(b->solution + i * nv, x, sizeof(double) * nv);
    const double raw_u0[3] = {x[0], x[1], x[2]};
    orc_postprocess(p, &m, &b->problems[i], &b->states[i], warm, x, out->status == NEO_MPC_STATUS_CONVERGED,
                    fc, out, b->predicted_path ? b->predicted_path + i * nv : NULL);
    st->has_prev_u0 = out->status == NEO_MPC_STATUS_CONVERGED;   /* (the build's own hint: mirror of K2) */
    for (int k = 0; k < 3; ++k) st->prev_u0[k] = raw_u0[k];
    if (b->velocities) for (int k = 0; k < 3; ++k) b->velocities[3 * i + k] = out->vel[k];
  }
}


/* which instances AUTO at control_steps 3 sends to the stage-wise direction (1) and which to the dense one (0): the reach
 * tile test of orc_wall_in_reach on every request (whatever the parameters' control_steps: the test only reads the map) */
void orc_route_batch(const neo_mpc_params* p, const uint8_t* cells, int32_t sx, int32_t sy, double res, double ox, double oy,
                     const neo_mpc_problem* problems, int32_t* routed, size_t count) {
  orc_map m = orc_make_map(cells, sx, sy, res, ox, oy);
  const int reach = neo_rules_reach_cells(p, res);
  for (size_t i = 0; i < count; ++i) {
    orc_ctx c;
    orc_ctx_init(&c, p, &m, &problems[i], 0.0);
    routed[i] = orc_wall_in_reach(&c, reach);
  }
}

/* test hooks of the solver mirror: the total gradient the solver uses at a (projected) point u --
 * adjoint gradient of the smooth part + control-norm gradient -- and both Newton directions */
void orc_gradient_batch(const neo_mpc_params* p, const uint8_t* cells, int32_t sx, int32_t sy, double res,
                        double ox, double oy, const neo_mpc_problem* probs, const double* u, double* g_out,
                        size_t count) {
  orc_map m = orc_make_map(cells, sx, sy, res, ox, oy);
  const int nv = 3 * p->control_steps;
  for (size_t i = 0; i < count; ++i) {
    orc_ctx c;
    orc_ctx_init(&c, p, &m, &probs[i], probs[i].footprint_cost);
    c.kink_radius = 0.0; /* plain gradient: no block is treated as sitting on the kink */
    double gs[ORC_MAXV], gr[ORC_MAXV];
    orc_active act;
    orc_grad_smooth(&c, u + i * nv, gs);
    orc_reduce(&c, u + i * nv, gs, g_out + i * nv, gr, &act);
  }
}

void orc_debug_newton_directions(const neo_mpc_params* p, const uint8_t* cells, int32_t sx, int32_t sy, double res,
                                 double ox, double oy, const neo_mpc_problem* q, const double* u, double* d_dense,
                                 double* d_stage, double* d_disp) {
  orc_map m = orc_make_map(cells, sx, sy, res, ox, oy);
  orc_ctx c;
  orc_ctx_init(&c, p, &m, q, q->footprint_cost);
  double gs[ORC_MAXV], gt[ORC_MAXV], gr[ORC_MAXV];
  orc_active act;
  orc_grad_smooth(&c, u, gs);
  orc_reduce(&c, u, gs, gt, gr, &act);
  const int nv = 3 * c.n;
  if (d_dense && nv <= ORC_NEWTON_MAXV) {
    orc_newton_f64 = 1;
    orc_newton_direction(&c, u, gs, gr, &act, d_dense);
    orc_newton_f64 = 0;
    orc_apply_active(&c, &act, d_dense);
  }
  if (d_stage) {
    orc_kink_predict = 0;
    orc_riccati_direction(&c, u, gs, gt, &act, d_stage);
    orc_kink_predict = 1;
    orc_apply_active(&c, &act, d_stage);
  }
  if (d_disp) {   /* the device's formulation: displacement coordinates, face-reduced stage solves */
    orc_kink_predict = 0;
    orc_riccati_direction_disp_tau(&c, u, gs, gt, &act, d_disp, 1.0);
    orc_kink_predict = 1;
    orc_apply_active(&c, &act, d_disp);
  }
}

/* ------------------------------------------------------------------ Part 3: carrot selection (cpp: = src/NeoMpcPlanner.cpp) */
/* createYawFromQuat (cpp:54-62) of a planar quaternion */
static double orc_yaw_planar(double z, double w) { return atan2(2.0 * w * z, 1.0 - 2.0 * z * z); }

/* transformGlobalPlan's pruning (cpp:83-104), getLookAheadDistance (cpp:157-171),
 * getLookAheadPoint (cpp:173-189) and the slow_down_ update (cpp:221-232) for one robot.
 * The tf2 transform plan frame -> base frame (cpp:107-115) is restated as the planar rigid
 * transform given by the robot pose (tf2 itself is not part of the reference: unpinned). */
static void orc_select_carrot(const neo_mpc_lookahead_params* lp, const double* poses, uint32_t np,
                              const double* robot, double footprint_cost, int32_t* slow_down,
                              neo_mpc_carrot* out) {
  memset(out, 0, sizeof(*out));
  out->q[3] = 1.0;
  out->slow_down = *slow_down;
  if (np == 0) { out->status = 1; return; }                             /* cpp:69-71 */
  const double rx = robot[0], ry = robot[1], rth = robot[2];
  uint32_t begin = 0;
  double best = INFINITY;
  for (uint32_t k = 0; k < np; ++k) {                                   /* min_by, cpp:83-88 */
    double d = hypot(poses[3 * k] - rx, poses[3 * k + 1] - ry);
    if (d < best) { best = d; begin = k; }
  }
  out->closer_to_goal = hypot(poses[3 * (np - 1)] - rx, poses[3 * (np - 1) + 1] - ry) <=
                        lp->lookahead_dist_close_to_goal;               /* cpp:95-100 */
  uint32_t end = np;
  for (uint32_t k = begin; k < np; ++k)                                 /* find_if, cpp:102-106 */
    if (hypot(poses[3 * k] - rx, poses[3 * k + 1] - ry) > lp->max_transform_dist) { end = k; break; }
  out->begin = begin; out->end = end;
  if (end == begin) { out->status = 2; return; }                        /* cpp:130-132 */
  double la = lp->lookahead_dist_min;                                   /* cpp:161-169 */
  if (!*slow_down || out->closer_to_goal) {
    la = lp->lookahead_dist_max;
    if (out->closer_to_goal) la = lp->lookahead_dist_close_to_goal;
  }
  out->lookahead_dist = la;
  const double c = cos(rth), s = sin(rth);
  uint32_t pick = end - 1;                                              /* cpp:183-186 */
  double lx = 0.0, ly = 0.0;
  for (uint32_t k = begin; k < end; ++k) {                              /* cpp:177-181 */
    double dx = poses[3 * k] - rx, dy = poses[3 * k + 1] - ry;
    lx = c * dx + s * dy; ly = -s * dx + c * dy;
    if (hypot(lx, ly) >= la) { pick = k; break; }
  }
  {
    double dx = poses[3 * pick] - rx, dy = poses[3 * pick + 1] - ry;
    lx = c * dx + s * dy; ly = -s * dx + c * dy;
  }
  const double yaw_local = poses[3 * pick + 2] - rth;
  out->xy[0] = lx; out->xy[1] = ly;
  out->q[0] = 0.0; out->q[1] = 0.0; out->q[2] = sin(0.5 * yaw_local); out->q[3] = cos(0.5 * yaw_local);
  const double cy = fabs(orc_yaw_planar(out->q[2], out->q[3]));
  int sd;                                                               /* cpp:221-232 */
  if (cy < 1.0) sd = 0;   /* the re-check at cpp:224-227 sees the same pose: never true */
  else if (cy >= 1.0 && footprint_cost > 200) sd = 1;
  else sd = 0;
  *slow_down = sd;
  out->slow_down = sd;
  if (footprint_cost == 255) out->status = 3;                           /* cpp:234-236: the plugin throws, no request */
}

void orc_select_carrots(const neo_mpc_lookahead_params* lp, const neo_mpc_plan_batch* b) {
  for (size_t i = 0; i < b->count; ++i) {
    const uint32_t o0 = b->plan_offsets[i], o1 = b->plan_offsets[i + 1];
    orc_select_carrot(lp, b->plan_poses + 3 * (size_t)o0, o1 - o0, b->robot_poses + 3 * i,
                      b->footprint_costs ? b->footprint_costs[i] : 0.0, &b->slow_down[i], &b->carrots[i]);
    if (b->problems && (b->carrots[i].status == 0 || b->carrots[i].status == 3)) {
      b->problems[i].carrot_xy[0] = b->carrots[i].xy[0];
      b->problems[i].carrot_xy[1] = b->carrots[i].xy[1];
      for (int k = 0; k < 4; ++k) b->problems[i].carrot_q[k] = b->carrots[i].q[k];
      b->problems[i].switch_opt = b->carrots[i].closer_to_goal;         /* cpp:245 */
    }
    /* every non-zero status is a throw in front of the service call (cpp:70, 131, 235): no request this tick */
    if (b->problems) b->problems[i].skip = b->carrots[i].status != 0;
  }
}
