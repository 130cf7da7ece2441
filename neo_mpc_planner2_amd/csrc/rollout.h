// rollout.h -- rollout + objective of one control sequence (py:224-268), set-up shared by the kernels, K2 (py:365-403)
// Part of libneo_mpc.so's device code (included by neo_mpc_kernels.hip only).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "neo_mpc_device.h"
#include "wave_ops.h"
#include "fast_math.h"
#include "solver_context.h"
#include "costmap.h"
#include "feasible_set.h"

namespace neo_mpc {
namespace {

// rollout + cost of one control sequence (py:224-268); Block(i, b0, b1, b2) yields the controls
struct NoRecord {
  __device__ __forceinline__ void operator()(int, double, double) const {}
};
// kSteps > 0: control_steps known at compile time (loops unroll); Record(i, sin, cos) lets the caller
// keep the rollout's trigonometry (the winner's is reused by the next adjoint sweep)
// term_sum (optional): the sum of the costmap terms alone -- 0.0 exactly when every stage sits in a free cell
template <int kSteps = 0, bool kTame = false, bool kCovered = false, class Block, class Record = NoRecord>
__device__ __forceinline__ double rollout_cost(const SolveArgs& a, const Ctx& c, const double* L, Block block,
                                               Record record = Record(), double* term_sum = nullptr) {
  const DevParams& p = a.p;
  const int n = kSteps ? kSteps : p.n;
  double f = 0.0, x = 0.0, y = 0.0, th = 0.0, ts = 0.0;
#pragma unroll
  for (int i = 0; i < n; ++i) {
    double vx, vy, w;
    block(i, vx, vy, w);
    th += w * p.dt;                                     // py:230
    double sn, cs;
    sincos_heading<kTame>(th, &sn, &cs);
    record(i, sn, cs);
    x += (vx * cs - vy * sn) * p.dt;                    // py:231
    y += (vx * sn + vy * cs) * p.dt;                    // py:232
    const double dx = c.cx - x, dy = c.cy - y, et = c.tyaw - th;
    const double e0 = c.v0 - vx, e1 = c.v1 - vy, e2 = c.v2 - w;
    f += p.wt_n * (dx * dx + dy * dy) + p.wo_n * (et * et);   // py:252
    f += p.wc_n * sqrt_fast(e0 * e0 + e1 * e1 + e2 * e2);      // py:253-254
    const double term = step_term<kCovered>(a, c, L, x, y);    // py:246-247, 257-260
    f += term;
    if (term_sum) ts += term;
  }
  if (term_sum) *term_sum = ts;
  const double et = c.fyaw - th;
  return f + p.wterm_o * (et * et) + c.konst;                  // py:266-268
}

// ---------------------------------------------------------------- set-up shared by the kernels
__device__ void load_term_table(const double* table, double* L, int term, int lane) {
  for (int k = lane; k < 256; k += kLanes) L[term + k] = table[k];
}

__device__ void make_ctx(const DevParams& p, const DevMap& m, const double* P, double fcost, Ctx& c) {
  c.cx = P[P_CARROT_X]; c.cy = P[P_CARROT_Y];
  c.tyaw = yaw_of(P + P_CARROT_Q);                       // py:211
  c.fyaw = yaw_of(P + P_GOAL_Q);                         // py:212
  double q[4] = {P[P_CUR_Q], P[P_CUR_Q + 1], P[P_CUR_Q + 2],
                 (p.compat & NEO_MPC_COMPAT_ODOM_YAW_GOAL_W) ? P[P_GOAL_Q + 3] : P[P_CUR_Q + 3]};
  const double psi0 = yaw_of(q);                          // py:213 (goal's w: reference quirk)
  sincos_fast(psi0, &c.s0, &c.c0);
  c.true_yaw = yaw_of(P + P_CUR_Q);                      // py:317
  c.X0 = P[P_CUR_X]; c.Y0 = P[P_CUR_Y];
  c.v0 = P[P_VEL]; c.v1 = P[P_VEL + 1]; c.v2 = P[P_VEL + 2];
  const double gdx = c.cx - P[P_GOAL], gdy = c.cy - P[P_GOAL + 1];
  c.konst = p.wterm_t * (gdx * gdx + gdy * gdy);          // py:266, 268: constant in u
  if (fcost == 1.0) c.konst += p.w_footprint;             // py:262-263: N steps * w_footprint/N
  c.tile_x0 = 0; c.tile_y0 = 0; c.tile_geom = 0;
}

// make_ctx for the wave-per-instance kernels: the four yaw extractions run side by side in lanes
// 0-3 (one atan2 instead of four); every lane leaves with the same wave-uniform values
__device__ void make_ctx_wave(const DevParams& p, const DevMap& m, const double* P, double fcost, Ctx& c, int lane) {
  // lane 0: carrot (py:211), 1: goal (py:212), 2: current pose with the goal's w (py:213), 3: current pose (py:317)
  const int base = lane == 0 ? P_CARROT_Q : lane == 1 ? P_GOAL_Q : P_CUR_Q;
  const bool goal_w = lane == 2 && (p.compat & NEO_MPC_COMPAT_ODOM_YAW_GOAL_W);
  const double q[4] = {P[base], P[base + 1], P[base + 2], goal_w ? P[P_GOAL_Q + 3] : P[base + 3]};
  const double yaw = yaw_of(q);
  c.tyaw = lane_value(yaw, 0);
  c.fyaw = lane_value(yaw, 1);
  const double psi0 = lane_value(yaw, 2);
  c.true_yaw = lane_value(yaw, 3);
  sincos_fast(psi0, &c.s0, &c.c0);
  c.cx = P[P_CARROT_X]; c.cy = P[P_CARROT_Y];
  c.X0 = P[P_CUR_X]; c.Y0 = P[P_CUR_Y];
  c.v0 = P[P_VEL]; c.v1 = P[P_VEL + 1]; c.v2 = P[P_VEL + 2];
  const double gdx = c.cx - P[P_GOAL], gdy = c.cy - P[P_GOAL + 1];
  c.konst = p.wterm_t * (gdx * gdx + gdy * gdy);          // py:266, 268: constant in u
  if (fcost == 1.0) c.konst += p.w_footprint;             // py:262-263: N steps * w_footprint/N
  c.tile_x0 = 0; c.tile_y0 = 0; c.tile_geom = 0;
}

// stage the reach tile: rows [my0-R, my0+R], columns from floor4(mx0-R), dword loads
__device__ void load_tile(const SolveArgs& a, Ctx& c, double* L, int lane) {
  c.tile_geom = kTileWall;   // (no tile: the kernels cannot look -- every instance counts as next to a wall)
  if (a.lds.tile_w == 0) { c.tile_x0 = 0; c.tile_y0 = 0; return; }
  c.tile_geom = (a.lds.tile_h << 8) | (__ffs(a.lds.tile_w) - 1);
  const int mx0 = cell_of(c.X0, a.map.origin_x, a.map.resolution, a.map.inv_resolution);
  const int my0 = cell_of(c.Y0, a.map.origin_y, a.map.resolution, a.map.inv_resolution);
  c.tile_x0 = (mx0 - a.lds.reach) & ~3;
  c.tile_y0 = my0 - a.lds.reach;
  uint32_t* tile = reinterpret_cast<uint32_t*>(L + a.lds.tile);
  const int wq = a.lds.tile_w >> 2;  // dwords per row (power of two)
  const int total = wq * a.lds.tile_h;
  const int shift = __ffs(wq) - 1;
  uint32_t seen = 0u, wall = 0u;
  for (int idx = lane; idx < total; idx += kLanes) {
    const int row = idx >> shift, col = idx & (wq - 1);
    const long gy = (long)c.tile_y0 + row, gx = (long)c.tile_x0 + 4 * col;
    uint32_t v = 0xFEFEFEFEu;  // lethal outside the padded map
    if (gy >= -a.map.border && gy < (long)a.map.size_y + a.map.border && gx >= -a.map.border &&
        gx + 4 <= (long)a.map.pitch - a.map.border)
      v = *reinterpret_cast<const uint32_t*>(a.map.cells + gy * a.map.pitch + gx);
    tile[idx] = v;
    seen |= v;
    const uint32_t z = v ^ 0xFEFEFEFEu;                      // (a zero byte of z: a lethal cell)
    wall |= (z - 0x01010101u) & ~z & 0x80808080u;
  }
  // Free neighbourhood: every cell the rollout of a feasible candidate can reach (the tile covers them all: its radius
  // is ceil(v_max H / resolution) + 1 cells) has raw cost 0 -- the costmap term is then identically term[0], nothing is
  // sticky and nothing is a hop away: the rollouts skip the lookup (about half of the instances of the BASELINE
  // workloads; a fifth of a candidate's per-stage instructions).
  if (__ballot(seen != 0u) == 0ull) c.tile_geom |= kTileFree;
  // Wall in reach: a lethal cell among them (the map's outside reads lethal) -- what the routed control_steps-3 kernel sends
  // to the stage-wise direction (solver_rules.h neo_rules_routes_by_neighbourhood)
  if (__ballot(wall != 0u) != 0ull) c.tile_geom |= kTileWall;
}

// Global-frame rollout of the controls x from the request's pose and TRUE yaw (py:293-306, 320-327):
// lane i < n leaves with pose i.  The three running sums are accumulated in the reference's order
// (sequentially, every lane alike); only the trigonometry and the products run side by side, one
// step per lane -- one sincos per wave instead of n in a row.
__device__ __forceinline__ void rollout_global(const double* x, int n, double dt, const Ctx& c, int lane,
                                               double& px_i, double& py_i, double& yaw_i) {
  double yaw = c.true_yaw;
  yaw_i = yaw;
  for (int i = 0; i < n; ++i) {
    yaw += x[3 * i + 2] * dt;
    if (lane == i) yaw_i = yaw;
  }
  double sn, cs;
  sincos_fast(yaw_i, &sn, &cs);
  const int k = lane < n ? lane : 0;
  const double inc_x = x[3 * k] * cs * dt - x[3 * k + 1] * sn * dt;   // py:326
  const double inc_y = x[3 * k] * sn * dt + x[3 * k + 1] * cs * dt;   // py:327
  double px = c.X0, py = c.Y0;
  px_i = px; py_i = py;
  for (int i = 0; i < n; ++i) {
    px += lane_value(inc_x, i);
    py += lane_value(inc_y, i);
    if (lane == i) { px_i = px; py_i = py; }
  }
}

// ---------------------------------------------------------------- K2: py:365-403
// `x` (LDS, 3N doubles) is the raw solver output; modified in place like `x.x`.
__device__ void postprocess(const SolveArgs& a, const Ctx& c, double* L, uint32_t b, int lane, double* x,
                            bool success, double fcost, int flags, double cost, int status, int nit, int nfev) {
  const DevParams& p = a.p;
  const int n = p.n, nv = 3 * n;
  double* S = L + a.lds.state;
  int* Si = reinterpret_cast<int*>(S);
  const double* P = L + a.lds.prob;
  // the `local_plan` rollout of the UNFILTERED solution (publishLocalPlan, py:293-306, runs
  // before the low-pass at py:366) from the request's current pose
  if (a.path) {
    double px, py, yaw;
    rollout_global(x, n, p.dt, c, lane, px, py, yaw);
    if (lane < n) {
      double* o = a.path + ((size_t)b * n + lane) * 3;
      o[0] = px; o[1] = py; o[2] = yaw;
    }
  }
  // low-pass on the first control, in place (py:366-367)
  const double g = p.low_pass_gain;
  const double raw0 = x[0], raw1 = x[1], raw2 = x[2];   // (the solution's own first block: next tick's un-shifted start)
  double x0 = x[0] * g + S[S_LAST + 0] * (1 - g);
  double x1 = x[1] * g + S[S_LAST + 1] * (1 - g);
  double x2 = x[2] * g + S[S_LAST + 2] * (1 - g);
  WAVE_SYNC();
  if (lane == 0) { x[0] = x0; x[1] = x1; x[2] = x2; }
  WAVE_SYNC();
  // collision_check (py:312-341): global-frame rollout from the TRUE yaw
  int collision = Si[SI_COLLISION];
  {
    double px, py, yaw;
    rollout_global(x, n, p.dt, c, lane, px, py, yaw);
    const int mx = cell_of(px, a.map.origin_x, a.map.resolution, a.map.inv_resolution);
    const int my = cell_of(py, a.map.origin_y, a.map.resolution, a.map.inv_resolution);
    // cost >= 0.99 (py:338-341) <=> occupancy >= 99: 99 / 100.0 is the double the literal 0.99 denotes
    const bool hit = lane < n && raw_occupancy(map_raw(a.map, mx, my)) >= 99;
    if (__ballot(hit) != 0ull) collision = 1;
  }
  const int coll_fp = (fcost == 1.0) ? 1 : 0;                       // py:343-347
  double out0, out1, out2, waiting = S[S_WAIT];
  if (collision || coll_fp) {                                       // py:374-382
    out0 = out1 = out2 = 0.0;
    flags |= NEO_MPC_FLAG_STOPPED;
    waiting += P[P_DELTA_T];
    if (waiting >= 3.0) { collision = 0; waiting = 0.0; }
  } else {                                                          // py:385-391
    const double ci = P[P_INTERVAL];
    out0 = fmax(fmin(x0, S[S_LAST + 0] + p.acc[0] * ci), S[S_LAST + 0] - p.acc[0] * ci);
    out1 = fmax(fmin(x1, S[S_LAST + 1] + p.acc[1] * ci), S[S_LAST + 1] - p.acc[1] * ci);
    out2 = fmax(fmin(x2, S[S_LAST + 2] + p.acc[2] * ci), S[S_LAST + 2] - p.acc[2] * ci);
  }
  // warm start (py:397-400, 198-202)
  double* warm = a.warm_out + (size_t)b * nv;
  for (int k = lane; k < nv; k += kLanes) {
    double v;
    if (success) v = (k < nv - 3) ? x[k + 3] : x[k - (nv - 3)];
    else v = x[k];
    warm[k] = v;
  }
  WAVE_SYNC();
  if (lane == 0) {
    S[S_LAST + 0] = out0; S[S_LAST + 1] = out1; S[S_LAST + 2] = out2;   // py:393-395
    for (int k = 0; k < 3; ++k) S[S_OLD_GOAL + k] = P[P_GOAL + k];       // py:402
    for (int k = 0; k < 4; ++k) S[S_OLD_GOAL + 3 + k] = P[P_GOAL_Q + k];
    S[S_WAIT] = waiting;
    Si[SI_HAS_GOAL] = 1; Si[SI_COLLISION] = collision; Si[SI_COLL_FP] = coll_fp;
    Si[SI_HAS_PREV] = success ? 1 : 0;   // (an un-shifted warm start, py:399-400, has nothing to un-shift)
    S[S_PREV_U0] = raw0; S[S_PREV_U0 + 1] = raw1; S[S_PREV_U0 + 2] = raw2;
    neo_mpc_command cmd;
    cmd.vel[0] = out0; cmd.vel[1] = out1; cmd.vel[2] = out2;
    cmd.cost = cost; cmd.status = status; cmd.iterations = nit; cmd.evaluations = nfev; cmd.flags = flags;
    a.commands[b] = cmd;
    if (a.velocities) { double* v = a.velocities + 3 * (size_t)b; v[0] = out0; v[1] = out1; v[2] = out2; }
  }
  WAVE_SYNC();
  // the state goes back field by field: last_control, waiting_time, the flags and the previous first block every tick
  // (72 bytes), old_goal (56 bytes) only when it changed -- py:402 assigns it every call, but between two resets it is the same goal
  const bool goal_changed = (flags & NEO_MPC_FLAG_RESET) != 0;
  if (lane < 3 || (lane >= S_WAIT && lane < kStateDoubles) || (goal_changed && lane >= S_OLD_GOAL && lane < S_WAIT))
    reinterpret_cast<double*>(a.states_out + b)[lane] = S[lane];
}

// py:358-361; returns true when the reset is taken.  x0 -> L[u]
__device__ bool reset_and_warm(const SolveArgs& a, double* L, uint32_t b, int lane) {
  double* S = L + a.lds.state;
  int* Si = reinterpret_cast<int*>(S);
  const double* P = L + a.lds.prob;
  bool same = Si[SI_HAS_GOAL] != 0;
  for (int k = 0; k < 3; ++k) same = same && (S[S_OLD_GOAL + k] == P[P_GOAL + k]);
  for (int k = 0; k < 4; ++k) same = same && (S[S_OLD_GOAL + 3 + k] == P[P_GOAL_Q + k]);
  same = uniform_int(same ? 1 : 0) != 0;
  const int nv = 3 * a.p.n;
  WAVE_SYNC();
  if (!same) for (int k = lane; k < nv; k += kLanes) L[a.lds.u + k] = 0.0;   // (else: the warm start load_records brought in)
  if (!same && lane == 0) { S[S_LAST] = 0.0; S[S_LAST + 1] = 0.0; S[S_LAST + 2] = 0.0; S[S_WAIT] = 0.0; }
  WAVE_SYNC();
  return !same;
}

// neo_mpc_problem.skip: the reference made no optimizer call for this robot this tick (the plugin threw in front of it,
// cpp:234-236).  The node's state does not advance -- state record and warm start are not touched --, the command is
// zero twist with NEO_MPC_FLAG_SKIPPED and the optional outputs of the instance are zero rows (defined bytes whatever
// path the batch took to the device).
__device__ void skip_instance(const SolveArgs& a, uint32_t b, int lane, bool solution_is_input = false) {
  const int nv = 3 * a.p.n;
  if (lane == 0) {
    neo_mpc_command cmd;
    cmd.vel[0] = 0.0; cmd.vel[1] = 0.0; cmd.vel[2] = 0.0; cmd.cost = 0.0;
    cmd.status = 0; cmd.iterations = 0; cmd.evaluations = 0; cmd.flags = NEO_MPC_FLAG_SKIPPED;
    a.commands[b] = cmd;
  }
  if (a.velocities && lane < 3) a.velocities[3 * (size_t)b + lane] = 0.0;
  for (int k = lane; k < nv; k += kLanes) {
    if (a.solution && !solution_is_input) a.solution[(size_t)b * nv + k] = 0.0;   // (K2 alone: `solution` is an INPUT)
    if (a.path) a.path[(size_t)b * nv + k] = 0.0;
  }
}

// costmap pool: point `m` at the map the request names (wave-uniform), the single map otherwise
template <bool kUniform = true>
__device__ __forceinline__ void select_map(DevMap& m, const double* P) {
  if (m.pool_count <= 0) return;
  int idx = reinterpret_cast<const int*>(P)[PI_MAP_INDEX];
  idx = idx < 0 ? 0 : (idx >= m.pool_count ? m.pool_count - 1 : idx);
  if (kUniform) idx = __builtin_amdgcn_readfirstlane(idx);
  m.cells += (long)idx * m.pool_stride;
  m.origin_x = m.pool_origins[2 * idx];
  m.origin_y = m.pool_origins[2 * idx + 1];
}

// (only the fields that exist are read: 216 of the 256 bytes of a request; the state record in full -- with
// page-locked host batches worked on in place these loads cross PCIe)
__device__ void load_records(const SolveArgs& a, double* L, uint32_t b, int lane) {
  if (lane < kProblemDoubles) L[a.lds.prob + lane] = reinterpret_cast<const double*>(a.problems + b)[lane];
  else if (lane >= 32 && lane < 32 + kStateDoubles) L[a.lds.state + lane - 32] = reinterpret_cast<const double*>(a.states + b)[lane - 32];
  // the warm start travels with the records (lanes 48-63), not behind the goal comparison that decides whether it is used:
  // one dependent round trip fewer in front of the search -- over PCIe, for page-locked host batches, a few microseconds
  const int nv = 3 * a.p.n;
  for (int k = lane - 48; k >= 0 && k < nv; k += 16) L[a.lds.u + k] = a.warm[(size_t)b * nv + k];
  load_term_table(a.term_table, L, a.lds.term, lane);
  WAVE_SYNC();
}

}  // namespace
}  // namespace neo_mpc
