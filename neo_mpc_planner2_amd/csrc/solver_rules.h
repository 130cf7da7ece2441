/* solver_rules.h -- every constant and derived tolerance of the search, ONCE.
 *
 * Plain C: included by the host side of the library (neo_mpc_capi.cpp: derive()), by the device code
 * (neo_mpc_kernels.hip, costmap.h) and by the CPU mirror of the search that the test infrastructure keeps (mpc_oracle.c, part 2 --
 * the mirror follows the build's algorithm by construction and takes its rule book from here; the restatement of the
 * REFERENCE in the same file shares nothing with the product).  A threshold changed here changes on the GPU and in
 * the mirror at once. */
#ifndef NEO_MPC_SOLVER_RULES_H_
#define NEO_MPC_SOLVER_RULES_H_

#include <math.h>

#include "../../include/neo_mpc.h"

/* ---- stop rules */
#define NEO_RULE_STALL_ITERATIONS 5     /* consecutive iterations gaining < ftol (relative) or moving < stall_step that end the search */
#define NEO_RULE_BLOCKED_RUN 3          /* dense direction: this many iterations in a row not won by a decent Newton step ... */
#define NEO_RULE_BLOCKED_STEP 0.25      /* ... (a proximal lane, or a Newton step cut below this) arm the blocked-run rule */
#define NEO_RULE_BLOCKED_TOL_MAP 0.1    /* ... which fires when they gained less than this x opt_tolerance together (costmap term under the rollout) */
#define NEO_RULE_BLOCKED_TOL_FREE 0.03  /* ... (no costmap term under the rollout) */
#define NEO_RULE_LATE_ITERATION 20      /* from here on the three-iteration window is the control_steps-3 one */
#define NEO_RULE_CORNER_ROOM 1e-9        /* second-order directions: a sliding block this close (m/s) to ANOTHER constraint sits in a corner of the feasible set (repin_corner_blocks) */
#define NEO_RULE_CLOSING_RUN 2           /* dense / L-BFGS: blocked iterations in a row before the closing-in rule may end the search */
#define NEO_RULE_WINDOW_STEP 0.5        /* stage-wise direction: an iteration won by a Newton step of at least this length is not "blocked" */
#define NEO_RULE_FINAL_FRAC_GN 0.3      /* a Gauss-Newton (not exact) full step has to be this much shorter than opt_tolerance to be the last */
#define NEO_RULE_TRIAL_RATIO 0.75       /* stage-wise direction, free space: the full step is taken alone when it achieves this share of the predicted decrease */
#define NEO_RULE_KINK_RADIUS 3e-3       /* blocks closer to the control norm's kink are left to the proximal step (dense / L-BFGS directions) */
#define NEO_RULE_KINK_RADIUS_STAGEWISE 1e-4   /* ... (the stage-wise direction predicts landings on the kink inside its sweep) */
/* ---- the piecewise-constant costmap term: wall model and hop candidates (costmap.h) */
#define NEO_RULE_STICKY 100.0           /* penalty on motion across a rising cost step, in units of the tracking curvature */
#define NEO_RULE_STICKY_DIST 0.02       /* ... for stages closer than this to the cell edge (cells) */
#define NEO_RULE_WALL 1e4               /* ... behind a lethal cell */
#define NEO_RULE_WALL_DIST 0.1          /* wall zone and stand-off (cells) */
#define NEO_RULE_HOP_DIST 0.25          /* a cheaper cell is worth a hop candidate when its edge is closer than this (cells) ... */
#define NEO_RULE_HOP_MAX_DV 0.05        /* ... and the hop changes a velocity by less than this (m/s) */
#define NEO_RULE_HOP_MARGIN 0.01        /* a hop lands this far inside the cheaper cell (cells) */
#define NEO_RULE_HOP_MIN_DROP 0.1       /* ... worth it when the term drops by more than this x opt_tolerance */
#define NEO_RULE_HOP_LANES 4
/* ---- cell scan (cell_scan.h): what a search that has ended does about cheaper cells further away than a hop */
#define NEO_RULE_SCAN_CELLS 3           /* the scan looks at the (2 x this + 1)^2 - 1 cells around every stage ... */
#define NEO_RULE_SCAN_RESUME_GAIN 1.0   /* ... and the search is taken up again behind a scan that gained more than this x opt_tolerance */
#define NEO_RULE_SCAN_REPEATS 1         /* scans in a row behind a search: a scan that found a cheaper cell could be followed by another from the new point -- measured (mirror, config 2): the answers that are fixed points of the whole solve to 1e-3 go 95.8 % (1) -> 97.3 % (2) -> 97.7 % (8), each to a LOWER objective; on the GPU every scan of a launch's last waves lengthens the launch: the dense kernel 0.093 -> 0.109 ms (2) -> 0.139 ms (8).  One it stays */

/* search direction of lanes 32-63 */
#define NEO_DIRECTION_LBFGS 0
#define NEO_DIRECTION_DENSE 1
#define NEO_DIRECTION_STAGEWISE 2

typedef struct neo_rules {
  int direction;        /* NEO_DIRECTION_* */
  int max_iterations, lbfgs_memory;
  double xtol;          /* step tolerance */
  double final_tol;     /* Newton directions: a full step below this is the last one */
  double ftol;          /* an iteration gaining less than this (relative to the u-dependent objective) counts as stalled */
  double stall_step;    /* ... or moving less than this */
  double wtol, wtol_late;   /* three iterations together gaining less than this (relative) end the search; 0: off */
  double btol_map, btol_free;   /* blocked-run rule (absolute); 0: off */
  double kink_radius;
  double hop_min_drop;  /* absolute */
  double scan_resume_gain;   /* absolute */
  double flat;          /* (3 / control_steps)^2 beyond 3 control steps with a Newton direction, else 1 */
} neo_rules;

/* AUTO = dense Newton at control_steps 3 unless the costmap weight is heavy (the wall model lives in the stage-wise
 * direction), stage-wise otherwise; NEWTON = dense (control_steps <= 8).  A pinned LBFGS / NEWTON at a heavy costmap weight
 * (refused by neo_mpc_create; reachable through neo_mpc_set_params on a live handle) runs the stage-wise direction. */
static inline int neo_rules_direction(const neo_mpc_params* p) {
  const int heavy_costmap = p->w_costmap > 0.25 * p->w_trans;
  return heavy_costmap && p->method != NEO_MPC_METHOD_AUTO ? NEO_DIRECTION_STAGEWISE
         : p->method == NEO_MPC_METHOD_LBFGS ? NEO_DIRECTION_LBFGS
         : p->method == NEO_MPC_METHOD_NEWTON ? NEO_DIRECTION_DENSE
         : p->method == NEO_MPC_METHOD_RICCATI ? NEO_DIRECTION_STAGEWISE
         : (p->control_steps == 3 && !heavy_costmap ? NEO_DIRECTION_DENSE : NEO_DIRECTION_STAGEWISE);
}

/* hop range in cells for a map of this resolution (dt = prediction_horizon / control_steps) */
static inline double neo_rules_hop_range(double dt, double resolution) {
  return fmin(NEO_RULE_HOP_DIST, NEO_RULE_HOP_MAX_DV * dt / resolution);
}

/* Cells a feasible rollout can get away from the robot's own cell, + 1: the radius of the LDS reach tile (neo_mpc_capi.cpp) and
 * of the cell scan -- the scan looks at no cell further than this from the robot's cell (Chebyshev), so that every kernel
 * variant, tile or no tile, and the CPU mirror consider the same cells. */
static inline int neo_rules_reach_cells(const neo_mpc_params* p, double resolution) {
  const double bx = fmax(fabs(p->min_vel_x), fabs(p->max_vel_x)), by = fmax(fabs(p->min_vel_y), fabs(p->max_vel_y));
  const double vmax = fmin(p->max_vel_trans, hypot(bx, by));
  const double cells = ceil(vmax * p->prediction_horizon / resolution);
  return cells < 1e6 ? (int)cells + 1 : 1000001;
}

/* Width in bytes of the reach tile's rows for a reach of R cells (a power of two that covers 2 R + 1 cells from a column
 * aligned to 4), 0: no tile (neo_mpc_capi.cpp derive(); the CPU mirror's copy of the neighbourhood test below). */
#define NEO_RULE_MAX_TILE_REACH 60
#define NEO_RULE_MAX_TILE_WIDTH 128
static inline int neo_rules_tile_width(int reach) {
  if (reach > NEO_RULE_MAX_TILE_REACH) return 0;
  int w = 4;
  while (w < 2 * reach + 4) w <<= 1;
  return w <= NEO_RULE_MAX_TILE_WIDTH ? w : 0;
}

/* Direction by neighbourhood (round 6).  AUTO at control_steps 3 below the heavy-costmap threshold used to mean the dense
 * direction for every instance.  It has no wall model, and every objective miss the random-parameter fuzz against the
 * reference found at control_steps 3 since round 4 (five in 7200 costmap cases: the builder's 30027, the round-5 review's
 * 61020, 61027, 62024 x 2) was a dense search hemmed in by LETHAL cells -- a wall no candidate crosses: the objective jumps by
 * 1000 / control_steps behind it, every longer step is rejected, the blocked-run rule ends the search with smooth cost left on
 * the table -- while the stage-wise direction, whose wall model slides along such walls, solved all five.  Each instance now
 * takes the direction its neighbourhood asks for: WALL IN REACH (a lethal cell, raw 254, or the outside of the map, among the
 * cells of the reach tile -- rows [my0 - R, my0 + R], columns from (mx0 - R) & ~3, neo_rules_tile_width(R) of them, (mx0, my0)
 * the robot's cell, R = neo_rules_reach_cells) -> stage-wise; no wall in reach -> dense, with the cell scan behind it for the
 * ordinary cost steps as before.  12 % of the BASELINE config-2 instances have a wall in reach.  One launch, one kernel: the
 * wave branches once, behind the set-up that stages the tile (k_solve_routed).  Measured on the CPU mirror against the
 * reference: 305 random parameter sets, 3660 costmap cases, no objective miss (routing on ANY non-free cell -- half of the
 * config-2 instances -- gives the same count). */
static inline int neo_rules_routes_by_neighbourhood(const neo_mpc_params* p) {
  return p->method == NEO_MPC_METHOD_AUTO && p->control_steps == 3 && !(p->w_costmap > 0.25 * p->w_trans);
}

/* All thresholds derive from opt_tolerance (the reference's SLSQP ftol, py:364) unless the caller set one explicitly.
 * neo_rules_derive_as: for a given direction (an instance routed to the stage-wise direction takes that direction's rules). */
static inline void neo_rules_derive_as(const neo_mpc_params* p, int direction, neo_rules* r) {
  const int n = p->control_steps;
  r->direction = direction;
  const int newton = r->direction != NEO_DIRECTION_LBFGS;
  r->max_iterations = p->max_iterations > 0 ? p->max_iterations : 100;
  r->lbfgs_memory = p->lbfgs_memory > 0 ? p->lbfgs_memory : 4;
  r->xtol = p->step_tolerance > 0.0 ? p->step_tolerance : 1e-3 * p->opt_tolerance;
  r->final_tol = p->step_tolerance > 0.0 ? p->step_tolerance : p->opt_tolerance;
  r->stall_step = p->stall_step > 0.0 ? p->stall_step : 0.3 * p->opt_tolerance;
  r->kink_radius = p->kink_radius > 0.0 ? p->kink_radius
                   : r->direction == NEO_DIRECTION_STAGEWISE ? NEO_RULE_KINK_RADIUS_STAGEWISE : NEO_RULE_KINK_RADIUS;
  r->hop_min_drop = NEO_RULE_HOP_MIN_DROP * p->opt_tolerance;
  r->scan_resume_gain = NEO_RULE_SCAN_RESUME_GAIN * p->opt_tolerance;
  /* Beyond 3 control steps the objective is flatter per block (the weights are divided by N, and two neighbouring blocks
   * of a long horizon can trade displacement at almost no cost): the gain thresholds of the Newton directions shrink with
   * (3/N)^2, the three-iteration window with (3/N)^3 */
  r->flat = (newton && n > 3) ? (3.0 / n) * (3.0 / n) : 1.0;
  r->ftol = p->cost_tolerance > 0.0 ? p->cost_tolerance : (newton ? 3e-4 * r->flat : 3e-6) * p->opt_tolerance;
  r->wtol = p->window_tolerance > 0.0 ? p->window_tolerance
            : (p->window_tolerance == 0.0 && newton) ? 3e-3 * p->opt_tolerance * r->flat * fmin(1.0, 3.0 / n) : 0.0;
  r->wtol_late = p->window_tolerance > 0.0 ? p->window_tolerance
                 : (p->window_tolerance == 0.0 && newton) ? 3e-3 * p->opt_tolerance : 0.0;
  /* blocked-run rule: dense direction only, part of the window rule (off with it); absolute (a search on its way out of a
   * lethal cell has a huge f), scaled with the horizon like the stall threshold */
  const int blocked = r->direction == NEO_DIRECTION_DENSE && r->wtol > 0.0;
  r->btol_map = blocked ? NEO_RULE_BLOCKED_TOL_MAP * r->flat * p->opt_tolerance : 0.0;
  r->btol_free = blocked ? NEO_RULE_BLOCKED_TOL_FREE * r->flat * p->opt_tolerance : 0.0;
}
static inline void neo_rules_derive(const neo_mpc_params* p, neo_rules* r) { neo_rules_derive_as(p, neo_rules_direction(p), r); }
/* The rules of an instance the routed control_steps-3 kernel sends to the stage-wise direction: that direction's prox-only zone
 * around the kink (it predicts landings inside its sweep), and the control_steps-3 STOP rules of the dense direction -- the
 * window rule on every run of three iterations, the blocked-run rule with its thresholds, closing-in behind two blocked
 * iterations.  What ends a search depends on the horizon, not on how the direction was computed; the stage-wise exceptions
 * (window and closing-in only behind three blocked iterations, no blocked-run rule) exist for long horizons, whose
 * Gauss-Newton steps converge linearly and whose long shots need blocked iterations to leave lethal cells.  Measured on the
 * mirror: the same objective misses against the reference (none in 3660 costmap cases, every fixture gate unchanged), closed
 * loop of 4096 robots: per-tick maximum 15 -> 13 iterations in the median, 21 -> 15 at worst. */
static inline void neo_rules_derive_routed(const neo_mpc_params* p, neo_rules* r) {
  neo_rules dense;
  neo_rules_derive_as(p, NEO_DIRECTION_DENSE, &dense);
  neo_rules_derive_as(p, NEO_DIRECTION_STAGEWISE, r);
  r->btol_map = dense.btol_map;
  r->btol_free = dense.btol_free;
}

#endif /* NEO_MPC_SOLVER_RULES_H_ */
