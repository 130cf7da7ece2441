// solver_context.h -- per-instance constants (Ctx) and the record indices of the request / state blocks
// Part of libneo_mpc.so's device code (included by neo_mpc_kernels.hip only).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "neo_mpc_device.h"

namespace neo_mpc {
namespace {

// ---------------------------------------------------------------- problem record indices (doubles)
enum : int {
  P_CUR_X = 0, P_CUR_Y = 1, P_CUR_Q = 2, P_CARROT_X = 6, P_CARROT_Y = 7, P_CARROT_Q = 8,
  P_GOAL = 12, P_GOAL_Q = 15, P_VEL = 19, P_INTERVAL = 22, P_DELTA_T = 23, P_FOOTPRINT = 24,
  PI_MAP_INDEX = 50,  // int32 view of the request record: neo_mpc_problem.map_index
  PI_SKIP = 52,       // ... neo_mpc_problem.skip (no request this tick, cpp:234-236)
  kProblemDoubles = 27,  // what of the 32-double request record the device reads (the rest is reserved)
  kStateDoubles = 16,    // ... of the 16-double state record
  S_LAST = 0, S_OLD_GOAL = 3, S_WAIT = 10, SI_HAS_GOAL = 22, SI_COLLISION = 23, SI_COLL_FP = 24,
  // the build's own hint (no counterpart in the node): the previous solve's first control block BEFORE the low-pass
  // (py:366-367 filters x.x[0:3] in place, so the warm start's last block is the filtered one) and whether it is there.
  // Inside a solve the LDS copy of the slot holds the first block of the un-shifted start, projected, and
  // SI_HAS_PREV == kAltArmed says that the un-shifted start is a candidate of the first iteration (k_solve).
  SI_HAS_PREV = 25, S_PREV_U0 = 13, kAltArmed = 2, kAltLane = 5
};

// The tolerance block of LDS (LdsLayout::tol, doubles): stop tolerances and cold per-instance constants, read once per
// iteration through an opaque offset so that they do not sit in scalar registers; from T_HOP_STAGE on: the hop
// candidates of the current iteration (costmap.h edge_hop) -- four stage indices + their count as int32, then four
// (dvx, dvy) pairs as float32.
enum : int { T_XTOL, T_EARLY, T_FINAL, T_FTOL, T_STALL, T_WTOL, T_KINK, T_KONST, T_TRUE_YAW, T_MU, T_WTOL_LATE,
             T_HOP_DROP, T_HOP_RANGE, T_HOP_STAGE = 13 /* int32[6]: stage[4], count, - */, T_HOP_VEC = 16 /* float[8] */,
             T_BTOL_MAP = 20, T_BTOL_FREE = 21,
             // what the out-of-line cell scan (cell_scan.h) reads instead of taking arguments: the rest of Ctx ...
             T_C0 = 22, T_S0 = 23, T_TYAW = 24, T_FYAW = 25, T_TILE = 26 /* int32[3]: tile_x0, tile_y0, tile_geom */,
             // the routed kernel's stage-wise branch parks what its search carries from one iteration to the next here (gains of
             // the previous two iterations, proximal step length): as registers they were live across the 64-candidate pass,
             // where the candidates themselves sit in registers, and came out spilled to scratch
             T_GAIN1 = 28, T_GAIN2 = 29, T_ALPHA = 30,
             T_FSCAN = 31 /* the objective in front of a round of cell scans (the round decides whether the search is taken up again) */,
             kTolDoubles = 32, kHopLanes = 4 /* = NEO_RULE_HOP_LANES (solver_rules.h) */ };

constexpr int kTileFree = 0x80;   // Ctx::tile_geom: every cell of the reach tile is free (raw cost 0)
constexpr int kTileWall = 0x40;   // ... a lethal cell (raw 254) -- or the outside of the map -- among them: a wall in reach (solver_rules.h)

struct Ctx {
  double cx, cy, tyaw, fyaw, c0, s0, X0, Y0, v0, v1, v2, konst, true_yaw;
  int tile_x0, tile_y0;
  int tile_geom;  // reach tile in LDS: rows << 8 | kTileFree | kTileWall | log2(row stride in bytes); 0: no tile
};

}  // namespace
}  // namespace neo_mpc
