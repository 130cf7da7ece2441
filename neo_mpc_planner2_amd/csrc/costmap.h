// costmap.h -- costmap lookups: world -> cell, bordered device map, LDS reach tile, footprint raster
// Part of libneo_mpc.so's device code (included by neo_mpc_kernels.hip only).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "neo_mpc_device.h"
#include "wave_ops.h"
#include "solver_context.h"
#include "fast_math.h"
#include "solver_rules.h"

namespace neo_mpc {
namespace {

// ---------------------------------------------------------------- costmap
// floor((w - origin) / resolution): multiply by the reciprocal, redo with the exact division
// only when the quotient sits on a cell edge, so the result always equals the division's.
__device__ __forceinline__ int cell_of(double w, double origin, double res, double inv) {
  double t = w - origin;
  double q = t * inv;
  double fl = floor(q);
  if (fabs(q - rint(q)) < 1e-6) fl = floor(t / res);
  // v_cvt_i32_f64 saturates (and maps NaN to 0) in hardware; spelled as an instruction because the
  // C conversion of an out-of-range value is undefined -- saturated indices fall outside every map
  // and read as lethal, like the +-1e9 clamp this replaces (three instructions cheaper per lookup)
  int cell;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(cell) : "v"(fl));
  return cell;
}

__device__ __forceinline__ int map_raw(const DevMap& m, int mx, int my) {
  if (mx < -m.border || my < -m.border || mx >= m.size_x + m.border || my >= m.size_y + m.border)
    return 254;  // contract: out of bounds is lethal
  return m.cells[(long)my * m.pitch + mx];
}

// normalised cost of a raw cell: nav2 occupancy translation / 100 (build's costmap contract)
__device__ __forceinline__ int raw_occupancy(int raw) {
  return raw == 0 ? 0 : raw == 253 ? 99 : raw == 254 ? 100 : raw == 255 ? -1 : 1 + (97 * (raw - 1)) / 251;
}
__device__ __forceinline__ double raw_cost(int raw) { return (double)raw_occupancy(raw) / 100.0; }

// raw cost of cell (mx, my): from the LDS reach tile when it covers the cell, else a bounds-checked global read
// kCovered: the tile is there and covers every cell a feasible rollout can reach, and their neighbours (capi: R =
// ceil(max_vel_trans x horizon / resolution) + 1 cells around the robot's cell; every candidate is projected onto the
// feasible set) -- the static-tile kernels: no bounds test, no global fallback, and the map's geometry words stay out of
// the scalar registers of the solver loop.
template <bool kCovered = false>
__device__ __forceinline__ int cell_raw(const SolveArgs& a, const Ctx& c, const double* L, int mx, int my) {
  const unsigned tx = (unsigned)(mx - c.tile_x0), ty = (unsigned)(my - c.tile_y0);
  const int lg = c.tile_geom & 31;  // (one scalar register for the tile geometry; unpacking is scalar ALU work)
  if (kCovered || ((tx >> lg) == 0u && ty < (unsigned)(c.tile_geom >> 8)))
    return reinterpret_cast<const uint8_t*>(L + a.lds.tile)[(ty << lg) + tx];
  return map_raw(a.map, mx, my);
}

template <bool kCovered = false>
__device__ __forceinline__ double step_term(const SolveArgs& a, const Ctx& c, const double* L, double x, double y) {
  if (c.tile_geom & kTileFree) return L[a.lds.term];   // free neighbourhood (load_tile): the term of a free cell, no lookup
  const double X = c.X0 + (c.c0 * x - c.s0 * y), Y = c.Y0 + (c.s0 * x + c.c0 * y);
  const int mx = cell_of(X, a.map.origin_x, a.map.resolution, a.map.inv_resolution);
  const int my = cell_of(Y, a.map.origin_y, a.map.resolution, a.map.inv_resolution);
  return L[a.lds.term + cell_raw<kCovered>(a, c, L, mx, my)];
}

// Wall model of the stage-wise direction (Riccati kernel).  A stage whose position (x, y: rollout frame) sits within
// kStickyDist cells of a cell edge behind which the costmap term is higher gets motion along that edge's normal
// penalised in the stage model, W += rho n n^T (n = the edge normal, a world axis, in the rollout's frame): the Newton
// direction then slides along the cost step instead of running into it at every step length (searches used to die
// creeping towards such an edge).  rho = kSticky x the tracking curvature for an ordinary cost step.  Behind a LETHAL
// cell (or the map's border) the edge is a wall no candidate will ever cross: within kWallDist cells of it rho = kWall x
// the tracking curvature and the penalty is centred kWallDist cells inside, 1/2 rho (n . dz - pb)^2 -- its linear term
// l = -rho pb n pushes the stage back to that stand-off, so that a finite step along the wall does not end inside it
// (with the soft penalty alone a search blocked by a lethal cell crept up to the wall and ended there with every
// candidate lethal; the stage's path along a straight wall is curved in the controls: a stand-off of 2 % of a cell lets
// a slide advance a centimetre per iteration, 10 % five times that).
// Returns the raw cost of the stage's own cell.
constexpr double kSticky = NEO_RULE_STICKY, kWall = NEO_RULE_WALL, kStickyDist = NEO_RULE_STICKY_DIST, kWallDist = NEO_RULE_WALL_DIST;
// Hop candidates.  The costmap term is piecewise constant: a stage within `hop_range` cells of a cell edge behind which
// the term is LOWER (by more than hop_min_drop) can gain that step for a displacement of millimetres, but no descent
// direction says so -- the term has no gradient -- and at a heavy costmap weight one such step is worth more than the
// whole 1e-3 budget (found with the G9 fixtures: the node's own defaults, every weight 0.5; SLSQP's line search lands
// across such edges by chance).  hop_x / hop_y: the change of THIS stage's block (vx, vy) that puts the stage
// kHopMargin cells inside the cheaper neighbour (every later stage shifts with it); lanes 1-4 of the search try the
// current point with one such block changed (feasible_set.h).
constexpr double kHopMargin = NEO_RULE_HOP_MARGIN;   // (well inside kStickyDist: a stage that has just hopped must not sit ON the edge of the sticky zone; deeper costs objective)
template <bool kCovered = false>
__device__ __forceinline__ int edge_stickiness(const SolveArgs& a, const Ctx& c, const double* L, double x, double y,
                                               double cs, double sn, double& wxx, double& wxy, double& wyy, double& lx,
                                               double& ly, bool& hop, float& hop_x, float& hop_y) {
  wxx = 0.0; wxy = 0.0; wyy = 0.0; lx = 0.0; ly = 0.0;
  hop = false; hop_x = 0.0f; hop_y = 0.0f;
  if (c.tile_geom & kTileFree) return 0;   // free neighbourhood (load_tile): no cost step anywhere near
  const double X = c.X0 + (c.c0 * x - c.s0 * y), Y = c.Y0 + (c.s0 * x + c.c0 * y);
  const int mx = cell_of(X, a.map.origin_x, a.map.resolution, a.map.inv_resolution);
  const int my = cell_of(Y, a.map.origin_y, a.map.resolution, a.map.inv_resolution);
  // (position inside the cell, in cells: by the reciprocal -- distances to an edge are compared with 0.01 ... 0.25)
  const double fx = (X - a.map.origin_x) * a.map.inv_resolution - (double)mx;
  const double fy = (Y - a.map.origin_y) * a.map.inv_resolution - (double)my;
  const int raw_here = cell_raw<kCovered>(a, c, L, mx, my);
  const double here = L[a.lds.term + raw_here];
  // (saturated cell indices -- positions far outside every map -- wrap in mx +- 1; such cells read lethal
  // on both sides, so no edge is sticky there)
  const bool far = mx <= -2147483647 || mx >= 2147483646 || my <= -2147483647 || my >= 2147483646;
  if (far) return raw_here;
  // at most one edge per axis can be within the (wider) wall / hop zone: the low side (push back along +axis) or the high side
  const double hop_range = L[a.lds.tol + T_HOP_RANGE];
  const double zone = fmax(kWallDist, hop_range);
  const bool xlo = fx < zone, xhi = 1.0 - fx < zone, ylo = fy < zone, yhi = 1.0 - fy < zone;
  double rx = 0.0, ry = 0.0, pbx = 0.0, pby = 0.0;
  double drop = L[a.lds.tol + T_HOP_DROP], hwx = 0.0, hwy = 0.0;   // best drop so far, hop in world axes (metres)
  if (xlo || xhi) {
    const int raw_n = cell_raw<kCovered>(a, c, L, xlo ? mx - 1 : mx + 1, my);
    const double dist = xlo ? fx : 1.0 - fx, there = L[a.lds.term + raw_n];
    const bool wall = raw_n == 254 && raw_here != 254;
    if (wall) { if (dist < kWallDist) { rx = kWall * 2.0 * a.p.wt_n; pbx = (xlo ? kWallDist - dist : dist - kWallDist) * a.map.resolution; } }
    else if (dist < kStickyDist && there > here) rx = kSticky * 2.0 * a.p.wt_n;
    if (dist < hop_range && here - there > drop) {
      drop = here - there; hop = true;
      hwx = (xlo ? -1.0 : 1.0) * (dist + kHopMargin) * a.map.resolution; hwy = 0.0;
    }
  }
  if (ylo || yhi) {
    const int raw_n = cell_raw<kCovered>(a, c, L, mx, ylo ? my - 1 : my + 1);
    const double dist = ylo ? fy : 1.0 - fy, there = L[a.lds.term + raw_n];
    const bool wall = raw_n == 254 && raw_here != 254;
    if (wall) { if (dist < kWallDist) { ry = kWall * 2.0 * a.p.wt_n; pby = (ylo ? kWallDist - dist : dist - kWallDist) * a.map.resolution; } }
    else if (dist < kStickyDist && there > here) ry = kSticky * 2.0 * a.p.wt_n;
    if (dist < hop_range && here - there > drop) {
      hop = true;
      hwx = 0.0; hwy = (ylo ? -1.0 : 1.0) * (dist + kHopMargin) * a.map.resolution;
    }
  }
  // world x axis in the rollout frame: (c0, -s0); world y axis: (s0, c0)
  wxx = rx * c.c0 * c.c0 + ry * c.s0 * c.s0;
  wxy = rx * c.c0 * -c.s0 + ry * c.s0 * c.c0;
  wyy = rx * c.s0 * c.s0 + ry * c.c0 * c.c0;
  lx = -(rx * pbx * c.c0 + ry * pby * c.s0);
  ly = -(rx * pbx * -c.s0 + ry * pby * c.c0);
  if (hop) {   // world -> rollout frame -> this block's frame, as a velocity change
    const double hrx = c.c0 * hwx + c.s0 * hwy, hry = -c.s0 * hwx + c.c0 * hwy, idt = rcp_fast(a.p.dt);
    hop_x = (float)((cs * hrx + sn * hry) * idt); hop_y = (float)((-sn * hrx + cs * hry) * idt);
  }
  return raw_here;
}

// Bresenham outline cost of one polygon edge (end points inclusive)
__device__ double edge_cost(const DevMap& m, double ax, double ay, double bx, double by) {
  int x0 = cell_of(ax, m.origin_x, m.resolution, m.inv_resolution);
  int y0 = cell_of(ay, m.origin_y, m.resolution, m.inv_resolution);
  const int x1 = cell_of(bx, m.origin_x, m.resolution, m.inv_resolution);
  const int y1 = cell_of(by, m.origin_y, m.resolution, m.inv_resolution);
  // an end point outside the map reads as lethal (contract), and lethal is the largest cost there is:
  // the outline's maximum is decided -- and a far-away or non-finite vertex (cell_of saturates at
  // +-2^31) cannot turn the walk below into billions of steps.  Inside, it is at most size_x + size_y long.
  if (x0 < 0 || y0 < 0 || x0 >= m.size_x || y0 >= m.size_y || x1 < 0 || y1 < 0 || x1 >= m.size_x || y1 >= m.size_y)
    return raw_cost(254);
  const long dx = labs((long)x1 - x0), dy = labs((long)y1 - y0);
  const int sx = x1 >= x0 ? 1 : -1, sy = y1 >= y0 ? 1 : -1;
  long err = dx - dy;
  double worst = -1.0;
  for (long guard = 0; guard <= dx + dy + 1; ++guard) {
    worst = fmax(worst, raw_cost(map_raw(m, x0, y0)));
    if (x0 == x1 && y0 == y1) break;
    long e2 = 2 * err;
    if (e2 > -dy) { err -= dy; x0 += sx; }
    if (e2 < dx) { err += dx; y0 += sy; }
  }
  return worst;
}

// getFootprintCost of the published footprint (py:343): lanes take edges, wave max
__device__ double footprint_cost(const SolveArgs& a, const double* L, uint32_t b, int lane) {
  if (!a.footprints || a.footprint_points == 0) return L[a.lds.prob + P_FOOTPRINT];
  const int np = (int)a.footprint_points;
  const double* pts = a.footprints + (size_t)b * 2 * np;
  double worst = -1.0;
  for (int e = lane; e < np; e += kLanes) {
    int j = (e + 1) % np;
    worst = fmax(worst, edge_cost(a.map, pts[2 * e], pts[2 * e + 1], pts[2 * j], pts[2 * j + 1]));
  }
  return wave_max(worst);
}

}  // namespace
}  // namespace neo_mpc
