// cell_scan.h -- what a search of K1 does when it has ended: one look at the costmap cells AROUND every stage.
//
// The costmap term (py:246-247, 256-260) is piecewise constant: it has no gradient, and the hop candidates of the
// stage-wise direction look one cell edge and a quarter of a cell ahead.  SLSQP (py:363-364) gets function values cells
// away from its iterate out of its line search and now and then lands in a cheaper cell one to three cells from where a
// descent method ends (random parameter sets against the reference: 12 of the 14 objective misses in 5568 costmap cases).
// The scan: lane L looks at stage L / lps (lps = 64 / control_steps lanes per stage) and at every lps-th of the 48 cells
// within NEO_RULE_SCAN_CELLS cells of that stage's cell -- none further than the reach tile's radius from the robot's own
// cell, so that every kernel variant and the CPU mirror see the same cells --, keeps the cell with the best estimate among
// those whose term is lower by more than hop_min_drop (term drop minus the tracking cost of moving that stage alone), and
// evaluates the current point with the stage displaced to land kHopMargin cells inside that cell, exactly, twice:
//   (A) block i changed alone -- every later stage shifts with it;
//   (B) block i changed and block i + 1 changed back -- only stage i moves.
// Candidates compete by objective value like every other candidate of the search (invariants 1 and
// 2: scored with the reference's objective, projected onto the feasible set, taken only when strictly lower).
// In: the final iterate u (LDS) and its objective f.  Out: true when a candidate lowered f -- u has one or two blocks
// changed, f is the lower value, *term_sum the sum of the winner's costmap terms; nfev counts the two evaluations.
#pragma once
#include "neo_mpc_device.h"
#include "wave_ops.h"
#include "solver_context.h"
#include "solver_rules.h"
#include "costmap.h"
#include "feasible_set.h"
#include "rollout.h"

namespace neo_mpc {
namespace {

template <int kSteps, bool kTame, bool kCovered>
__device__ __forceinline__ bool cell_scan(const SolveArgs& a, const Ctx& c, double* L, double& f, double* term_sum, int& nfev,
                                       int lane, int n) {
  const DevParams& p = a.p;
  double* u = L + a.lds.u;
  constexpr int kR = NEO_RULE_SCAN_CELLS, kW = 2 * kR + 1, kCells = kW * kW - 1;
  const int lps = kSteps ? kLanes / kSteps : (n < kLanes ? kLanes / n : 1);   // lanes per stage
  const int stage = lane / lps, sector = lane - stage * lps;
  const bool on = stage < n;
  // position and heading of this lane's stage at u, and the heading of the stage behind it (type B)
  double x = 0.0, y = 0.0, th = 0.0, cs = 1.0, sn = 0.0, cs1 = 1.0, sn1 = 0.0;
#pragma unroll
  for (int k = 0; k < (kSteps ? kSteps : n); ++k) {
    if (k <= stage + 1) {
      th += u[3 * k + 2] * p.dt;
      double s_, c_;
      sincos_heading<kTame>(th, &s_, &c_);
      if (k <= stage) {
        x += (u[3 * k] * c_ - u[3 * k + 1] * s_) * p.dt;
        y += (u[3 * k] * s_ + u[3 * k + 1] * c_) * p.dt;
        cs = c_; sn = s_;
      } else { cs1 = c_; sn1 = s_; }
    }
  }
  const double X = c.X0 + (c.c0 * x - c.s0 * y), Y = c.Y0 + (c.s0 * x + c.c0 * y);
  const int mx = cell_of(X, a.map.origin_x, a.map.resolution, a.map.inv_resolution);
  const int my = cell_of(Y, a.map.origin_y, a.map.resolution, a.map.inv_resolution);
  const double here = on ? L[a.lds.term + cell_raw<kCovered>(a, c, L, mx, my)] : 0.0;
  // (no stage of the iterate has a costmap term under it: nothing is cheaper anywhere)
  if (__ballot(here != 0.0) == 0ull) return false;
  const double fx = (X - a.map.origin_x) * a.map.inv_resolution - (double)mx;
  const double fy = (Y - a.map.origin_y) * a.map.inv_resolution - (double)my;
  const int mx0 = uniform_int(cell_of(c.X0, a.map.origin_x, a.map.resolution, a.map.inv_resolution));
  const int my0 = uniform_int(cell_of(c.Y0, a.map.origin_y, a.map.resolution, a.map.inv_resolution));
  const double ex = c.cx - x, ey = c.cy - y;
  const double min_drop = L[a.lds.tol + T_HOP_DROP];
  bool have = false;
  double best_score = 0.0, brx = 0.0, bry = 0.0;   // the displacement of the stage, rollout frame (metres)
  if (on) {
    for (int cc = sector; cc < kCells; cc += lps) {
      const int c2 = cc < kCells / 2 ? cc : cc + 1, dy = c2 / kW - kR, dx = c2 - (dy + kR) * kW - kR;
      const int tx = mx + dx, ty = my + dy;
      if (abs(tx - mx0) > p.scan_reach || abs(ty - my0) > p.scan_reach) continue;
      const double there = L[a.lds.term + cell_raw<kCovered>(a, c, L, tx, ty)];
      if (!(here - there > min_drop)) continue;
      // the nearest point of that cell, kHopMargin cells inside it (cells, relative to the stage)
      const double gx = dx < 0 ? (double)(dx + 1) - fx - kHopMargin : dx > 0 ? (double)dx - fx + kHopMargin : 0.0;
      const double gy = dy < 0 ? (double)(dy + 1) - fy - kHopMargin : dy > 0 ? (double)dy - fy + kHopMargin : 0.0;
      const double wx = gx * a.map.resolution, wy = gy * a.map.resolution;
      const double rx = c.c0 * wx + c.s0 * wy, ry = -c.s0 * wx + c.c0 * wy;
      const double score = (here - there) - p.wt_n * (rx * rx + ry * ry - 2.0 * (rx * ex + ry * ey));
      if (!have || score > best_score) { have = true; best_score = score; brx = rx; bry = ry; }
    }
  }
  if (__ballot(have) == 0ull) { nfev += 2; return false; }   // (the mirror counts its two passes either way)
  // the candidate's blocks: block `stage` changed (A, B), block `stage + 1` changed back (B)
  const double idt = 1.0 / p.dt;
  const int i0 = on ? stage : 0, i1 = stage + 1 < n ? stage + 1 : 0;
  double h0 = u[3 * i0] + (cs * brx + sn * bry) * idt, h1 = u[3 * i0 + 1] + (-sn * brx + cs * bry) * idt, hw = u[3 * i0 + 2];
  project_block<kTame>(p, h0, h1, hw);
  double g0 = u[3 * i1] - (cs1 * brx + sn1 * bry) * idt, g1 = u[3 * i1 + 1] - (-sn1 * brx + cs1 * bry) * idt, gw = u[3 * i1 + 2];
  project_block<kTame>(p, g0, g1, gw);
  double fbest = INFINITY, tbest = 0.0;   // this lane's better candidate: value, term sum
  bool best_b = false;
  // (one rollout body run twice, and the iterate re-read through an offset the compiler cannot see through: unrolled, or with
  // the loads of u hoisted, the two passes keep thirty more vector registers alive than there are)
#pragma nounroll
  for (int type = 0; type < 2; ++type) {
    int u_off = a.lds.u;
    asm volatile("" : "+s"(u_off));
    const double* uu = L + u_off;
    double ct = 0.0;
    double fc = rollout_cost<kSteps, kTame, kCovered>(
        a, c, L,
        [&](int i, double& b0, double& b1, double& b2) {
          b0 = uu[3 * i]; b1 = uu[3 * i + 1]; b2 = uu[3 * i + 2];
          if (i == stage) { b0 = h0; b1 = h1; }
          if (type == 1 && i == stage + 1) { b0 = g0; b1 = g1; }
        },
        NoRecord(), &ct);
    if (!(fc == fc) || !have || (type == 1 && stage >= n - 1)) fc = INFINITY;
    if (fc < fbest) { fbest = fc; tbest = ct; best_b = type == 1; }
  }
  nfev += 2;
  int best = lane;
  double fw = fbest;
  wave_argmin(fw, best);
  if (!(fw < f)) return false;
  WAVE_SYNC();
  if (lane == best) {
    u[3 * stage] = h0; u[3 * stage + 1] = h1;
    if (best_b) { u[3 * stage + 3] = g0; u[3 * stage + 4] = g1; }
  }
  f = fw;
  if (term_sum) *term_sum = lane_value(tbest, best);
  WAVE_SYNC();
  return true;
}

}  // namespace
}  // namespace neo_mpc
