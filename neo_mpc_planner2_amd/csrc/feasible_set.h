// feasible_set.h -- projection onto box x disc and the 64 candidates of one iteration
// Part of libneo_mpc.so's device code (included by neo_mpc_kernels.hip only).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "neo_mpc_device.h"
#include "fast_math.h"
#include "solver_context.h"

namespace neo_mpc {
namespace {

// ---------------------------------------------------------------- feasible set
__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Euclidean projection of (vx, vy) onto box ∩ disc; omega clamped (py:125-134, 157-158)
// kTame: the caller knows at compile time that the disc lies inside the vx/vy box (DevParams.tame)
template <bool kTame = false>
__device__ __forceinline__ void project_block(const DevParams& p, double& b0, double& b1, double& b2) {
  // (v_max / v_min: two instructions where the compare-and-select form of clampd takes eight; a NaN turn rate -- only ever
  // next to NaN velocities, whose candidate is discarded -- comes out as a bound)
  b2 = fmin(fmax(b2, p.lo[2]), p.hi[2]);
  const double zx = b0, zy = b1, r = p.r;
  if (kTame || p.disc_in_box) {  // README parameters: the box never binds, the projection is radial
    const double n2 = zx * zx + zy * zy;
    // (one Newton step on v_rsq_f64: the point lands within 1e-13 of the disc -- and never outside r (1 + 1e-12), which is
    // what the tangent-cone pass calls "on the disc")
    if (n2 > r * r) { const double sc = r * rsq_fast1(n2); b0 = zx * sc; b1 = zy * sc; }
    return;
  }
  const double px = clampd(zx, p.lo[0], p.hi[0]), py = clampd(zy, p.lo[1], p.hi[1]);
  if (px * px + py * py <= r * r) { b0 = px; b1 = py; return; }
  const double nz = sqrt(zx * zx + zy * zy);
  const double qx = zx * (r / nz), qy = zy * (r / nz);
  if (qx >= p.lo[0] && qx <= p.hi[0] && qy >= p.lo[1] && qy <= p.hi[1]) { b0 = qx; b1 = qy; return; }
  double best = INFINITY, bx = px, by = py;  // both bind: closest circle / box-edge intersection
  for (int e = 0; e < 4; ++e) {
    const double fixed = (e == 0) ? p.lo[0] : (e == 1) ? p.hi[0] : (e == 2) ? p.lo[1] : p.hi[1];
    if (fabs(fixed) > r) continue;
    const double o = sqrt(r * r - fixed * fixed);
    for (int s = -1; s <= 1; s += 2) {
      const double ex = (e < 2) ? fixed : s * o, ey = (e < 2) ? s * o : fixed;
      if (ex < p.lo[0] || ex > p.hi[0] || ey < p.lo[1] || ey > p.hi[1]) continue;
      const double dd = (ex - zx) * (ex - zx) + (ey - zy) * (ey - zy);
      if (dd < best) { best = dd; bx = ex; by = ey; }
    }
  }
  b0 = bx; b1 = by;
}

// candidate step multipliers: lanes 0..31 scale the proximal-gradient step by 2^(-12 + l/2),
// lanes 32..63 are step lengths along the L-BFGS direction
#define NEO_QN_STEPS                                                                                  \
  1.0, 0.84, 1.19, 0.71, 1.41, 0.59, 1.68, 0.5, 2.0, 0.42, 2.38, 0.35, 2.83, 0.25, 4.0, 0.177,         \
  0.125, 0.088, 0.0625, 0.044, 0.03125, 0.0156, 0.0078, 0.0039, 0.00195, 0.00098, 4.9e-4, 2.4e-4,    \
  1.2e-4, 6e-5, 3e-5, 1.5e-5
__constant__ double kQnSteps[32] = {NEO_QN_STEPS};
// Which Newton lanes (32-63, long shots included) step at least / less than a given length: bit `lane` of a 64-bit mask -- a
// rule that asks "was the iteration won by a step of at least 0.8" tests one bit with scalar instructions instead of
// comparing a float64 against a constant the compiler keeps in (and, at four waves per SIMD, spills from) a vector register
// pair all through the solver loop.
constexpr unsigned long long newton_lanes_at_least(double t) {
  constexpr double steps[32] = {NEO_QN_STEPS};
  unsigned long long m = 0;
  for (int l = 32; l < 64; ++l) {
    const double s = l >= 61 ? (double)(1 << (l - 58)) : steps[l - 32];
    if (s >= t) m |= 1ull << l;
  }
  return m;
}
constexpr unsigned long long kNewtonLanes = 0xffffffff00000000ull;

// kLongShots (Newton directions): lanes 61-63 trade the three shortest step lengths (L-BFGS's last resort)
// for 8, 16 and 32 -- long shots that find the way out of a lethal cell
template <bool kLongShots>
__device__ __forceinline__ double lane_scale(int lane) {
  if (kLongShots && lane >= 61) return ldexp(1.0, lane - 58);
  if (lane >= 32) return kQnSteps[lane - 32];
  double s = ldexp(1.0, -12 + (lane >> 1));
  return (lane & 1) ? s * 1.4142135623730951 : s;
}

// control block i of this lane's candidate
// (`step`: this lane's step along its own family; `pstep`: its proximal-gradient step length, used
// by the L-BFGS lanes for blocks sitting next to the control-norm kink)
// kRiccati (run-time-sized kernel with the stage-wise Newton direction):
//  * the proximal-gradient step of block i is scaled by N / (N - i): the curvature of a block's own
//    tracking terms is proportional to the number of stages it still moves (diagonal of the
//    Gauss-Newton Hessian, 2 w_trans/N dt^2 (N - i)) -- one step length for all blocks leaves the late
//    blocks crawling at long horizons;
//  * blocks the Riccati sweep sends onto the kink (AMODE slot 3) stop there: step min(t, 1), and
//    exactly v_cur at t >= 1.
//  * hop candidates (lanes 1..count of the hop table, costmap.h): the current point with block `hop_stage` changed by
//    (hop_x, hop_y); hop_stage < 0: an ordinary candidate.
template <bool kTame = false, bool kRiccati = false>
__device__ __forceinline__ void candidate_block(const SolveArgs& a, const Ctx& c, const double* L, int lane,
                                                double step, double pstep, int i, double& b0, double& b1,
                                                double& b2, int hop_stage = -1, float hop_x = 0.0f, float hop_y = 0.0f) {
  const double* u = L + a.lds.u + 3 * i;
  if (kRiccati && hop_stage >= 0) {
    b0 = u[0]; b1 = u[1]; b2 = u[2];
    if (i == hop_stage) { b0 += (double)hop_x; b1 += (double)hop_y; project_block<kTame>(a.p, b0, b1, b2); }
    return;
  }
  const int* am = reinterpret_cast<const int*>(L + a.lds.mode) + 4 * i;
  const bool near = am[2] != 0;
  // (round 4) Newton lanes: every other step length leaves the blocks next to the kink where they are -- their proximal
  // step is made with the gradient at u, the other blocks' Newton step was computed with them held: when both correct
  // the same residual the candidate overshoots, the search cuts the step length for everybody (the "hovering" searches
  // of warm-started ticks: 2.2 % of the converged reference's commands missed by more than 1e-3, none with this)
  if (near && lane >= 32 && (lane & 1)) { b0 = u[0]; b1 = u[1]; b2 = u[2]; return; }
  if (lane < 32 || near) {  // proximal gradient: forward step on the smooth part (for a block next to the kink: reduced
                            // on its face by the tangent-cone pass), prox of the control norm
    if (lane >= 32) step = pstep;
    if (kRiccati) step *= (double)a.p.n * rcp_fast((double)(a.p.n - i));
    const double* gs = L + a.lds.gs + 3 * i;
    const double e0 = (u[0] - step * gs[0]) - c.v0, e1 = (u[1] - step * gs[1]) - c.v1,
                 e2 = (u[2] - step * gs[2]) - c.v2;
    const double ne2 = e0 * e0 + e1 * e1 + e2 * e2;
    const double sh = (ne2 > 0.0) ? fmax(0.0, 1.0 - step * a.p.wc_n * rsq_fast1(ne2)) : 0.0;
    b0 = c.v0 + sh * e0; b1 = c.v1 + sh * e1; b2 = c.v2 + sh * e2;
  } else {          // quasi-Newton / Newton direction
    const double* d = L + a.lds.d + 3 * i;
    if (kRiccati && am[3]) {
      if (step >= 1.0) { b0 = c.v0; b1 = c.v1; b2 = c.v2; }
      else { b0 = u[0] + step * d[0]; b1 = u[1] + step * d[1]; b2 = u[2] + step * d[2]; }
    } else {
      b0 = u[0] + step * d[0]; b1 = u[1] + step * d[1]; b2 = u[2] + step * d[2];
      const double rl = a.p.r * (1.0 - 1e-12);
      if (!kTame && !a.p.disc_in_box && am[0] == 1 && !(am[1] & 2) && u[0] * u[0] + u[1] * u[1] < rl * rl) {
        // (round 4) a block sliding along a box bound stops where the bound meets the speed disc: the Euclidean projection
        // of a point beyond that corner slides DOWN the disc, away from the bound, so the step along the bound used to be
        // cut to the fraction that reaches the corner -- and every other block's step with it.  (A block AT the corner
        // already is left to the projection: from there it slides along the disc.)
        const double nxb = L[a.lds.nx + i];
        const double fixed = nxb != 0.0 ? u[0] : u[1];
        const double lim2 = a.p.r * a.p.r - fixed * fixed, lim = lim2 > 0.0 ? sqrt(lim2) * (1.0 - 1e-15) : 0.0;
        if (nxb != 0.0) { b0 = fixed; b1 = clampd(b1, -lim, lim); }
        else { b1 = fixed; b0 = clampd(b0, -lim, lim); }
      }
    }
  }
  project_block<kTame>(a.p, b0, b1, b2);
}

}  // namespace
}  // namespace neo_mpc
