// lbfgs.h -- the projected L-BFGS direction of K1 (kDir 0: any control_steps; the fallback of `method = LBFGS`).
//
// Per iteration: the newest curvature pair (s, y) = (u - u_prev, gt - gt_prev) joins the ring of `mem` pairs when it has
// positive curvature; the two-loop recursion runs on the REDUCED gradient (lanes take vector elements, up to 192 = 3
// per lane; pair slots are walked with compile-time indices so the alphas stay in registers); the result is restricted
// to the face of the tangent cone the active-set pass described (tangent_cone.h).
// LDS in: u, u_prev, gt, gt_prev, gr, S, Y, rho, mode / nx / ny.  LDS out: d, S, Y, rho.  head / npairs: the ring's state.
#pragma once
#include "neo_mpc_device.h"
#include "wave_ops.h"
#include "fast_math.h"

namespace neo_mpc {
namespace {

template <int kSteps, int kPairs>
__device__ __forceinline__ void lbfgs_direction(const SolveArgs& a, double* L, int it, int mem, int& head, int& npairs, int lane, int n) {
  const int nv = 3 * n;
  const double* u = L + a.lds.u;
  const double* u_prev = L + a.lds.u_prev;
  const double* gt = L + a.lds.gt;
  const double* gt_prev = L + a.lds.gt_prev;
  const double* gr = L + a.lds.gr;
  double* d = L + a.lds.d;
  double* Sm = L + a.lds.S;
  double* Ym = L + a.lds.Y;
  double* rho = L + a.lds.rho;
  const double* ANX = L + a.lds.nx;
  const double* ANY = L + a.lds.ny;
  const int* AMODE = reinterpret_cast<const int*>(L + a.lds.mode);
  // ---- new curvature pair (blocks next to the kink, now or at the previous iterate, stay out of it)
  if (it > 0) {
    double* s = Sm + head * nv;
    double* yv = Ym + head * nv;
    double sy = 0.0, ss = 0.0, yy = 0.0;
    for (int k = lane; k < nv; k += kLanes) {
      const int blk = k / 3;
      const bool skip = (AMODE[4 * blk + 2] | AMODE[4 * blk + 3]) != 0;
      const double sk = skip ? 0.0 : u[k] - u_prev[k], yk = skip ? 0.0 : gt[k] - gt_prev[k];
      s[k] = sk; yv[k] = yk;
      sy += sk * yk; ss += sk * sk; yy += yk * yk;
    }
    sy = wave_sum(sy); ss = wave_sum(ss); yy = wave_sum(yy);
    const int ok = uniform_int((ss > 0.0 && sy > 1e-10 * sqrt_fast(ss * yy)) ? 1 : 0);
    if (ok) {
      if (lane == 0) rho[head] = rcp_fast(sy);
      head = (head + 1) % mem;
      if (npairs < mem) ++npairs;
    }
    WAVE_SYNC();
  }
  // ---- L-BFGS two-loop recursion on the reduced gradient; lanes take vector elements.
  //      Pair slots are walked with compile-time indices so the alphas stay in registers.
  {
    constexpr bool kWide = (kSteps == 0) || (3 * kSteps > 64);  // more than 64 variables
    double al[kPairs];
    double q0 = lane < nv ? gr[lane] : 0.0;
    double q1 = (kWide && lane + 64 < nv) ? gr[lane + 64] : 0.0;
    double q2 = (kWide && lane + 128 < nv) ? gr[lane + 128] : 0.0;
#pragma unroll
    for (int j = 0; j < kPairs; ++j) {
      if (j < npairs) {
        const int idx = (head - 1 - j + 2 * mem) % mem;
        const double* s = Sm + idx * nv;
        const double* yv = Ym + idx * nv;
        double part = 0.0;
        if (lane < nv) part += s[lane] * q0;
        if (kWide && lane + 64 < nv) part += s[lane + 64] * q1;
        if (kWide && lane + 128 < nv) part += s[lane + 128] * q2;
        al[j] = rho[idx] * wave_sum(part);
        if (lane < nv) q0 -= al[j] * yv[lane];
        if (kWide && lane + 64 < nv) q1 -= al[j] * yv[lane + 64];
        if (kWide && lane + 128 < nv) q2 -= al[j] * yv[lane + 128];
      }
    }
    if (npairs > 0) {
      const int idx = (head - 1 + mem) % mem;
      const double* yv = Ym + idx * nv;
      double part = 0.0;
      if (lane < nv) part += yv[lane] * yv[lane];
      if (kWide && lane + 64 < nv) part += yv[lane + 64] * yv[lane + 64];
      if (kWide && lane + 128 < nv) part += yv[lane + 128] * yv[lane + 128];
      const double gamma = rcp_fast(rho[idx] * wave_sum(part));
      q0 *= gamma; q1 *= gamma; q2 *= gamma;
    }
#pragma unroll
    for (int j = kPairs - 1; j >= 0; --j) {
      if (j < npairs) {
        const int idx = (head - 1 - j + 2 * mem) % mem;
        const double* s = Sm + idx * nv;
        const double* yv = Ym + idx * nv;
        double part = 0.0;
        if (lane < nv) part += yv[lane] * q0;
        if (kWide && lane + 64 < nv) part += yv[lane + 64] * q1;
        if (kWide && lane + 128 < nv) part += yv[lane + 128] * q2;
        const double be = rho[idx] * wave_sum(part);
        if (lane < nv) q0 += s[lane] * (al[j] - be);
        if (kWide && lane + 64 < nv) q1 += s[lane + 64] * (al[j] - be);
        if (kWide && lane + 128 < nv) q2 += s[lane + 128] * (al[j] - be);
      }
    }
    if (lane < nv) d[lane] = -q0;
    if (lane + 64 < nv) d[lane + 64] = -q1;
    if (lane + 128 < nv) d[lane + 128] = -q2;
    WAVE_SYNC();
  }
  {
    // (the Newton system is built on the face: H_r = P H P + (I - P) with a right-hand side inside
    // it, so its solution needs no restriction -- what rounding leaves outside is removed by the
    // projection of every candidate)
    for (int i = lane; i < n; i += kLanes) {  // restrict the direction to the tangent cone's face
      if (AMODE[4 * i + 2]) { d[3 * i] = 0.0; d[3 * i + 1] = 0.0; d[3 * i + 2] = 0.0; continue; }
      if (AMODE[4 * i + 1] & 1) d[3 * i + 2] = 0.0;
      const int mode = AMODE[4 * i];
      if (mode == 1) {
        const double dot = d[3 * i] * ANX[i] + d[3 * i + 1] * ANY[i];
        d[3 * i] -= dot * ANX[i]; d[3 * i + 1] -= dot * ANY[i];
      } else if (mode == 2) {
        d[3 * i] = 0.0; d[3 * i + 1] = 0.0;
      }
    }
    WAVE_SYNC();
  }
}

}  // namespace
}  // namespace neo_mpc
