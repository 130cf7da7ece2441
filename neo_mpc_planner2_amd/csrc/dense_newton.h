// dense_newton.h -- the dense projected-Newton direction of K1 (control_steps <= 8: a system of at most 24 x 24).
//
// In: hc = column `lane` of the smooth part's Hessian (finite differences of the adjoint gradient, k_solve), the reduced
// gradient gr and the face records the tangent-cone pass wrote (tangent_cone.h: projector P and block curvature C per
// control block).  The system H_r = P (H + C) P + (I - P) is built column-wise then row-wise through LDS, eliminated with
// the rows in registers (pivot row by v_readlane, float32: it only yields a search direction -- the arc search and the
// float64 objective decide) and solved by back substitution.  Out: d (LDS) and this lane's entry of it (return value).
// Blocks sliding in a corner of the feasible set that the step sends outward are pinned and the system solved once more
// (tangent_cone.h: repin_corner_blocks).
#pragma once
#include "neo_mpc_device.h"
#include "wave_ops.h"
#include "tangent_cone.h"

namespace neo_mpc {
namespace {

template <int kSteps, int kNwSteps, bool kTame>
__device__ __forceinline__ float dense_newton_direction(const SolveArgs& a, double* L, float (&hc)[3 * kNwSteps], bool corner_any,
                                                        int lane, int n, int nvr) {
  constexpr int kVars = 3 * kNwSteps;
  const float* NB = reinterpret_cast<const float*>(L + a.lds.cs);
  const double* gr = L + a.lds.gr;
  double* d = L + a.lds.d;
  float newton_sol = 0.0f;
  // From here on the Newton system lives in float32: it only yields a search direction (the arc
  // search and the float64 objective decide), and single precision halves registers, readlanes
  // and VALU time of this section.
  float* Hm = reinterpret_cast<float*>(L + a.lds.hess);
  // this lane's own block (lane = variable index 3 * kb + kq) and that block's record
  const int kv = lane < nvr ? lane : 0, kb = (kv * 11) >> 5, kq = kv - 3 * kb;   // kv / 3 for kv < 32
  const int hs = nvr;  // row stride of the system in LDS
  const float* own = NB + kNewtonRecord * kb;
  // (a block sliding in a corner of the feasible set: the finite-difference columns are kept -- behind the system,
  // in the float64-sized half of its LDS slot -- in case the system has to be solved once more with that block pinned)
  float* Hraw = Hm + hs * hs;
  if (!kTame && corner_any && lane < nvr) {
#pragma unroll
    for (int j = 0; j < kVars; ++j)
      if (kSteps || j < nvr) Hraw[j * hs + lane] = hc[j];
  }
  auto solve_on_the_face = [&]() {
  // ---- lane k < 3N holds Hessian column k: add column kq of its block's curvature, apply P on
  //      the row index, store the column
  {
    const float cn0 = own[4 + kq], cn1 = own[7 + kq], cn2 = own[10 + kq];  // (C is symmetric)
#pragma unroll
    for (int bk = 0; bk < kNwSteps; ++bk) {
      if (kSteps || bk < n) {
        const float* nb = NB + kNewtonRecord * bk;
        const bool mine = lane < nvr && kb == bk;
        const float hx = hc[3 * bk] + (mine ? cn0 : 0.0f), hy = hc[3 * bk + 1] + (mine ? cn1 : 0.0f);
        hc[3 * bk] = nb[0] * hx + nb[1] * hy;
        hc[3 * bk + 1] = nb[1] * hx + nb[2] * hy;
        hc[3 * bk + 2] = (hc[3 * bk + 2] + (mine ? cn2 : 0.0f)) * nb[3];
      }
    }
  }
  if (lane < nvr) {
#pragma unroll
    for (int j = 0; j < kVars; ++j)
      if (kSteps || j < nvr) Hm[j * hs + lane] = hc[j];
  }
  WAVE_SYNC();
  // ---- lane j < 3N holds row j: apply P on the column index, add I - P, eliminate
  float rhsf = 0.0f, diag = 0.0f;
#pragma unroll
  for (int q = 0; q < kVars; ++q) hc[q] = (lane < nvr && (kSteps || q < nvr)) ? Hm[lane * hs + q] : 0.0f;
  if (lane < nvr) rhsf = -(float)gr[lane];
  {
    // row kq of I - P of this lane's block
    const float p00 = own[0], p01 = own[1], p11 = own[2], pw = own[3];
    const float a0 = kq == 0 ? 1.0f - p00 : kq == 1 ? -p01 : 0.0f;
    const float a1 = kq == 0 ? -p01 : kq == 1 ? 1.0f - p11 : 0.0f;
    const float a2 = kq == 2 ? 1.0f - pw : 0.0f;
#pragma unroll
    for (int bk = 0; bk < kNwSteps; ++bk) {
      if (kSteps || bk < n) {
        const float* nb = NB + kNewtonRecord * bk;
        const bool mine = lane < nvr && kb == bk;
        const float hx = hc[3 * bk], hy = hc[3 * bk + 1];
        hc[3 * bk] = hx * nb[0] + hy * nb[1] + (mine ? a0 : 0.0f);
        hc[3 * bk + 1] = hx * nb[1] + hy * nb[2] + (mine ? a1 : 0.0f);
        hc[3 * bk + 2] = hc[3 * bk + 2] * nb[3] + (mine ? a2 : 0.0f);
        if (mine) diag = kq == 0 ? hc[3 * bk] : kq == 1 ? hc[3 * bk + 1] : hc[3 * bk + 2];
      }
    }
  }
  const float deltaf = fmaxf(1e-6f * wave_max_f(fabsf(diag)), 1e-30f);
  auto lane_f = [](float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
  };
  float own_pinv = 0.0f;  // lane pv keeps the reciprocal of its own pivot
#pragma unroll
  for (int pv = 0; pv < kVars; ++pv) {  // Gaussian elimination, rows in registers, pivot row by readlane
    if (kSteps || pv < nvr) {
      float piv = lane_f(hc[pv], pv);
      if (!(piv > deltaf)) piv = fmaxf(fabsf(piv), deltaf);
      const float pinv = __builtin_amdgcn_rcpf(piv);
      if (lane == pv) own_pinv = pinv;
      // (control_steps specialisations: Gauss-Jordan -- the rows ABOVE the pivot are reduced too, the same instructions in
      // lockstep, and the back substitution with its 3N dependent lane reads falls away)
      const float fac = ((kSteps ? lane != pv : lane > pv) && lane < nvr) ? hc[pv] * pinv : 0.0f;
      if (kSteps) {
#pragma unroll
        for (int q = pv + 1; q < kVars; ++q) hc[q] -= fac * lane_f(hc[q], pv);
      } else {
#pragma unroll
        for (int qb = pv / 3; qb < kNwSteps; ++qb) {   // (whole blocks of three columns at a time)
          if (qb < n) {
#pragma unroll
            for (int q = 3 * qb; q < 3 * qb + 3; ++q)
              if (q > pv) hc[q] -= fac * lane_f(hc[q], pv);
          }
        }
      }
      rhsf -= fac * lane_f(rhsf, pv);
    }
  }
  // back substitution, column by column: x_pv leaves lane pv and every row above takes its
  // share off its right-hand side (one readlane + one fma per unknown)
  float sol = 0.0f;
  if (kSteps) sol = rhsf * own_pinv;   // (own_pinv is 0 in the lanes that hold no row)
  else {
#pragma unroll
    for (int pv = kVars - 1; pv >= 0; --pv) {
      if (pv < nvr) {
        const float x = lane_f(rhsf * own_pinv, pv);
        if (lane == pv) sol = x;
        rhsf -= hc[pv] * x;
      }
    }
  }
  if (lane < nvr) d[lane] = (double)sol;
  newton_sol = sol;
  WAVE_SYNC();
  };
  auto columns_again = [&]() {
    if (lane < nvr) {
#pragma unroll
      for (int j = 0; j < kVars; ++j)
        if (kSteps || j < nvr) hc[j] = Hraw[j * hs + lane];
    }
  };
  // one-sided slides (repin_corner_blocks): once more with those blocks pinned.  control_steps specialisations: a second
  // copy of the solve behind a branch, so that the first keeps its straight-line code (as a loop the general
  // control_steps-3 kernel lost a quarter of its rate); the run-time-sized kernel: one copy in a two-trip loop
  if constexpr (kSteps != 0) {
    solve_on_the_face();
    if (!kTame && corner_any && repin_corner_blocks<false>(a, L, n, lane)) { columns_again(); solve_on_the_face(); }
  } else {
    for (int pass = 0;; ++pass) {
      solve_on_the_face();
      if (kTame || pass == 1 || !corner_any || !repin_corner_blocks<false>(a, L, n, lane)) break;
      columns_again();
    }
  }
  return newton_sol;
}

}  // namespace
}  // namespace neo_mpc
