// riccati.h -- projected Newton direction by a stage-wise (Riccati) recursion over the rollout chain
// Part of libneo_mpc.so's device code (included by neo_mpc_kernels.hip only).
//
// The system is the one the dense Newton kernels solve -- H_r d = -g_r, H the exact Hessian of the
// smooth part of the objective (py:250-252, 266-268) plus the control norm's curvature (py:253-254)
// and a binding disc's (py:157-158), restricted to the tangent cone's face -- but the rollout
// (py:230-232) is a chain z_i = F(z_{i-1}, u_i), z = (x, y, theta), so
//     H = sum_i J_i^T W_i J_i + sum_i lambda_i . d2F_i + blockdiag(R_i)
// (W_i: Hessian of the stage cost, lambda_i = (SX_i, SY_i): position costates of the adjoint sweep,
// J_i: sensitivity of z_i) and H d = -g is a linear-quadratic problem: one backward sweep with 3x3
// value-function Hessians and one forward sweep, O(control_steps), no (3N)^2 matrix and no finite
// differences.
//
// Far from the minimiser the sweep keeps the GAUSS-NEWTON part of H (the lambda_i . d2F_i terms are left out): every
// stage system is then positive definite by construction, the value function stays positive semi-definite, and the
// recursion is safe in float32 -- with the second-order terms the stage systems turn indefinite there, pivots get
// replaced and the forward sweep can blow up (seen at control_steps 64: |d| ~ 1e53; a search blocked by a lethal wall
// crept along it, G8 "turn").  Round 4: behind an iteration won by a Newton step of at least half its length (the model
// held there) the sweep carries the second-order terms (k_solve scales the costates it stores by tau = 0 / 1): the
// Gauss-Newton direction converges LINEARLY wherever the tracking residuals are large -- on the held-out parameter sets
// (G10 "a", "c": heavy control / tracking weights) first controls ended 1.4e-3 ... 3.2e-3 from SLSQP's converged ones
// when a stop rule cut in; with the exact Hessian near the minimiser: <= 3e-4 everywhere, in 15-35 % fewer iterations
// on those sets and as many as before at the README's weights.
//
// Beyond 8 control steps the block curvature carries an adaptive Levenberg-Marquardt term (k_solve keeps mu: two
// neighbouring blocks of a long horizon trade displacement at almost no cost, and the undamped step along such a
// valley leaves the region where the model holds); a stage next to a cost step carries the wall model of costmap.h
// (curvature on the stage position, and behind a lethal cell a linear push-back term that enters the value gradient).
//
// Three passes:
//   riccati_prepare  lane = stage: rotates everything the sweep needs into the stage's DISPLACEMENT
//                    coordinates w = B0 du, B0 = dt diag(Rot(theta_i), 1) -- the linearised step is then
//                    dz_i = A_i (dz_{i-1} + w_i), A_i = [[1 0 -py] [0 1 px] [0 0 1]], and the Gauss-Newton part
//                    of Quu, Quz and Qzz is ONE matrix M = A^T S A (derivation in DESIGN.md).
//   riccati_sweep    wave-uniform: every lane runs the recursion on the same LDS records, in float32 (it
//                    only yields a search direction; the float64 objective decides); each stage is solved
//                    in the coordinates of its face (0-3 free directions) instead of through 3x3 projector
//                    products.
//   riccati_finish   lane = stage: back to control coordinates, du = B0^-1 w.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "neo_mpc_device.h"
#include "fast_math.h"
#include "solver_context.h"

namespace neo_mpc {
namespace {

// float32 record of one stage (LDS, 28 floats = seven 16-byte words): rollout step, face, wall penalty and push-back, flags,
// block curvature, trigonometry; then the stage's linear terms, which the backward sweep replaces IN PLACE by
// the stage's gains once it has consumed them
enum : int {
  RS_PX = 0, RS_PY, RS_TX, RS_TY,
  RS_WXX, RS_WXY, RS_WYY, RS_FLAGS,
  RS_C00, RS_C01, RS_C02, RS_C11,
  RS_C12, RS_C22, RS_CS, RS_SN,
  RS_GT = 16,          // [3] total gradient, displacement coordinates   -> K row 0, K[1][0]
  RS_GS = 19,          // [3] smooth gradient                             -> K[1][1..2], K[2][0]
  RS_WK = 22,          // [3] the step onto the kink                      -> K[2][1..2], k[0]
  RS_WLX = 25, RS_WLY, // wall push-back (linear term on the stage position, costmap.h) -> k[1], k[2]
  RS_SY = 27,          // tau x SY_i: position costate of the stage (second-order terms; tau x SX_i sits next to the disc
                       // curvature in the rt slot of the stage)                   -> -
  RS_GAIN = 16,        // K (row-major 3x3) then k: 12 floats from here
  kRicStage = 28
};
// RS_FLAGS: the stage's case (what is free) + whether it may be sent onto the kink
enum : int { RC_FREE3 = 0, RC_SLIDE_W, RC_XY, RC_SLIDE, RC_W, RC_NONE, RC_KINK, RF_CASE = 7, RF_KINK_OK = 8 };

typedef float ric_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float ric_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double ric_rcp(double x) { return rcp_fast(x); }
template <typename T> __device__ __forceinline__ T ric_max(T a, T b) { return a > b ? a : b; }
template <typename T> __device__ __forceinline__ T ric_abs(T a) { return a < (T)0 ? -a : a; }
__device__ __forceinline__ float ric_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double ric_fma(double a, double b, double c) { return fma(a, b, c); }
// non-positive pivots are replaced: the factorisation is then that of a positive definite matrix and the stage
// step a descent direction (Gauss-Newton stage systems are positive definite; this guards rounding)
template <typename T> __device__ __forceinline__ T ric_pivot(T p, T delta) { return p > delta ? p : ric_max(ric_abs(p), delta); }

// lane = stage.  In: gt / gs / u / tangent-cone description (mode, wfroz, near | nx, ny, k2 = lambda/r of a
// binding disc) and the stage's cs, sn (record).  Out (record): gradients and the step onto the kink in
// displacement coordinates, tangent, flags, block curvature.
// mu: Levenberg-Marquardt damping in units of one stage's tracking weights (k_solve adapts it).
__device__ __forceinline__ void riccati_prepare(const SolveArgs& a, const Ctx& c, double* L, int n, int lane,
                                                bool v_feasible, float mu) {
  if (lane >= n) return;
  const DevParams& p = a.p;
  float* rs = reinterpret_cast<float*>(L + a.lds.ric) + kRicStage * lane;
  const int* am = reinterpret_cast<const int*>(L + a.lds.mode) + 4 * lane;
  const double* gt = L + a.lds.gt + 3 * lane;
  const double* gs = L + a.lds.gs + 3 * lane;
  const double* u = L + a.lds.u + 3 * lane;
  const double cs = rs[RS_CS], sn = rs[RS_SN];   // (float32 copies: a direction's worth of accuracy)
  const double idt = rcp_fast(p.dt);
  // gradients: g~ = B0^-T g = (Rot g_xy, g_w) / dt
  rs[RS_GT] = (float)((cs * gt[0] - sn * gt[1]) * idt); rs[RS_GT + 1] = (float)((sn * gt[0] + cs * gt[1]) * idt);
  rs[RS_GT + 2] = (float)(gt[2] * idt);
  rs[RS_GS] = (float)((cs * gs[0] - sn * gs[1]) * idt); rs[RS_GS + 1] = (float)((sn * gs[0] + cs * gs[1]) * idt);
  rs[RS_GS + 2] = (float)(gs[2] * idt);
  // the step onto the kink u_i = v_cur: w = B0 (v - u_i)
  const double e0 = u[0] - c.v0, e1 = u[1] - c.v1, e2 = u[2] - c.v2;
  rs[RS_WK] = (float)(-(cs * e0 - sn * e1) * p.dt); rs[RS_WK + 1] = (float)(-(sn * e0 + cs * e1) * p.dt);
  rs[RS_WK + 2] = (float)(-e2 * p.dt);
  // face: tangent of a sliding block, rotated; flags
  const double nx = L[a.lds.nx + lane], ny = L[a.lds.ny + lane];
  const double tx = -ny, ty = nx;
  const float rtx = (float)(cs * tx - sn * ty), rty = (float)(sn * tx + cs * ty);
  rs[RS_TX] = rtx; rs[RS_TY] = rty;
  const int near = am[2];
  const int xy = near ? 2 : am[0];          // 0: both velocity directions free, 1: sliding along the tangent, 2: pinned
  const bool wfree = !(near || (am[1] & 1));
  const int kase = xy == 0 ? (wfree ? RC_FREE3 : RC_XY) : xy == 1 ? (wfree ? RC_SLIDE_W : RC_SLIDE) : (wfree ? RC_W : RC_NONE);
  const int flags = kase | ((!near && v_feasible) ? RF_KINK_OK : 0);
  rs[RS_FLAGS] = (float)flags;   // (a small integer as a float VALUE: its bit pattern would be a denormal)
  // block curvature R~ = B0^-T R B0^-1: the control norm's Hessian (w/|e|)(I - h h^T), h = e/|e| rotated, plus
  // lambda/r t t^T of a binding disc -- both divided by dt^2
  float c00 = 0.0f, c01 = 0.0f, c02 = 0.0f, c11 = 0.0f, c12 = 0.0f, c22 = 0.0f;
  if (!near) {
    const float f0 = (float)(cs * e0 - sn * e1), f1 = (float)(sn * e0 + cs * e1), f2 = (float)e2;
    const float fn2 = f0 * f0 + f1 * f1 + f2 * f2;
    const float ine = fn2 > 0.0f ? __builtin_amdgcn_rsqf(fn2) : 0.0f;
    const float i2 = (float)(idt * idt);
    const float sN = (float)p.wc_n * ine * i2, h0 = f0 * ine, h1 = f1 * ine, h2 = f2 * ine;
    const float k2 = reinterpret_cast<const float*>(L + a.lds.rt)[2 * lane] * i2;
    c00 = sN * (1.0f - h0 * h0) + k2 * rtx * rtx; c01 = -sN * h0 * h1 + k2 * rtx * rty; c02 = -sN * h0 * h2;
    c11 = sN * (1.0f - h1 * h1) + k2 * rty * rty; c12 = -sN * h1 * h2; c22 = sN * (1.0f - h2 * h2);
    const float mt = mu * (float)(2.0 * p.wt_n);
    c00 += mt; c11 += mt; c22 += mu * (float)(2.0 * p.wo_n);
  }
  rs[RS_C00] = c00; rs[RS_C01] = c01; rs[RS_C02] = c02; rs[RS_C11] = c11; rs[RS_C12] = c12; rs[RS_C22] = c22;
}

// The backward sweep writes each stage's gains over the linear terms of its record (RS_GAIN .. the end).  A sweep that may
// have to be repeated with another active set (k_solve: repin_corner_blocks) keeps those twelve floats per stage aside
// (save) and puts them back (!save); lane = stage.
__device__ __forceinline__ void riccati_keep_linear_terms(const SolveArgs& a, double* L, int n, int lane, bool save) {
  ric_f4* rec = reinterpret_cast<ric_f4*>(reinterpret_cast<float*>(L + a.lds.ric) + kRicStage * lane + RS_GAIN);
  ric_f4* keep = reinterpret_cast<ric_f4*>(L + a.lds.keep) + 3 * lane;
  for (int i = lane; i < n; i += kLanes, rec += 7 * kLanes, keep += 3 * kLanes) {
    if (save) { keep[0] = rec[0]; keep[1] = rec[1]; keep[2] = rec[2]; }
    else { rec[0] = keep[0]; rec[1] = keep[1]; rec[2] = keep[2]; }
  }
  WAVE_SYNC();
}

// Backward + forward sweep in displacement coordinates, wave-uniform.  The record of stage i - 1 is fetched
// (seven 16-byte LDS reads) while stage i is worked on: nothing in it depends on the recursion.  Output: w_i
// in d (float64 slots), and per block tokink (AMODE slot 3): the stage model's minimiser is the kink itself.
// kPrefetch: fetch the next stage's record one stage ahead (28 more live registers: the 4-waves/SIMD build
// reads each record when it needs it instead)
// kFew > 0: control_steps == kFew at compile time (the loops unroll and the first stage takes its shortcut)
template <typename T, bool kPrefetch = true, int kFew = 0>
__device__ __forceinline__ void riccati_sweep(const SolveArgs& a, double* L, int n, int lane) {
  const DevParams& p = a.p;
  float* RS = static_cast<float*>(__builtin_assume_aligned(reinterpret_cast<float*>(L + a.lds.ric), 16));
  int* AMODE = reinterpret_cast<int*>(L + a.lds.mode);
  double* d = L + a.lds.d;
  const float* ARTF = reinterpret_cast<const float*>(L + a.lds.rt);
  const T w2 = (T)(2.0 * p.wt_n), wo2 = (T)(2.0 * p.wo_n), wterm2 = (T)(2.0 * p.wterm_o);
  // kink test |B0^T r| <= w_control/N  <=>  |r| <= w_control / (N dt)
  const T wc2 = (T)((p.wc_n * p.wc_n) / (p.dt * p.dt));

  T V00 = (T)0.0, V01 = (T)0.0, V02 = (T)0.0, V11 = (T)0.0, V12 = (T)0.0, V22 = (T)0.0, v0 = (T)0.0, v1 = (T)0.0, v2 = (T)0.0;
  ric_f4 r0, r1, r2, r3, r4, r5, r6;
  if (kPrefetch) {
    const ric_f4* R4 = reinterpret_cast<const ric_f4*>(RS + kRicStage * (n - 1));
    r0 = R4[0]; r1 = R4[1]; r2 = R4[2]; r3 = R4[3]; r4 = R4[4]; r5 = R4[5]; r6 = R4[6];
  }
  // (the first stage, i == 0, has nothing behind it: its feedback gains K would multiply dz_{-1} = 0 and the value function
  // V, v it would hand on is never read -- in the unrolled sweeps every case below stops after the stage's own step k; K
  // stays zero.  The run-time-sized sweep does the work: one stage of eight or more, against a test in every stage.)
  for (int i = (kFew ? kFew : n) - 1; i >= 0; --i) {
    const bool first = kFew > 0 && i == 0;
    if (!kPrefetch) {
      const ric_f4* R4 = reinterpret_cast<const ric_f4*>(RS + kRicStage * i);
      r0 = R4[0]; r1 = R4[1]; r2 = R4[2]; r3 = R4[3]; r4 = R4[4]; r5 = R4[5]; r6 = R4[6];
    }
    // this stage's record (registers), the next one on its way
    const T px = r0.x, py = r0.y, rtx = r0.z, rty = r0.w;
    const T wxx = r1.x, wxy = r1.y, wyy = r1.z;
    const int flags = __builtin_amdgcn_readfirstlane((int)r1.w);   // (wave-uniform: scalar branches below)
    const T c00 = r2.x, c01 = r2.y, c02 = r2.z, c11 = r2.w, c12 = r3.x, c22 = r3.y;
    const T gt0 = r4.x, gt1 = r4.y, gt2 = r4.z, gs0 = r4.w, gs1 = r5.x, gs2 = r5.y, e0 = r5.z, e1 = r5.w, e2 = r6.x;
    v0 += (T)r6.y; v1 += (T)r6.z;   // wall push-back: linear term of this stage's own cost in its position
    // position costates of the stage, already scaled by tau (0: Gauss-Newton sweep)
    const T sy = (T)r6.w, sx = (T)ARTF[2 * i + 1];
    if (kPrefetch && i > 0) {
      const ric_f4* R4 = reinterpret_cast<const ric_f4*>(RS + kRicStage * (i - 1));
      r0 = R4[0]; r1 = R4[1]; r2 = R4[2]; r3 = R4[3]; r4 = R4[4]; r5 = R4[5]; r6 = R4[6];
    }
    // S = W_i + wall_i + V;  M = A^T S A  (Gauss-Newton: Quu_s = Quz = Qzz = M)
    const T S00 = V00 + w2 + wxx, S01 = V01 + wxy, S11 = V11 + w2 + wyy;
    const T S22 = V22 + wo2 + (i == n - 1 ? wterm2 : (T)0.0);
    const T M02 = ric_fma(-py, S00, ric_fma(px, S01, V02));
    const T M12 = ric_fma(-py, S01, ric_fma(px, S11, V12));
    const T M22 = ric_fma(-py, M02, ric_fma(px, M12, ric_fma(-py, V02, ric_fma(px, V12, S22))));
    // second-order terms of the rollout step, weighted by the costate of its result (displacement coordinates):
    // T = [[0 0 SY] [0 0 -SX] [SY -SX kappa]], kappa = -(SX px + SY py).  They enter Quu whole, Quz in its theta
    // COLUMN (rows of Quz: (S00 S01 Z02), (S01 S11 Z12), (M02 M12 Z22)) and Qzz in its corner only.
    const T Z02 = M02 + sy, Z12 = M12 - sx, Z22 = M22 - ric_fma(sx, px, sy * py);
    // linear terms: Qz = A^T v, Qu = g~ + Qz
    const T z0 = v0, z1 = v1, z2 = ric_fma(-py, v0, ric_fma(px, v1, v2));
    // ---- the stage system in the coordinates of its face.  Six straight-line cases (wave-uniform switch):
    //      what is free -- both velocity directions or one (the tangent a of a sliding block) or none, and omega.
    //      Rows of Quz (= M): (S00 S01 M02), (S01 S11 M12), (M02 M12 M22).  v = Qz + Quz^T k, V = Qzz + Quz^T K,
    //      upper triangle (symmetric in exact arithmetic), written as rank-one updates per free direction.
    T k0 = (T)0.0, k1 = (T)0.0, k2 = (T)0.0, K00 = (T)0.0, K01 = (T)0.0, K02 = (T)0.0, K10 = (T)0.0, K11 = (T)0.0, K12 = (T)0.0,
      K20 = (T)0.0, K21 = (T)0.0, K22 = (T)0.0;
    bool tokink = false;
    int kase = flags & RF_CASE;
    if (flags & RF_KINK_OK) {
      // does the stage model put this block ON the kink?  0 in Qu_s + M k + w d|.| at k = the step onto the kink
      // <=> |Qu_s + M k| <= w  (smooth parts only)
      const T q0 = gs0 + z0 + S00 * e0 + S01 * e1 + Z02 * e2;
      const T q1 = gs1 + z1 + S01 * e0 + S11 * e1 + Z12 * e2;
      const T q2 = gs2 + z2 + Z02 * e0 + Z12 * e1 + Z22 * e2;
      if (__builtin_amdgcn_readfirstlane((int)(q0 * q0 + q1 * q1 + q2 * q2 <= wc2))) { tokink = true; kase = RC_KINK; }
    }
    const T q0 = gt0 + z0, q1 = gt1 + z1, q2 = gt2 + z2;   // gradient on the stage
    switch (kase) {
      case RC_FREE3: {
        // L D L^T of Quu = M + R~ without pivoting, pivots made positive
        const T Q00 = S00 + c00, Q01 = S01 + c01, Q02 = Z02 + c02, Q11 = S11 + c11, Q12 = Z12 + c12, Q22 = Z22 + c22;
        const T delta = ric_max((T)1e-6 * ric_max(ric_abs(Q00), ric_max(ric_abs(Q11), ric_abs(Q22))), (T)1e-30);
        const T d0 = ric_pivot(Q00, delta), i0 = ric_rcp(d0);
        const T l10 = Q01 * i0, l20 = Q02 * i0;
        const T d1 = ric_pivot(Q11 - l10 * Q01, delta), i1 = ric_rcp(d1);
        const T h12 = Q12 - l20 * Q01;
        const T l21 = h12 * i1;
        const T d2 = ric_pivot(Q22 - l20 * Q02 - l21 * h12, delta), i2 = ric_rcp(d2);
#define NEO_RIC_SOLVE3(r0_, r1_, r2_, x0_, x1_, x2_)                                           \
        {                                                                                        \
          const T y0 = -(r0_), y1 = -(r1_) - l10 * y0, y2 = -(r2_) - l20 * y0 - l21 * y1;        \
          x2_ = y2 * i2;                                                                         \
          x1_ = y1 * i1 - l21 * x2_;                                                             \
          x0_ = y0 * i0 - l10 * x1_ - l20 * x2_;                                                 \
        }
        NEO_RIC_SOLVE3(q0, q1, q2, k0, k1, k2)
        if (first) break;   // (the first stage: no gains, no value function behind it -- see the top of the loop)
        NEO_RIC_SOLVE3(S00, S01, M02, K00, K10, K20)
        NEO_RIC_SOLVE3(S01, S11, M12, K01, K11, K21)
        NEO_RIC_SOLVE3(Z02, Z12, Z22, K02, K12, K22)
#undef NEO_RIC_SOLVE3
        v0 = z0 + S00 * k0 + S01 * k1 + M02 * k2;
        v1 = z1 + S01 * k0 + S11 * k1 + M12 * k2;
        v2 = z2 + Z02 * k0 + Z12 * k1 + Z22 * k2;
        V00 = S00 + S00 * K00 + S01 * K10 + M02 * K20;
        V01 = S01 + S00 * K01 + S01 * K11 + M02 * K21;
        V02 = M02 + S00 * K02 + S01 * K12 + M02 * K22;
        V11 = S11 + S01 * K01 + S11 * K11 + M12 * K21;
        V12 = M12 + S01 * K02 + S11 * K12 + M12 * K22;
        V22 = Z22 + Z02 * K02 + Z12 * K12 + Z22 * K22;
        break;
      }
      case RC_SLIDE_W:     // a = tangent (rtx, rty, 0), b = omega
      case RC_XY: {        // a = x axis, b = y axis (omega frozen)
        const bool bw = kase == RC_SLIDE_W;
        const T ax = bw ? rtx : (T)1.0, ay = bw ? rty : (T)0.0;
        const T Q00 = S00 + c00, Q01 = S01 + c01, Q11 = S11 + c11;
        const T Qa0 = ax * Q00 + ay * Q01, Qa1 = ax * Q01 + ay * Q11;
        const T haa = Qa0 * ax + Qa1 * ay;
        const T hab = bw ? ax * (Z02 + c02) + ay * (Z12 + c12) : Qa1;
        const T hbb = bw ? Z22 + c22 : Q11;
        const T ga = ax * q0 + ay * q1, gb = bw ? q2 : q1;
        const T Za0 = ax * S00 + ay * S01, Za1 = ax * S01 + ay * S11, Za2 = ax * Z02 + ay * Z12;
        const T Zb0 = bw ? M02 : S01, Zb1 = bw ? M12 : S11, Zb2 = bw ? Z22 : Z12;
        const T delta = ric_max((T)1e-6 * ric_max(ric_abs(haa), ric_abs(hbb)), (T)1e-30);
        const T d0 = ric_pivot(haa, delta), i0 = ric_rcp(d0);
        const T l = hab * i0;
        const T d1 = ric_pivot(hbb - l * hab, delta), i1 = ric_rcp(d1);
        T ka, kb, Ka0, Ka1, Ka2, Kb0, Kb1, Kb2;
#define NEO_RIC_SOLVE2(ra_, rb_, xa_, xb_)                       \
        {                                                          \
          const T y0 = -(ra_), y1 = -(rb_) - l * y0;               \
          xb_ = y1 * i1;                                           \
          xa_ = y0 * i0 - l * xb_;                                 \
        }
        NEO_RIC_SOLVE2(ga, gb, ka, kb)
        if (first) {   // (the first stage: the step alone)
          k0 = ax * ka; k1 = ay * ka;
          if (bw) k2 = kb; else k1 = kb;
          break;
        }
        NEO_RIC_SOLVE2(Za0, Zb0, Ka0, Kb0)
        NEO_RIC_SOLVE2(Za1, Zb1, Ka1, Kb1)
        NEO_RIC_SOLVE2(Za2, Zb2, Ka2, Kb2)
#undef NEO_RIC_SOLVE2
        v0 = z0 + Za0 * ka + Zb0 * kb; v1 = z1 + Za1 * ka + Zb1 * kb; v2 = z2 + Za2 * ka + Zb2 * kb;
        V00 = S00 + Za0 * Ka0 + Zb0 * Kb0; V01 = S01 + Za0 * Ka1 + Zb0 * Kb1; V02 = M02 + Za0 * Ka2 + Zb0 * Kb2;
        V11 = S11 + Za1 * Ka1 + Zb1 * Kb1; V12 = M12 + Za1 * Ka2 + Zb1 * Kb2; V22 = Z22 + Za2 * Ka2 + Zb2 * Kb2;
        k0 = ax * ka; k1 = ay * ka; K00 = ax * Ka0; K01 = ax * Ka1; K02 = ax * Ka2; K10 = ay * Ka0; K11 = ay * Ka1; K12 = ay * Ka2;
        if (bw) { k2 = kb; K20 = Kb0; K21 = Kb1; K22 = Kb2; }
        else { k1 = kb; K10 = Kb0; K11 = Kb1; K12 = Kb2; }
        break;
      }
      case RC_SLIDE: {     // the tangent only (omega frozen)
        const T Q00 = S00 + c00, Q01 = S01 + c01, Q11 = S11 + c11;
        const T haa = (rtx * Q00 + rty * Q01) * rtx + (rtx * Q01 + rty * Q11) * rty;
        const T i0 = -ric_rcp(ric_pivot(haa, ric_max((T)1e-6 * ric_abs(haa), (T)1e-30)));
        const T Za0 = rtx * S00 + rty * S01, Za1 = rtx * S01 + rty * S11, Za2 = rtx * Z02 + rty * Z12;
        const T ka = (rtx * q0 + rty * q1) * i0, Ka0 = Za0 * i0, Ka1 = Za1 * i0, Ka2 = Za2 * i0;
        if (first) { k0 = rtx * ka; k1 = rty * ka; break; }
        v0 = z0 + Za0 * ka; v1 = z1 + Za1 * ka; v2 = z2 + Za2 * ka;
        V00 = S00 + Za0 * Ka0; V01 = S01 + Za0 * Ka1; V02 = M02 + Za0 * Ka2;
        V11 = S11 + Za1 * Ka1; V12 = M12 + Za1 * Ka2; V22 = Z22 + Za2 * Ka2;
        k0 = rtx * ka; k1 = rty * ka; K00 = rtx * Ka0; K01 = rtx * Ka1; K02 = rtx * Ka2; K10 = rty * Ka0; K11 = rty * Ka1; K12 = rty * Ka2;
        break;
      }
      case RC_W: {         // omega only (velocity pinned)
        const T hbb = Z22 + c22;
        const T i0 = -ric_rcp(ric_pivot(hbb, ric_max((T)1e-6 * ric_abs(hbb), (T)1e-30)));
        k2 = q2 * i0;
        if (first) break;
        K20 = M02 * i0; K21 = M12 * i0; K22 = Z22 * i0;
        v0 = z0 + M02 * k2; v1 = z1 + M12 * k2; v2 = z2 + Z22 * k2;
        V00 = S00 + M02 * K20; V01 = S01 + M02 * K21; V02 = M02 + M02 * K22;
        V11 = S11 + M12 * K21; V12 = M12 + M12 * K22; V22 = Z22 + Z22 * K22;
        break;
      }
      case RC_KINK:        // fixed step onto the kink, no feedback
        k0 = e0; k1 = e1; k2 = e2;
        if (first) break;
        v0 = z0 + S00 * k0 + S01 * k1 + M02 * k2;
        v1 = z1 + S01 * k0 + S11 * k1 + M12 * k2;
        v2 = z2 + Z02 * k0 + Z12 * k1 + Z22 * k2;
        V00 = S00; V01 = S01; V02 = M02; V11 = S11; V12 = M12; V22 = Z22;
        break;
      default:             // RC_NONE: nothing moves
        v0 = z0; v1 = z1; v2 = z2;
        V00 = S00; V01 = S01; V02 = M02; V11 = S11; V12 = M12; V22 = Z22;
        break;
    }
    if (lane == 0) {   // the gains take the place of the stage's linear terms
      ric_f4* G4 = reinterpret_cast<ric_f4*>(RS + kRicStage * i + RS_GAIN);
      G4[0] = ric_f4{(float)K00, (float)K01, (float)K02, (float)K10};
      G4[1] = ric_f4{(float)K11, (float)K12, (float)K20, (float)K21};
      G4[2] = ric_f4{(float)K22, (float)k0, (float)k1, (float)k2};
      AMODE[4 * i + 3] = tokink ? 1 : 0;
    }
  }
  WAVE_SYNC();
  // forward sweep: w_i = k_i + K_i dz_{i-1}, dz_i = A_i (dz_{i-1} + w_i)
  T z0 = (T)0.0, z1 = (T)0.0, z2 = (T)0.0;
  for (int i = 0; i < n; ++i) {
    const ric_f4* R4 = reinterpret_cast<const ric_f4*>(RS + kRicStage * i);
    const ric_f4 s0 = R4[0], g0 = R4[4], g1 = R4[5], g2 = R4[6];
    const T w0 = (T)g2.y + (T)g0.x * z0 + (T)g0.y * z1 + (T)g0.z * z2;
    const T w1 = (T)g2.z + (T)g0.w * z0 + (T)g1.x * z1 + (T)g1.y * z2;
    const T w2f = (T)g2.w + (T)g1.z * z0 + (T)g1.w * z1 + (T)g2.x * z2;
    if (lane == 0) { d[3 * i] = (double)w0; d[3 * i + 1] = (double)w1; d[3 * i + 2] = (double)w2f; }
    const T e0 = z0 + w0, e1 = z1 + w1, e2 = z2 + w2f;
    z0 = ric_fma(-(T)s0.y, e2, e0); z1 = ric_fma((T)s0.x, e2, e1); z2 = e2;
  }
  WAVE_SYNC();
}

// lane = stage: du = B0^-1 w = (Rot^T w_xy, w_w) / dt; the step of a block sent onto the kink is exact
__device__ __forceinline__ void riccati_finish(const SolveArgs& a, const Ctx& c, double* L, int n, int lane) {
  if (lane < n) {
    const float* rs = reinterpret_cast<const float*>(L + a.lds.ric) + kRicStage * lane;
    const int* am = reinterpret_cast<const int*>(L + a.lds.mode) + 4 * lane;
    double* d = L + a.lds.d + 3 * lane;
    if (am[3]) {
      const double* u = L + a.lds.u + 3 * lane;
      d[0] = c.v0 - u[0]; d[1] = c.v1 - u[1]; d[2] = c.v2 - u[2];
    } else {
      const double cs = rs[RS_CS], sn = rs[RS_SN], idt = rcp_fast(a.p.dt);
      const double w0 = d[0], w1 = d[1], w2 = d[2];
      double d0 = (cs * w0 + sn * w1) * idt, d1 = (-sn * w0 + cs * w1) * idt;
      // back on the face EXACTLY: the rotation round trip leaves ~1e-10 of a sliding block's step along the
      // constraint normal; an inward residue would take the block off its bound, the next tangent-cone pass
      // would find it free, and the Newton step would run straight back into the bound
      const int mode = am[2] ? 2 : am[0];
      if (mode == 1) {
        const double nx = L[a.lds.nx + lane], ny = L[a.lds.ny + lane];
        const double dot = d0 * nx + d1 * ny;
        d0 -= dot * nx; d1 -= dot * ny;
      } else if (mode == 2) { d0 = 0.0; d1 = 0.0; }
      d[0] = d0; d[1] = d1; d[2] = (am[2] || (am[1] & 1)) ? 0.0 : w2 * idt;
    }
  }
  WAVE_SYNC();
}

}  // namespace
}  // namespace neo_mpc
