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
// differences.  The sweep is wave-uniform: every lane runs it on the same LDS records (written one
// stage per lane by the passes in front of it), so nothing crosses lanes inside the recursion.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "neo_mpc_device.h"
#include "fast_math.h"
#include "solver_context.h"

namespace neo_mpc {
namespace {

constexpr int kRicCurv = 6;    // doubles per stage: block curvature R_i (symmetric 3x3: 00 01 02 11 12 22)
constexpr int kRicGain = 12;   // doubles per stage: feedback K_i (3x3, row-major) and feed-forward k_i
constexpr int kRicWall = 3;    // doubles per stage: wall-sliding penalty on the stage position (xx xy yy), costmap.h

// non-positive pivots are replaced (the exact Hessian is indefinite away from the minimiser): the
// factorisation is then that of a positive definite matrix and the stage step a descent direction
__device__ __forceinline__ double ric_pivot(double p, double delta) { return p > delta ? p : fmax(fabs(p), delta); }

// Backward + forward sweep.  Inputs per stage i (LDS, doubles): cs/sn/px/py = rollout trigonometry and
// position increments, ax/ay/kap = costate-weighted second-order terms of the step, curv = R_i, wall =
// the wall-sliding penalty on the stage position,
// gt = total gradient (0 on blocks next to the kink), gs = smooth gradient, u, and the tangent-cone
// description (mode, wfroz, near | nx, ny).  Output: d (3N) and, per block, tokink (AMODE slot 3): the
// stage model's minimiser is the kink u_i = v_cur itself, d_i = v_cur - u_i.
template <bool kTame>
__device__ __forceinline__ void riccati_direction(const SolveArgs& a, const Ctx& c, double* L, int n, int lane,
                                                  bool v_feasible) {
  const DevParams& p = a.p;
  const double dt = p.dt;
  const double* ACS = L + a.lds.cs;
  const double* ASN = L + a.lds.sn;
  const double* APX = L + a.lds.dxs;
  const double* APY = L + a.lds.dys;
  const double* AAX = L + a.lds.rx;
  const double* AAY = L + a.lds.ry;
  const double* AKP = L + a.lds.rt;
  const double* ANX = L + a.lds.nx;
  const double* ANY = L + a.lds.ny;
  const double* CURV = L + a.lds.ric;
  double* GAIN = L + a.lds.ric + kRicCurv * n;
  const double* WALL = GAIN + kRicGain * n;
  int* AMODE = reinterpret_cast<int*>(L + a.lds.mode);
  const double* u = L + a.lds.u;
  const double* gs = L + a.lds.gs;
  const double* gt = L + a.lds.gt;
  double* d = L + a.lds.d;
  const double w2 = 2.0 * p.wt_n, wo2 = 2.0 * p.wo_n;
  const double wc2 = p.wc_n * p.wc_n;

  double V00 = 0.0, V01 = 0.0, V02 = 0.0, V11 = 0.0, V12 = 0.0, V22 = 0.0, v0 = 0.0, v1 = 0.0, v2 = 0.0;
  for (int i = n - 1; i >= 0; --i) {
    const double cs = ACS[i], sn = ASN[i], px = APX[i], py = APY[i];
    // S = W_i + V; M = A^T S A with A = [[1 0 -py] [0 1 px] [0 0 1]]
    const double* wl = WALL + kRicWall * i;
    const double S00 = V00 + w2 + wl[0], S01 = V01 + wl[1], S02 = V02, S11 = V11 + w2 + wl[2], S12 = V12;
    const double S22 = V22 + wo2 + (i == n - 1 ? 2.0 * p.wterm_o : 0.0);
    const double M02 = fma(-py, S00, fma(px, S01, S02));
    const double M12 = fma(-py, S01, fma(px, S11, S12));
    const double M22 = fma(-py, M02, fma(px, M12, fma(-py, S02, fma(px, S12, S22))));
    // B = A B0, B0 = dt [[cs -sn 0] [sn cs 0] [0 0 1]]:  G = B0^T M (= Quz before the second-order terms)
    const double dc = dt * cs, ds = dt * sn;
    const double G00 = dc * S00 + ds * S01, G01 = dc * S01 + ds * S11, G02 = dc * M02 + ds * M12;
    const double G10 = dc * S01 - ds * S00, G11 = dc * S11 - ds * S01, G12 = dc * M12 - ds * M02;
    const double G20 = dt * M02, G21 = dt * M12, G22 = dt * M22;
    // Quu = G B0 (symmetric) + second-order terms + block curvature
    const double ax = AAX[i], ay = AAY[i], kap = AKP[i];
    double Q00 = dc * G00 + ds * G01, Q01 = dc * G01 - ds * G00, Q02 = dt * G02 + dt * ax;
    double Q11 = dc * G11 - ds * G10, Q12 = dt * G12 + dt * ay, Q22 = dt * G22 + dt * dt * kap;
    // Quz = G + second-order terms (column theta); Qzz = M + kappa e_theta e_theta^T
    const double Z02 = G02 + ax, Z12 = G12 + ay, Z22 = G22 + dt * kap;
    const double Y22 = M22 + kap;
    // linear terms: Qz = A^T v, Qu = gt_i + B0^T Qz
    const double z0 = v0, z1 = v1, z2 = fma(-py, v0, fma(px, v1, v2));
    const double b0 = dc * z0 + ds * z1, b1 = dc * z1 - ds * z0, b2 = dt * z2;
    const double u0 = u[3 * i], u1 = u[3 * i + 1], u2 = u[3 * i + 2];
    const int mode = AMODE[4 * i], wfroz = AMODE[4 * i + 1], near = AMODE[4 * i + 2];
    double k0, k1, k2, K00, K01, K02, K10, K11, K12, K20, K21, K22;
    bool tokink = false;
    if (!near && v_feasible) {
      // does the stage model put this block ON the kink?  0 in Qu_s + Quu_s k + w d|u_i + k - v| at
      // k = v - u_i  <=>  |Qu_s + Quu_s (v - u_i)| <= w  (smooth parts only)
      const double e0 = c.v0 - u0, e1 = c.v1 - u1, e2 = c.v2 - u2;
      const double r0 = gs[3 * i] + b0 + Q00 * e0 + Q01 * e1 + Q02 * e2;
      const double r1 = gs[3 * i + 1] + b1 + Q01 * e0 + Q11 * e1 + Q12 * e2;
      const double r2 = gs[3 * i + 2] + b2 + Q02 * e0 + Q12 * e1 + Q22 * e2;
      tokink = r0 * r0 + r1 * r1 + r2 * r2 <= wc2;
    }
    if (tokink) {   // fixed step onto the kink, no feedback: v = Qz + Quz^T k, V = Qzz
      k0 = c.v0 - u0; k1 = c.v1 - u1; k2 = c.v2 - u2;
      K00 = K01 = K02 = K10 = K11 = K12 = K20 = K21 = K22 = 0.0;
      v0 = z0 + G00 * k0 + G10 * k1 + G20 * k2;
      v1 = z1 + G01 * k0 + G11 * k1 + G21 * k2;
      v2 = z2 + Z02 * k0 + Z12 * k1 + Z22 * k2;
      V00 = S00; V01 = S01; V02 = M02; V11 = S11; V12 = M12; V22 = Y22;
    } else {
      const double* cv = CURV + kRicCurv * i;
      Q00 += cv[0]; Q01 += cv[1]; Q02 += cv[2]; Q11 += cv[3]; Q12 += cv[4]; Q22 += cv[5];
      // face projector P = [[p00 p01 0] [p01 p11 0] [0 0 pw]]
      double p00, p01, p11;
      if (near || mode == 2) { p00 = 0.0; p01 = 0.0; p11 = 0.0; }
      else if (mode == 1) { const double nx = ANX[i], ny = ANY[i]; p00 = 1.0 - nx * nx; p01 = -nx * ny; p11 = 1.0 - ny * ny; }
      else { p00 = 1.0; p01 = 0.0; p11 = 1.0; }
      const double pw = (near || wfroz) ? 0.0 : 1.0;
      // rhs and Quz on the face
      const double q0 = gt[3 * i] + b0, q1 = gt[3 * i + 1] + b1, q2 = gt[3 * i + 2] + b2;
      const double g0 = p00 * q0 + p01 * q1, g1 = p01 * q0 + p11 * q1, g2 = pw * q2;
      const double R00 = p00 * G00 + p01 * G10, R01 = p00 * G01 + p01 * G11, R02 = p00 * Z02 + p01 * Z12;
      const double R10 = p01 * G00 + p11 * G10, R11 = p01 * G01 + p11 * G11, R12 = p01 * Z02 + p11 * Z12;
      const double R20 = pw * G20, R21 = pw * G21, R22 = pw * Z22;
      // P Quu P + (I - P)
      const double T00 = p00 * Q00 + p01 * Q01, T01 = p00 * Q01 + p01 * Q11, T02 = p00 * Q02 + p01 * Q12;
      const double T10 = p01 * Q00 + p11 * Q01, T11 = p01 * Q01 + p11 * Q11, T12 = p01 * Q02 + p11 * Q12;
      const double H00 = T00 * p00 + T01 * p01 + (1.0 - p00), H01 = T00 * p01 + T01 * p11 - p01, H02 = T02 * pw;
      const double H11 = T10 * p01 + T11 * p11 + (1.0 - p11), H12 = T12 * pw;
      const double H22 = pw * Q22 * pw + (1.0 - pw);
      // L D L^T without pivoting, pivots made positive
      const double delta = fmax(1e-6 * fmax(fabs(H00), fmax(fabs(H11), fabs(H22))), 1e-30);
      const double d0 = ric_pivot(H00, delta), i0 = rcp_fast(d0);
      const double l10 = H01 * i0, l20 = H02 * i0;
      const double d1 = ric_pivot(H11 - l10 * H01, delta), i1 = rcp_fast(d1);
      const double h12 = H12 - l20 * H01;
      const double l21 = h12 * i1;
      const double d2 = ric_pivot(H22 - l20 * H02 - l21 * h12, delta), i2 = rcp_fast(d2);
      // solve (L D L^T) x = -r for r = g and the three columns of Quz on the face
#define NEO_RIC_SOLVE(r0_, r1_, r2_, x0_, x1_, x2_)                     \
      {                                                                  \
        const double y0 = -(r0_), y1 = -(r1_) - l10 * y0, y2 = -(r2_) - l20 * y0 - l21 * y1; \
        x2_ = y2 * i2;                                                   \
        x1_ = y1 * i1 - l21 * x2_;                                       \
        x0_ = y0 * i0 - l10 * x1_ - l20 * x2_;                           \
      }
      NEO_RIC_SOLVE(g0, g1, g2, k0, k1, k2)
      NEO_RIC_SOLVE(R00, R10, R20, K00, K10, K20)
      NEO_RIC_SOLVE(R01, R11, R21, K01, K11, K21)
      NEO_RIC_SOLVE(R02, R12, R22, K02, K12, K22)
#undef NEO_RIC_SOLVE
      // v = Qz + Quz_r^T k, V = Qzz + Quz_r^T K (symmetrised)
      v0 = z0 + R00 * k0 + R10 * k1 + R20 * k2;
      v1 = z1 + R01 * k0 + R11 * k1 + R21 * k2;
      v2 = z2 + R02 * k0 + R12 * k1 + R22 * k2;
      V00 = S00 + R00 * K00 + R10 * K10 + R20 * K20;
      V11 = S11 + R01 * K01 + R11 * K11 + R21 * K21;
      V22 = Y22 + R02 * K02 + R12 * K12 + R22 * K22;
      V01 = S01 + 0.5 * ((R00 * K01 + R10 * K11 + R20 * K21) + (R01 * K00 + R11 * K10 + R21 * K20));
      V02 = M02 + 0.5 * ((R00 * K02 + R10 * K12 + R20 * K22) + (R02 * K00 + R12 * K10 + R22 * K20));
      V12 = M12 + 0.5 * ((R01 * K02 + R11 * K12 + R21 * K22) + (R02 * K01 + R12 * K11 + R22 * K21));
    }
    if (lane == 0) {
      double* gn = GAIN + kRicGain * i;
      gn[0] = K00; gn[1] = K01; gn[2] = K02; gn[3] = K10; gn[4] = K11; gn[5] = K12;
      gn[6] = K20; gn[7] = K21; gn[8] = K22; gn[9] = k0; gn[10] = k1; gn[11] = k2;
      AMODE[4 * i + 3] = tokink ? 1 : 0;
    }
  }
  WAVE_SYNC();
  // forward sweep of the linearised chain: du_i = k_i + K_i dz_{i-1}, dz_i = A_i (dz_{i-1} + B0_i du_i)
  double z0 = 0.0, z1 = 0.0, z2 = 0.0;
  for (int i = 0; i < n; ++i) {
    const double* gn = GAIN + kRicGain * i;
    const double du0 = gn[9] + gn[0] * z0 + gn[1] * z1 + gn[2] * z2;
    const double du1 = gn[10] + gn[3] * z0 + gn[4] * z1 + gn[5] * z2;
    const double du2 = gn[11] + gn[6] * z0 + gn[7] * z1 + gn[8] * z2;
    if (lane == 0) { d[3 * i] = du0; d[3 * i + 1] = du1; d[3 * i + 2] = du2; }
    const double cs = ACS[i], sn = ASN[i], px = APX[i], py = APY[i];
    const double e0 = z0 + dt * (cs * du0 - sn * du1), e1 = z1 + dt * (sn * du0 + cs * du1), e2 = z2 + dt * du2;
    z0 = fma(-py, e2, e0); z1 = fma(px, e2, e1); z2 = e2;
  }
  WAVE_SYNC();
}

}  // namespace
}  // namespace neo_mpc
