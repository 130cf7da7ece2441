// fast_math.h -- float64 elementary functions sized for the solver loop (gfx950)
// Part of libneo_mpc.so's device code (included by neo_mpc_kernels.hip only).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace neo_mpc {
namespace {

// sin and cos of a moderate argument: three-term Cody-Waite reduction by pi/2 (exact products for
// |x| < 1e6) and the degree-13/14 minimax kernels on [-pi/4, pi/4]; ~40 f64 operations, no
// table, no branch, no call -- the library sincos carries a Payne-Hanek path the rollout never needs.
// 1/x by v_rcp_f64 and two Newton steps (error ~1 ulp, no scaling/fix-up for denormals or
// infinities): used where only a search direction or a unit vector depends on it
__device__ __forceinline__ double rcp_fast(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// a * b + c as ONE three-address v_fma_f64.  Left to itself the compiler turns the Horner steps of the polynomial kernels
// below into v_mov_b64 (copy the coefficient) + v_fmac_f64 (two-address multiply-accumulate into the copy): the coefficients
// live across the solver loop, so every step pays a 64-bit move -- ten per sincos, sixty per solver iteration of the
// control_steps-3 kernel (tools/opcode_histogram.py: v_mov_b64 was the third most frequent vector opcode of the loop).
__device__ __forceinline__ double fma3(double a, double b, double c) {
  // (the addend -- a literal coefficient -- in a SCALAR register pair: two s_mov_b32 on the scalar unit instead of sixteen
  // loop-resident vector registers of coefficients)
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
  return r;
}

// 1/sqrt(x), x > 0, to ~1e-13 relative: v_rsq_f64 (2^-26.x, tools/rsq_check.hip) and ONE Newton step -- for quantities that
// place a candidate (the prox step's shrink factor, the radial projection onto the speed disc): the objective of every
// candidate is evaluated at the point it actually is
__device__ __forceinline__ double rsq_fast1(double x) {
  double r = __builtin_amdgcn_rsq(x);
  return r * fma(fma(-x * r, r, 1.0), 0.5, 1.0);
}

// 1/sqrt(x), x > 0: v_rsq_f64 and two Newton steps
__device__ __forceinline__ double rsq_fast(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * fma(fma(-x * r, r, 1.0), 0.5, 1.0);
  r = r * fma(fma(-x * r, r, 1.0), 0.5, 1.0);
  return r;
}

// sqrt(x), x >= 0 and not huge: v_rsq_f64, one coupled Goldschmidt step and one residual correction.
// Measured on gfx950 against the correctly rounded root (tools/sqrt_check.hip, 4 M arguments over
// 2^-300..2^300): identical except 1 ulp on a denormal; 11 operations where the library's
// sequence (input scaling, a second correction, class test) takes 18.
__device__ __forceinline__ double sqrt_fast(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  g = fma(fma(-g, g, x), h, g);
  return x > 0.0 ? g : 0.0;
}

__device__ __forceinline__ void sincos_fast(double x, double* sn, double* cs) {
  // (|x| beyond ~1e6 -- never produced by a feasible rollout, |theta| <= max_vel_theta * horizon
  // plus a yaw -- only loses accuracy; a non-finite x gives NaN, which the arc search discards)
  const double k = rint(x * 6.36619772367581382433e-01);
  double r = fma(-k, 1.57079632673412561417e+00, x);
  r = fma(-k, 6.07710050630396597660e-11, r);
  r = fma(-k, 2.02226624879595063154e-21, r);
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma3(z, ps, 2.75573137070700676789e-06);
  ps = fma3(z, ps, -1.98412698298579493134e-04);
  ps = fma3(z, ps, 8.33333333332248946124e-03);
  ps = fma3(z, ps, -1.66666666666666324348e-01);
  const double sr = fma(z * r, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma3(z, pc, -2.75573143513906633035e-07);
  pc = fma3(z, pc, 2.48015872894767294178e-05);
  pc = fma3(z, pc, -1.38888888888741095749e-03);
  pc = fma3(z, pc, 4.16666666666666019037e-02);
  const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
  const int q = (int)fmin(fmax(k, -2.0e9), 2.0e9) & 3;
  const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
  *sn = (q & 2) ? -s0 : s0;
  *cs = ((q + 1) & 2) ? -c0 : c0;
}

// The same two kernels without the reduction, for |x| <= pi/4: 17 operations.  The rollout's heading
// never leaves that range when max|omega| * prediction_horizon <= 0.78 (DevParams.tame; 0.56 with
// the README's parameters).
__device__ __forceinline__ void sincos_small(double r, double* sn, double* cs) {
  // (the two Horner chains step by step side by side: each step of one is independent of the other's, so a wave that is
  // alone with its latency -- the tail of a one-round launch -- issues them back to back)
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  ps = fma3(z, ps, 2.75573137070700676789e-06);
  pc = fma3(z, pc, -2.75573143513906633035e-07);
  ps = fma3(z, ps, -1.98412698298579493134e-04);
  pc = fma3(z, pc, 2.48015872894767294178e-05);
  ps = fma3(z, ps, 8.33333333332248946124e-03);
  pc = fma3(z, pc, -1.38888888888741095749e-03);
  ps = fma3(z, ps, -1.66666666666666324348e-01);
  pc = fma3(z, pc, 4.16666666666666019037e-02);
  *sn = fma(z * r, ps, r);
  *cs = fma(z * z, pc, fma(z, -0.5, 1.0));
}
template <bool kTame>
__device__ __forceinline__ void sincos_heading(double th, double* sn, double* cs) {
  if (kTame) sincos_small(th, sn, cs);
  else sincos_fast(th, sn, cs);
}

// py:176-178
__device__ __forceinline__ double yaw_of(const double* q) {
  double t3 = 2.0 * (q[3] * q[2] + q[0] * q[1]);
  double t4 = 1.0 - 2.0 * (q[1] * q[1] + q[2] * q[2]);
  return atan2(t3, t4);
}

}  // namespace
}  // namespace neo_mpc
