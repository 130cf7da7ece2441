// wave_ops.h -- cross-lane primitives of a 64-lane wavefront on the DPP network
// Part of libneo_mpc.so's device code (included by neo_mpc_kernels.hip only).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "neo_mpc_device.h"

namespace neo_mpc {
namespace {

#define WAVE_SYNC() __syncthreads()

// ---------------------------------------------------------------- wave primitives
// Cross-lane reductions on the DPP network (row_shr within the 16-lane rows, row_bcast across
// rows), not through LDS (`__shfl` lowers to ds_bpermute, ~100 cycles per hop): the L-BFGS
// recursion is a chain of dependent dot products, so the reduction latency is on the critical path.
template <int kCtrl, int kRowMask, bool kZeroFill>
__device__ __forceinline__ double dpp_move(double v, double fill) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int flo = __double2loint(fill), fhi = __double2hiint(fill);
  const int rlo = __builtin_amdgcn_update_dpp(flo, lo, kCtrl, kRowMask, 0xf, kZeroFill);
  const int rhi = __builtin_amdgcn_update_dpp(fhi, hi, kCtrl, kRowMask, 0xf, kZeroFill);
  return __hiloint2double(rhi, rlo);
}
__device__ __forceinline__ double lane_value(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// every lane returns the sum over the 64 lanes (bitwise identical in all lanes)
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_move<0x111, 0xf, true>(v, 0.0);  // row_shr:1
  v += dpp_move<0x112, 0xf, true>(v, 0.0);  // row_shr:2
  v += dpp_move<0x114, 0xf, true>(v, 0.0);  // row_shr:4
  v += dpp_move<0x118, 0xf, true>(v, 0.0);  // row_shr:8   -> lane 15 of each row: row sum
  v += dpp_move<0x142, 0xa, false>(v, 0.0); // row_bcast:15 into rows 1 and 3
  v += dpp_move<0x143, 0xc, false>(v, 0.0); // row_bcast:31 into rows 2 and 3 -> lane 63: total
  return lane_value(v, 63);
}
// inclusive prefix sum over the lanes (lane i: sum of lanes 0..i) -- the same DPP ladder as wave_sum
__device__ __forceinline__ double wave_scan(double v) {
  v += dpp_move<0x111, 0xf, true>(v, 0.0);
  v += dpp_move<0x112, 0xf, true>(v, 0.0);
  v += dpp_move<0x114, 0xf, true>(v, 0.0);
  v += dpp_move<0x118, 0xf, true>(v, 0.0);
  v += dpp_move<0x142, 0xa, false>(v, 0.0);
  v += dpp_move<0x143, 0xc, false>(v, 0.0);
  return v;
}
// ... over the first kFew lanes only (kFew a compile-time bound; the other lanes hold zeros): the ladder stops where the
// lanes it would still reach hold nothing -- two DPP steps for three stages instead of six.  Lanes < kFew get, bit for bit,
// what wave_scan gives them (the steps left out add zero-filled values there).
template <int kFew>
__device__ __forceinline__ double wave_scan_few(double v) {
  static_assert(kFew >= 1 && kFew <= 16, "one DPP row");
  if (kFew > 1) v += dpp_move<0x111, 0xf, true>(v, 0.0);
  if (kFew > 2) v += dpp_move<0x112, 0xf, true>(v, 0.0);
  if (kFew > 4) v += dpp_move<0x114, 0xf, true>(v, 0.0);
  if (kFew > 8) v += dpp_move<0x118, 0xf, true>(v, 0.0);
  return v;
}
// sum over the first kFew lanes (the other lanes hold zeros), every lane returns it
template <int kFew>
__device__ __forceinline__ double wave_sum_few(double v) { return lane_value(wave_scan_few<kFew>(v), kFew - 1); }
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_move<0x111, 0xf, false>(v, v));
  v = fmax(v, dpp_move<0x112, 0xf, false>(v, v));
  v = fmax(v, dpp_move<0x114, 0xf, false>(v, v));
  v = fmax(v, dpp_move<0x118, 0xf, false>(v, v));
  v = fmax(v, dpp_move<0x142, 0xa, false>(v, v));
  v = fmax(v, dpp_move<0x143, 0xc, false>(v, v));
  return lane_value(v, 63);
}
__device__ __forceinline__ double wave_min(double v) { return -wave_max(-v); }
// maximum of non-negative float32 values (step lengths, pivots: compared against tolerances, single
// precision is plenty): one DPP-fused v_max_f32 per hop instead of two moves and a 64-bit max
template <int kCtrl, int kRowMask>
__device__ __forceinline__ float dpp_move_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v),
                                                              kCtrl, kRowMask, 0xf, false));
}
__device__ __forceinline__ float wave_max_f(float v) {
  v = fmaxf(v, dpp_move_f<0x111, 0xf>(v));
  v = fmaxf(v, dpp_move_f<0x112, 0xf>(v));
  v = fmaxf(v, dpp_move_f<0x114, 0xf>(v));
  v = fmaxf(v, dpp_move_f<0x118, 0xf>(v));
  v = fmaxf(v, dpp_move_f<0x142, 0xa>(v));
  v = fmaxf(v, dpp_move_f<0x143, 0xc>(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// lowest value, ties to the lowest lane; every lane returns the same pair
__device__ __forceinline__ void wave_argmin(double& v, int& idx) {
  const double m = wave_min(v);
  const unsigned long long hit = __ballot(v == m);
  idx = (int)__ffsll((long long)hit) - 1;
  v = m;
}
__device__ __forceinline__ int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }

}  // namespace
}  // namespace neo_mpc
