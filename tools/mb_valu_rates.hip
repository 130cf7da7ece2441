// VALU issue rates and dependent-issue latencies on gfx950, one-wave workgroups at 1 / 2 / 4 waves per SIMD (study):
//   hipcc --offload-arch=gfx950 -O2 -w -o /tmp/mb_valu_rates tools/mb_valu_rates.hip && /tmp/mb_valu_rates
// (results: profiles/r03_b_ab_experiments.txt)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void k(float* out, int iters, float s) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 b0 = {a0, a1}, b1 = {a2, a3}, b2 = {a4, a5}, b3 = {a6, a7}, b4 = {a1, a0}, b5 = {a3, a2}, b6 = {a5, a4}, b7 = {a7, a6};
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
  const f2 sv = {s, s};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#define F(x) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(s));
      F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
    } else if (MODE == 1) {
#define F(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(sv));
      F(b0) F(b1) F(b2) F(b3) F(b4) F(b5) F(b6) F(b7)
#undef F
    } else if (MODE == 2) {
#define F(x) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x) : "v"((double)s));
      F(d0) F(d1) F(d2) F(d3) F(d4) F(d5) F(d6) F(d7)
#undef F
    } else if (MODE == 3) {
#define F(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(sv));
      F(b0) F(b1) F(b2) F(b3) F(b4) F(b5) F(b6) F(b7)
#undef F
    } else if (MODE == 4) {
#define F(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(sv));
      F(b0) F(b1) F(b2) F(b3) F(b4) F(b5) F(b6) F(b7)
#undef F
    } else if (MODE == 5) {  // dependent chain, one accumulator: latency
#define F(x) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(s));
      F(a0) F(a0) F(a0) F(a0) F(a0) F(a0) F(a0) F(a0)
#undef F
    } else if (MODE == 6) {
#define F(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(sv));
      F(b0) F(b0) F(b0) F(b0) F(b0) F(b0) F(b0) F(b0)
#undef F
    } else if (MODE == 8) {
#define F(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(i));
      F(i0) F(i1) F(i2) F(i3) F(i4) F(i5) F(i6) F(i7)
#undef F
    } else if (MODE == 9) {
#define F(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(s));
      F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
    } else if (MODE == 10) {
#define F(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"((double)s));
      F(d0) F(d1) F(d2) F(d3) F(d4) F(d5) F(d6) F(d7)
#undef F
    } else if (MODE == 11) {
#define F(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(s));
      F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
    } else if (MODE == 12) {
#define F(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(s));
      F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
    } else if (MODE == 13) {
#define F(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
      F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
    } else if (MODE == 14) {
#define F(x) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(x));
      F(d0) F(d1) F(d2) F(d3) F(d4) F(d5) F(d6) F(d7)
#undef F
    } else if (MODE == 7) {
#define F(x) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x) : "v"((double)s));
      F(d0) F(d0) F(d0) F(d0) F(d0) F(d0) F(d0) F(d0)
#undef F
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0.x + b1.y + b2.x + b3.y + b4.x + b5.x + b6.x + b7.x +
                                               (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (float)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);
}
template <int MODE> void run(const char* name, int waves_per_simd) {
  float* out; hipMalloc(&out, 1 << 24);
  const int iters = 20000;
  const int blocks = 256 * 4 * waves_per_simd;   // one wave per block
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 64>>>(out, 100, 1.0001f);
  hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks, 64>>>(out, iters, 1.0001f); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // cycles per instruction per SIMD: time * clock / (iters * 8 * waves_per_simd)
  printf("%-22s waves/SIMD %d: %.3f ms  -> %.2f cycles per instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms,
         ms * 1e-3 * 2.4e9 / ((double)iters * 8 * waves_per_simd));
  hipFree(out);
}
int main() {
  for (int w : {1, 4, 8}) {
    run<0>("v_fma_f32 x8 indep", w); run<1>("v_pk_fma_f32 x8 indep", w); run<2>("v_fma_f64 x8 indep", w);
    run<3>("v_pk_mul_f32 x8 indep", w); run<4>("v_pk_add_f32 x8 indep", w);
    run<12>("v_mul_f32 x8 indep", w); run<8>("v_add_u32 x8 indep", w); run<9>("v_mov_b32 x8 indep", w); run<11>("v_cndmask_b32 x8 indep", w);
    run<10>("v_add_f64 x8 indep", w); run<14>("v_lshlrev_b64 x8 indep", w); run<13>("v_rcp_f32 x8 indep", w);
    run<5>("v_fma_f32 dependent", w); run<6>("v_pk_fma_f32 dependent", w); run<7>("v_fma_f64 dependent", w);
  }
  return 0;
}
