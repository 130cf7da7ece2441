// issue cost of selects, compares, lane reads and DPP moves on gfx950 (study; companion of mb_valu_rates.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE> __global__ void k(float* out, int iters, float s) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned long long m = 0x5555555555555555ull + iters;
  asm volatile("s_mov_b64 vcc, %0" :: "s"(m) : "vcc");
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#define F(x) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(s));
      F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
    } else if (MODE == 1) {
#define F(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(s), "s"(m));
      F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
    } else if (MODE == 2) {   // different destination than sources
#define F(x, y) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(x) : "v"(y), "v"(s));
      F(a0, a1) F(a2, a3) F(a4, a5) F(a6, a7) F(a1, a0) F(a3, a2) F(a5, a4) F(a7, a6)
#undef F
    } else if (MODE == 3) {
#define F(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(s) : "vcc");
      F(a0) F(a1) F(a2) F(a3)
#undef F
    } else if (MODE == 4) {
#define F(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(s));
      F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
    } else if (MODE == 5) {
#define F(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x), "v"(s) : "vcc");
      F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
    } else if (MODE == 6) {
#define F(x) asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(x) : "s20");
      F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
    } else if (MODE == 7) {
#define F(x) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
      F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE> void run(const char* name, int waves_per_simd, int per_iter = 8) {
  float* out; hipMalloc(&out, 1 << 24);
  const int iters = 20000;
  const int blocks = 256 * 4 * waves_per_simd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 64>>>(out, 100, 1.0001f);
  hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks, 64>>>(out, iters, 1.0001f); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s waves/SIMD %d: %.3f ms -> %.2f cycles per instruction per SIMD\n", name, waves_per_simd, ms,
         ms * 1e-3 * 2.4e9 / ((double)iters * per_iter * waves_per_simd));
  hipFree(out);
}
int main() {
  for (int w : {1, 4}) {
    run<0>("v_cndmask_b32_e32 vcc (x = f(x))", w); run<1>("v_cndmask_b32_e64 sgpr mask", w); run<2>("v_cndmask_b32_e32 dst != src", w);
    run<3>("v_cmp + v_cndmask pairs", w, 8); run<4>("v_max_f32", w); run<5>("v_cmp_lt_f32 -> vcc", w); run<6>("v_readlane_b32", w); run<7>("v_mov_b32_dpp row_shr", w);
  }
}
