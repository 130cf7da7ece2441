// rsq_check.hip -- relative error of v_rsq_f64 (the seed) and of one / two Newton steps on it, 4 M arguments over 2^-60..2^60
// build: hipcc --offload-arch=gfx950 -O3 tools/rsq_check.hip -o tools/_build/rsq_check
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
__global__ void k(const double* x, double* e0, double* e1, double* e2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r = __builtin_amdgcn_rsq(v);
  const double exact = 1.0 / sqrt(v);   // (correctly rounded division and root: ~1 ulp)
  e0[i] = fabs(r - exact) / exact;
  r = r * fma(fma(-v * r, r, 1.0), 0.5, 1.0);
  e1[i] = fabs(r - exact) / exact;
  r = r * fma(fma(-v * r, r, 1.0), 0.5, 1.0);
  e2[i] = fabs(r - exact) / exact;
}
int main() {
  const int n = 1 << 22;
  double *x, *e0, *e1, *e2;
  hipMallocManaged(&x, n * 8); hipMallocManaged(&e0, n * 8); hipMallocManaged(&e1, n * 8); hipMallocManaged(&e2, n * 8);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    x[i] = ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 121) - 60);
  }
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, e0, e1, e2, n);
  hipDeviceSynchronize();
  double m0 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < n; ++i) { m0 = fmax(m0, e0[i]); m1 = fmax(m1, e1[i]); m2 = fmax(m2, e2[i]); }
  printf("v_rsq_f64 max relative error: seed %.3e (2^%.1f), one Newton step %.3e, two %.3e\n", m0, log2(m0), m1, m2);
  return 0;
}
