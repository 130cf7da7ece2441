// accuracy of candidate fast square roots against the correctly rounded sqrt, on the GPU
// build+run: hipcc --offload-arch=gfx950 -O3 tools/sqrt_check.hip -o /tmp/sqrt_check && /tmp/sqrt_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ double sqrt_a(double x) {  // 1 Goldschmidt iteration + 1 correction
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  double d = fma(-g, g, x);
  g = fma(d, h, g);
  return x > 0.0 ? g : 0.0;
}
__device__ __forceinline__ double sqrt_b(double x) {  // + second correction (LLVM's own sequence without scaling)
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  double d = fma(-g, g, x);
  g = fma(d, h, g);
  d = fma(-g, g, x);
  g = fma(d, h, g);
  return x > 0.0 ? g : 0.0;
}
__global__ void k(const double* x, double* a, double* b, double* ref, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { a[i] = sqrt_a(x[i]); b[i] = sqrt_b(x[i]); ref[i] = sqrt(x[i]); }
}
int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), a(n), b(n), r(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    double m = 1.0 + (double)rand() / RAND_MAX + (double)rand() / RAND_MAX * 1e-9;
    int e = (i % 4 == 0) ? (rand() % 600 - 300) : (rand() % 40 - 30);
    x[i] = ldexp(m, e);
  }
  x[0] = 0.0; x[1] = 1e-310; x[2] = 4.0; x[3] = 1e300;
  double *dx, *da, *db, *dr;
  hipMalloc(&dx, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dr, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, da, db, dr, n);
  hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost);
  double ea = 0, eb = 0, er = 0; long na = 0, nb = 0;
  for (int i = 0; i < n; ++i) {
    double t = std::sqrt(x[i]);
    if (t == 0) { if (a[i] != 0 || b[i] != 0) printf("zero case wrong %g %g\n", a[i], b[i]); continue; }
    double u = std::fabs(std::nextafter(t, INFINITY) - t);
    ea = std::fmax(ea, std::fabs(a[i] - t) / u); eb = std::fmax(eb, std::fabs(b[i] - t) / u);
    er = std::fmax(er, std::fabs(r[i] - t) / u);
    na += a[i] != t; nb += b[i] != t;
  }
  printf("max ulp error: sqrt_a %.2f (%ld of %d differ)  sqrt_b %.2f (%ld differ)  device sqrt %.2f; special: sqrt_a(0)=%g sqrt_a(1e-310)=%g (ref %g) sqrt_a(1e300)=%g (ref %g)\n",
         ea, na, n, eb, nb, er, a[0], a[1], std::sqrt(1e-310), a[3], std::sqrt(1e300));
  return 0;
}
