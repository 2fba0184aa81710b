// accuracy of the two 1/sqrt(x) refinements used on the Cholesky pivot chain (kernels_chol.hip), in ulps of the result,
// against the correctly rounded value computed on the host in long double.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/ubench_rsqrt.hip -o tools/probes/ubench_rsqrt
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ double rs_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double e = __builtin_fma(-x * y, y, 1.0);
  y = __builtin_fma(y * 0.5, e, y);
  e = __builtin_fma(-x * y, y, 1.0);
  y = __builtin_fma(y * 0.5, e, y);
  return y;
}
__device__ double rs_h(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double t = x * y;
  const double e = __builtin_fma(-t, y, 1.0);
  double p = __builtin_fma(0.375, e, 0.5);
  p = p * e;
  return __builtin_fma(y, p, y);
}
__device__ double rs_h2(double x) {  // Halley with the residual formed from the squared estimate (one rounding less inside e)
  const double y = __builtin_amdgcn_rsq(x);
  const double y2 = y * y;
  const double e = __builtin_fma(-x, y2, 1.0);
  double p = __builtin_fma(0.375, e, 0.5);
  p = p * e;
  return __builtin_fma(y, p, y);
}
__global__ void k(const double* x, double* a, double* b, double* c, double* s, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { a[i] = rs_nr(x[i]); b[i] = rs_h(x[i]); c[i] = rs_h2(x[i]); s[i] = __builtin_amdgcn_rsq(x[i]); }
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), a(n), b(n), c(n), s(n);
  srand(3);
  for (int i = 0; i < n; ++i) x[i] = std::exp((rand() / (double)RAND_MAX) * 40.0 - 30.0);
  double *dx, *da, *db, *dc, *ds;
  hipMalloc(&dx, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dc, n * 8); hipMalloc(&ds, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), 256, 0, 0, dx, da, db, dc, ds, n);
  hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost); hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost);
  double ma = 0, mb = 0, mc = 0, ms = 0, sa = 0, sb = 0, sc = 0;
  for (int i = 0; i < n; ++i) {
    const long double r = 1.0L / sqrtl((long double)x[i]);
    const double ulp = std::ldexp(1.0, std::ilogb((double)r) - 52);
    const double ea = std::fabs((double)((long double)a[i] - r)) / ulp, eb = std::fabs((double)((long double)b[i] - r)) / ulp;
    const double ec = std::fabs((double)((long double)c[i] - r)) / ulp, es = std::fabs((double)((long double)s[i] - r)) / (double)r;
    ma = std::fmax(ma, ea); mb = std::fmax(mb, eb); mc = std::fmax(mc, ec); ms = std::fmax(ms, es);
    sa += ea; sb += eb; sc += ec;
  }
  printf("v_rsq_f64 seed: max relative error %.3g (2^%.1f)\n", ms, std::log2(ms));
  printf("two Newton steps : max %.3f ulp, mean %.3f ulp\n", ma, sa / n);
  printf("one Halley step  : max %.3f ulp, mean %.3f ulp\n", mb, sb / n);
  printf("Halley (e from y^2): max %.3f ulp, mean %.3f ulp\n", mc, sc / n);
  return 0;
}
