// r05: cost (FP64-FMA equivalents per call, as the producer k_corr_chunk issues it: 8 independent values per thread) and accuracy
// (ulps against long double on the host) of the radial-profile building blocks -- library sqrt / exp against domain-restricted forms:
//   sqrt:  v_rsq_f64 + one Halley step, root = x * inv          [+ one residual correction]
//   exp(-K), K >= 0:  n = rint(-K log2 e), two-constant Cody-Waite, degree-11 polynomial, v_ldexp_f64        (no special cases)
//                     the same with a 64-entry table of 2^(j/64) (hi, lo) in LDS and a degree-5 polynomial
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/ubench_profile.hip -o tools/probes/ubench_profile
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ double sqrt_h(double x) {  // x >= 0; 0 -> ~1e-150
  const double p = fmax(x, 1e-300);
  const double y = __builtin_amdgcn_rsq(p);
  const double t = p * y;
  const double e = __builtin_fma(-t, y, 1.0);
  double q = __builtin_fma(0.375, e, 0.5);
  q = q * e;
  const double inv = __builtin_fma(y, q, y);
  return p * inv;
}
__device__ __forceinline__ double sqrt_hc(double x) {  // + residual correction: correctly rounded but for rare ties
  const double p = fmax(x, 1e-300);
  const double y = __builtin_amdgcn_rsq(p);
  const double t = p * y;
  const double e = __builtin_fma(-t, y, 1.0);
  double q = __builtin_fma(0.375, e, 0.5);
  q = q * e;
  const double inv = __builtin_fma(y, q, y);
  const double g = p * inv;
  const double d = __builtin_fma(-g, g, p);
  return __builtin_fma(d, 0.5 * inv, g);
}
// exp(-K) for K >= 0 (any K: the result underflows through v_ldexp_f64 like the library's)
__device__ __forceinline__ double expn_poly(double K) {
  const double t = K * -1.4426950408889634;
  const double n = __builtin_rint(t);
  double r = __builtin_fma(n, -0.693147180559663, -K);  // 42-bit ln2_hi: n * hi exact for |n| < 2^11
  r = __builtin_fma(n, -2.8235290563031577e-13, r);
  double q = 0x1.af389ecfc4b9cp-26;
  q = __builtin_fma(q, r, 0x1.28917c89a43a7p-22);
  q = __builtin_fma(q, r, 0x1.71de0db2f6b19p-19);
  q = __builtin_fma(q, r, 0x1.a019b9149a41cp-16);
  q = __builtin_fma(q, r, 0x1.a01a01a7c2efep-13);
  q = __builtin_fma(q, r, 0x1.6c16c17889ef1p-10);
  q = __builtin_fma(q, r, 0x1.11111111109b5p-7);
  q = __builtin_fma(q, r, 0x1.5555555553d68p-5);
  q = __builtin_fma(q, r, 0x1.5555555555556p-3);
  q = __builtin_fma(q, r, 0x1.0000000000001p-1);
  q = __builtin_fma(q, r, 1.0);
  q = __builtin_fma(q, r, 1.0);
  return __builtin_amdgcn_ldexp(q, (int)n);
}
__device__ __forceinline__ double expn_tab(double K, const double2* __restrict__ tab) {
  const double t = K * -92.33248261689366;  // -64 / ln 2
  const double n = __builtin_rint(t);
  double r = __builtin_fma(n, -0.010830424696244734, -K);
  r = __builtin_fma(n, -4.411764150473684e-15, r);
  const int ni = (int)n;
  const double2 T = tab[ni & 63];
  double q = 0.0083333391516934971814;
  q = __builtin_fma(q, r, 0.041666707395191960633);
  q = __builtin_fma(q, r, 0.16666666666664533908);
  q = __builtin_fma(q, r, 0.49999999999985070691);
  const double p = __builtin_fma(r * r, q, r);
  const double v = __builtin_fma(T.x, p, T.y) + T.x;
  return __builtin_amdgcn_ldexp(v, ni >> 6);
}

template <int V>
__device__ __forceinline__ double f(double x, const double2* tab) {
  if (V == 0) return sqrt(x);
  if (V == 1) return sqrt_h(x);
  if (V == 2) return sqrt_hc(x);
  if (V == 3) return exp(-x);
  if (V == 4) return expn_poly(x);
  if (V == 5) return expn_tab(x, tab);
  if (V == 6) {  // Matern-5/2, library
    const double K = sqrt(x) * 2.23606797749979;
    return (1.0 + K + (K * K) * 0.3333333333333333) * exp(-K);
  }
  if (V == 7) {
    const double K = sqrt_h(x) * 2.23606797749979;
    return (1.0 + K + (K * K) * 0.3333333333333333) * expn_poly(K);
  }
  if (V == 8) {
    const double K = sqrt_h(x) * 2.23606797749979;
    return (1.0 + K + (K * K) * 0.3333333333333333) * expn_tab(K, tab);
  }
  if (V == 9) return __builtin_fma(x, 0.999, 0.001);  // one FMA: the unit
  return x;
}

template <int V>
__global__ __launch_bounds__(256) void k_time(const double* in, double* out, int iters) {
  __shared__ double2 tab[64];
  if (threadIdx.x < 64) tab[threadIdx.x] = make_double2(exp2((double)threadIdx.x / 64.0), 0.0);
  __syncthreads();
  double a[8];
  const int gid = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = in[(gid * 8 + i) & 4095];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(f<V>(a[i], tab), 0.37, 3.1 + 0.01 * i);  // values stay in [3.1, 5.5]
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[gid] = s;
}

template <int V>
__global__ void k_acc(const double* in, double* out, int n, const double2* gtab) {
  __shared__ double2 tab[64];
  if (threadIdx.x < 64) tab[threadIdx.x] = gtab[threadIdx.x];
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = f<V>(in[i], tab);
}

template <int V>
double time_variant(const double* din, double* dout, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 256 * 8;
  hipLaunchKernelGGL(k_time<V>, dim3(blocks), 256, 0, 0, din, dout, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_time<V>, dim3(blocks), 256, 0, 0, din, dout, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // wave-calls per SIMD: blocks * 4 waves * 8 values * iters / (256 CUs * 4 SIMDs)
  const double calls = (double)blocks * 4 * 8 * iters / 1024.0;
  return ms * 1e-3 * 2.4e9 / calls;  // cycles per wave-call per SIMD at the nominal 2.4 GHz
}

int main() {
  const int n = 1 << 23;
  std::vector<double> x(n), y(n);
  srand(5);
  for (int i = 0; i < n; ++i) x[i] = std::exp((rand() / (double)RAND_MAX) * 16.0 - 9.0) * (i % 7 == 0 ? 50.0 : 1.0);  // 1e-4 .. 1100 * 50
  for (int i = 0; i < n; ++i) if (x[i] > 745.0) x[i] = 745.0 * (rand() / (double)RAND_MAX);
  double *din, *dout;
  double2* dtab;
  CK(hipMalloc(&din, n * 8)); CK(hipMalloc(&dout, n * 8)); CK(hipMalloc(&dtab, 64 * 16));
  CK(hipMemcpy(din, x.data(), n * 8, hipMemcpyHostToDevice));
  std::vector<double> tab(128);
  for (int j = 0; j < 64; ++j) {
    const long double v = exp2l((long double)j / 64.0L);
    tab[2 * j] = (double)v;
    tab[2 * j + 1] = (double)(v - (long double)tab[2 * j]);
  }
  CK(hipMemcpy(dtab, tab.data(), 64 * 16, hipMemcpyHostToDevice));
  const int iters = 2000;
  const double unit = time_variant<9>(din, dout, iters);
  const char* names[9] = {"sqrt (library)", "sqrt rsq+Halley", "sqrt rsq+Halley+corr", "exp(-x) (library)", "exp(-x) CW + deg-11", "exp(-x) table64 + deg-5",
                          "Matern-5/2 library", "Matern-5/2 Halley + deg-11", "Matern-5/2 Halley + table"};
  double cyc[9];
  cyc[0] = time_variant<0>(din, dout, iters); cyc[1] = time_variant<1>(din, dout, iters); cyc[2] = time_variant<2>(din, dout, iters);
  cyc[3] = time_variant<3>(din, dout, iters); cyc[4] = time_variant<4>(din, dout, iters); cyc[5] = time_variant<5>(din, dout, iters);
  cyc[6] = time_variant<6>(din, dout, iters); cyc[7] = time_variant<7>(din, dout, iters); cyc[8] = time_variant<8>(din, dout, iters);
  printf("unit: one v_fma_f64 per value = %.2f cycles per wave-call per SIMD\n", unit);
  // accuracy
  for (int v = 0; v < 9; ++v) {
    switch (v) {
      case 0: hipLaunchKernelGGL(k_acc<0>, dim3(n / 256), 256, 0, 0, din, dout, n, dtab); break;
      case 1: hipLaunchKernelGGL(k_acc<1>, dim3(n / 256), 256, 0, 0, din, dout, n, dtab); break;
      case 2: hipLaunchKernelGGL(k_acc<2>, dim3(n / 256), 256, 0, 0, din, dout, n, dtab); break;
      case 3: hipLaunchKernelGGL(k_acc<3>, dim3(n / 256), 256, 0, 0, din, dout, n, dtab); break;
      case 4: hipLaunchKernelGGL(k_acc<4>, dim3(n / 256), 256, 0, 0, din, dout, n, dtab); break;
      case 5: hipLaunchKernelGGL(k_acc<5>, dim3(n / 256), 256, 0, 0, din, dout, n, dtab); break;
      case 6: hipLaunchKernelGGL(k_acc<6>, dim3(n / 256), 256, 0, 0, din, dout, n, dtab); break;
      case 7: hipLaunchKernelGGL(k_acc<7>, dim3(n / 256), 256, 0, 0, din, dout, n, dtab); break;
      case 8: hipLaunchKernelGGL(k_acc<8>, dim3(n / 256), 256, 0, 0, din, dout, n, dtab); break;
    }
    CK(hipMemcpy(y.data(), dout, n * 8, hipMemcpyDeviceToHost));
    double worst = 0, sum = 0;
    long cnt = 0, differ = 0;
    for (int i = 0; i < n; ++i) {
      long double t;
      if (v < 3) t = sqrtl((long double)x[i]);
      else if (v < 6) t = expl(-(long double)x[i]);
      else { const long double K = sqrtl((long double)x[i]) * 2.23606797749979L; t = (1.0L + K + K * K / 3.0L) * expl(-K); }
      if (t < 1e-300L) continue;  // subnormal results: not the question
      const double ulp = std::ldexp(1.0, std::ilogb((double)t) - 52);
      const double e = std::fabs((double)((long double)y[i] - t)) / ulp;
      worst = std::fmax(worst, e); sum += e; ++cnt;
      if (y[i] != (double)t) ++differ;
    }
    printf("%-30s %7.2f cycles = %5.1f FMA-equivalents | max %.3f ulp, mean %.3f ulp, %.3f %% of %ld values differ from the correctly rounded one\n",
           names[v], cyc[v] - unit, (cyc[v] - unit) / unit, worst, sum / cnt, 100.0 * differ / cnt, cnt);
  }
  return 0;
}
