// second micro-benchmark: v_mfma_f64_4x4x4_4b_f64 issue rate, and real shader clock via s_memtime/wall ratio
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_mfma4(double* out, int iters, double a0, double b0, long long* clk) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = 0;
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma16(double* out, int iters, double a0, double b0, long long* clk) {
  d4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}
__global__ __launch_bounds__(256) void k_valu(double* out, int iters, double a0, long long* clk) {
  double v[16];
  for (int i = 0; i < 16; ++i) v[i] = a0 + i + threadIdx.x * 1e-3;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], 1.0000001, 1e-9);
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}
int main() {
  double* out; long long* clk; long long h;
  hipMalloc(&out, 256 * 2048 * 8 * sizeof(double)); hipMalloc(&clk, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int wpb = 1; wpb <= 2; ++wpb) for (int which = 0; which < 3; ++which) {
    int grid = 256 * wpb; float ms;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k_mfma4, grid, 256, 0, 0, out, iters, 1.0, 2.0, clk);
      if (which == 1) hipLaunchKernelGGL(k_mfma16, grid, 256, 0, 0, out, iters, 1.0, 2.0, clk);
      if (which == 2) hipLaunchKernelGGL(k_valu, grid, 256, 0, 0, out, iters, 1.0, clk);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    double ninstr = (double)iters * 16;
    double flops_per = which == 0 ? 512.0 : which == 1 ? 2048.0 : 128.0;
    const char* nm[3] = {"mfma_f64_4x4x4_4b", "mfma_f64_16x16x4", "v_fma_f64"};
    printf("%-20s blocks/CU %d: %.3f ms  %.2f TF/s  counter ticks/instr(one wave) %.1f  (counter %.1f MHz-equivalent)\n", nm[which], wpb, ms,
           ninstr * flops_per * grid * 4 / ms * 1e-9, (double)h / ninstr, (double)h / (ms * 1e3));
  }
  return 0;
}
