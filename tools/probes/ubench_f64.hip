// Micro-benchmarks that fix the design constants of the posterior kernel on gfx950:
//   (1) v_mfma_f64_16x16x4_f64 issue rate (cycles / instruction / SIMD) with 16 independent accumulators
//   (2) v_fma_f64 VALU rate
//   (3) whether f64 VALU work issued between MFMAs hides in the MFMA shadow (same wave) or beside it (2 waves/SIMD)
//   (4) software exp() / sqrt() f64 cost
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/ubench_f64.hip -o tools/probes/ubench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NVALU>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a0, double b0) {
  d4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NVALU; ++j) v[(i * NVALU + j) & 7] = __builtin_fma(v[(i * NVALU + j) & 7], 1.0000001, 1e-9);
    }
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_valu(double* out, int iters, double a0) {
  double v[16];
  for (int i = 0; i < 16; ++i) v[i] = a0 + i + threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], 1.0000001, 1e-9);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
__global__ __launch_bounds__(256) void k_fn(double* out, int iters, double a0) {
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = a0 + 0.01 * i + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) v[i] = exp(-v[i]) + 0.5;
      if (OP == 1) v[i] = sqrt(v[i]) + 0.5;
      if (OP == 2) v[i] = erfc(v[i]) + 0.5;
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float timeit(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  double* out;
  hipMalloc(&out, 256 * 2048 * 8 * sizeof(double));
  const int iters = 2000;
  for (int wpb = 1; wpb <= 2; ++wpb) {  // blocks per CU (256 threads = 1 wave per SIMD each)
    int grid = 256 * wpb;
    auto rep = [&](const char* name, float ms, int nv) {
      double mfma = (double)grid * 4 * iters * 16;  // wave-level MFMA instructions
      double flops = mfma * 2048.0;
      printf("%-28s blocks/CU %d : %8.3f ms  %7.2f TF/s f64-mfma  (%.1f cyc/MFMA/SIMD @2.4GHz, %d valu/mfma)\n", name, wpb, ms,
             flops / ms * 1e-9, ms * 1e-3 * 2.4e9 / (iters * 16.0 * wpb), nv);
    };
    rep("mfma only", timeit([&] { hipLaunchKernelGGL(k_mfma<0>, grid, 256, 0, 0, out, iters, 1.0, 2.0); }), 0);
    rep("mfma + 2 fma/mfma", timeit([&] { hipLaunchKernelGGL(k_mfma<2>, grid, 256, 0, 0, out, iters, 1.0, 2.0); }), 2);
    rep("mfma + 4 fma/mfma", timeit([&] { hipLaunchKernelGGL(k_mfma<4>, grid, 256, 0, 0, out, iters, 1.0, 2.0); }), 4);
    rep("mfma + 8 fma/mfma", timeit([&] { hipLaunchKernelGGL(k_mfma<8>, grid, 256, 0, 0, out, iters, 1.0, 2.0); }), 8);
    rep("mfma + 16 fma/mfma", timeit([&] { hipLaunchKernelGGL(k_mfma<16>, grid, 256, 0, 0, out, iters, 1.0, 2.0); }), 16);
    {
      float ms = timeit([&] { hipLaunchKernelGGL(k_valu, grid, 256, 0, 0, out, iters, 1.0); });
      double n = (double)grid * 4 * iters * 64;
      printf("%-28s blocks/CU %d : %8.3f ms  %7.2f TF/s f64-valu  (%.2f cyc/v_fma_f64/SIMD)\n", "valu fma only", wpb, ms, n * 128 / ms * 1e-9,
             ms * 1e-3 * 2.4e9 / (iters * 64.0 * wpb));
    }
    const char* nm[3] = {"exp", "sqrt", "erfc"};
    float t[3];
    t[0] = timeit([&] { hipLaunchKernelGGL(k_fn<0>, grid, 256, 0, 0, out, iters, 1.0); });
    t[1] = timeit([&] { hipLaunchKernelGGL(k_fn<1>, grid, 256, 0, 0, out, iters, 1.0); });
    t[2] = timeit([&] { hipLaunchKernelGGL(k_fn<2>, grid, 256, 0, 0, out, iters, 1.0); });
    for (int i = 0; i < 3; ++i)
      printf("%-28s blocks/CU %d : %8.3f ms  (%.1f cyc per wave-call/SIMD = %.1f fma-equivalents)\n", nm[i], wpb, t[i],
             t[i] * 1e-3 * 2.4e9 / (iters * 8.0 * wpb), t[i] * 1e-3 * 2.4e9 / (iters * 8.0 * wpb) / 4.0);
  }
  return 0;
}
