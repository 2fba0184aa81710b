// ubench_mfma16.hip -- issue rate of v_mfma_f64_16x16x4_f64 on gfx950 when it accumulates IN PLACE (inline asm, vDst ==
// SrcC, accumulators in AGPRs) -- the first probe (tools/probes/ubench_f64.hip, builtin, 137-150 cycles) may have measured the
// compiler's accumulator copies rather than the instruction.  Variants: N independent accumulators (N = 4, 8, 16), same
// or distinct A/B registers, 1 or 2 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/ubench_mfma16.hip -o tools/probes/ubench_mfma16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma16(double a, double b, d4& c) {
  asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16v(double a, double b, d4& c) {  // accumulator in architectural VGPRs
  asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma4(double a, double b, double& c) {
  asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <int NACC, bool DISTINCT>
__global__ __launch_bounds__(256) void k16(double* out, int iters, long long* clk) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = 1.0 + threadIdx.x * 1e-3 + i; b[i] = 2.0 - threadIdx.x * 1e-3 - i; }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) mfma16(DISTINCT ? a[i & 3] : a[0], DISTINCT ? b[(i >> 2) & 3] : b[0], acc[i]);
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16v(double* out, int iters, long long* clk) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 2.0 - threadIdx.x * 1e-3;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) mfma16v(a, b, acc[i]);
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

__global__ __launch_bounds__(256) void k4(double* out, int iters, long long* clk) {
  double acc[64];
  for (int i = 0; i < 64; ++i) acc[i] = 0;
  double a = 1.0 + threadIdx.x * 1e-3, b = 2.0 - threadIdx.x * 1e-3;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 64; ++i) mfma4(a, b, acc[i]);
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < 64; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

template <typename F>
void run(const char* name, F launch, int per_iter, double flops_per, int iters, int blocks_per_cu) {
  double* out; long long* clk;
  hipMalloc((void**)&out, 256 * 256 * 2 * 8 * 2);
  hipMalloc((void**)&clk, 8);
  launch(out, 10, clk, blocks_per_cu);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  launch(out, iters, clk, blocks_per_cu);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  const double n_mfma = (double)iters * per_iter;
  const double tf = n_mfma * flops_per * 256.0 * 4.0 * blocks_per_cu / (ms * 1e-3) / 1e12;
  printf("%-46s blocks/CU %d: %8.3f ms  %6.1f cycles/MFMA (counter %lld / %.0f)  %6.2f TF/s\n", name, blocks_per_cu, ms,
         (double)c / n_mfma, c, n_mfma, tf);
}

int main() {
  const int iters = 20000;
  for (int bpc = 1; bpc <= 4; ++bpc) {
    run("16x16x4 in place, 16 accumulators, same A/B", [](double* o, int it, long long* c, int b) { hipLaunchKernelGGL((k16<16, false>), dim3(256 * b), 256, 0, 0, o, it, c); }, 16, 2048.0, iters, bpc);
    run("16x16x4 in place, 8 accumulators", [](double* o, int it, long long* c, int b) { hipLaunchKernelGGL((k16<8, false>), dim3(256 * b), 256, 0, 0, o, it, c); }, 8, 2048.0, iters, bpc);
    run("16x16x4 in place, 4 accumulators", [](double* o, int it, long long* c, int b) { hipLaunchKernelGGL((k16<4, false>), dim3(256 * b), 256, 0, 0, o, it, c); }, 4, 2048.0, iters, bpc);
    run("16x16x4 in place, VGPR acc, 16 accumulators", [](double* o, int it, long long* c, int b) { hipLaunchKernelGGL((k16v<16>), dim3(256 * b), 256, 0, 0, o, it, c); }, 16, 2048.0, iters, bpc);
    run("16x16x4 in place, VGPR acc, 8 accumulators", [](double* o, int it, long long* c, int b) { hipLaunchKernelGGL((k16v<8>), dim3(256 * b), 256, 0, 0, o, it, c); }, 8, 2048.0, iters, bpc);
    run("16x16x4 in place, 2 accumulators", [](double* o, int it, long long* c, int b) { hipLaunchKernelGGL((k16<2, false>), dim3(256 * b), 256, 0, 0, o, it, c); }, 2, 2048.0, iters, bpc);
    run("4x4x4_4b in place, 64 accumulators", [](double* o, int it, long long* c, int b) { hipLaunchKernelGGL(k4, dim3(256 * b), 256, 0, 0, o, it, c); }, 64, 512.0, iters, bpc);
  }
  return 0;
}
