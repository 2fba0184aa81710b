// What limits a realistic v_mfma_f64_4x4x4_4b_f64 stream?  One wave per SIMD (256 threads, 1 block/CU), 64 accumulators,
// 16 A registers x 4 B registers per k-step like k_contract.  Variants:
//   0  same A, B for every MFMA            1  distinct A (16) / B (4) registers, no memory
//   2  (1) + A re-read from LDS each k-step (16 ds_read_b64)     3  (2) + B from global (4 x 16 B per 2 k-steps)
// Prints cycles/MFMA (s_memtime) and the shader clock (s_memtime / s_memrealtime @100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double mf(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
template <int V>
__global__ __launch_bounds__(256, 2) void k(double* out, const double2* __restrict__ gB, int iters, long long* clk) {
  __shared__ __attribute__((aligned(16))) double lds[32 * 80];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 32 * 80; i += 256) lds[i] = 1.0 + i * 1e-6;
  __syncthreads();
  double acc[4][4][4];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int c = 0; c < 4; ++c) acc[a][b][c] = 0;
  double af[4][4]; double2 bq[4];
  for (int a = 0; a < 4; ++a) for (int t = 0; t < 4; ++t) af[a][t] = 1.0 + lane * 1e-3 + a + 0.1 * t;
  for (int n = 0; n < 4; ++n) bq[n] = make_double2(2.0 - lane * 1e-3 + n, 1.0 + n);
  const int lk = lane >> 4, lb = (lane >> 2) & 3, li = lane & 3;
  int aoff[4]; for (int t = 0; t < 4; ++t) aoff[t] = lk * 80 + 4 * ((lb + t) & 3) + li;
  long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  if (V == 5) {  // one ds_read_b128 per (row tile, rotation) feeds BOTH k-steps of a k-pair
    const double2* l2 = reinterpret_cast<const double2*>(lds);
    int ao[4]; for (int t = 0; t < 4; ++t) ao[t] = lk * 64 + 4 * ((lb + t) & 3) + li;
    for (int it = 0; it < iters; ++it) {
      for (int n = 0; n < 4; ++n) bq[n] = gB[(size_t)((it * 4 + n) & 1023) * 64 + lane];
      double2 a2[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t) a2[a][t] = l2[(it & 1) * 256 + ao[t] + 16 * a];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const double bv = h == 0 ? bq[n].x : bq[n].y;
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[a][n][t] = mf(h == 0 ? a2[a][t].x : a2[a][t].y, bv, acc[a][n][t]);
        }
    }
  } else
  if (V == 4) {
    double a2[2][4][4];
    {
      const double* trow = lds;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t) a2[0][a][t] = trow[aoff[t] + 16 * a];
    }
    for (int it = 0; it < iters; ++it) {
      for (int n = 0; n < 4; ++n) bq[n] = gB[(size_t)((it * 4 + n) & 1023) * 64 + lane];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const double* trow = lds + (4 * ((2 * it + h + 1) & 7)) * 80;  // NEXT k-step
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int t = 0; t < 4; ++t) a2[(h + 1) & 1][a][t] = trow[aoff[t] + 16 * a];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const double bv = h == 0 ? bq[n].x : bq[n].y;
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[a][n][t] = mf(a2[h & 1][a][t], bv, acc[a][n][t]);
        }
      }
    }
  } else
  for (int it = 0; it < iters; ++it) {
    if (V >= 3) { for (int n = 0; n < 4; ++n) bq[n] = gB[(size_t)((it * 4 + n) & 1023) * 64 + lane]; }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (V >= 2) {
        const double* trow = lds + (4 * ((2 * it + h) & 7)) * 80;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int t = 0; t < 4; ++t) af[a][t] = trow[aoff[t] + 16 * a];
      }
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const double bv = h == 0 ? bq[n].x : bq[n].y;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[a][n][t] = (V == 0) ? mf(af[0][0], bq[0].x, acc[a][n][t]) : mf(af[a][t], bv, acc[a][n][t]);
      }
    }
  }
  long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  double s = 0;
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int c = 0; c < 4; ++c) s += acc[a][b][c];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
int main() {
  double* out; double2* gB; long long* clk; long long h[2];
  hipMalloc(&out, 512 * 256 * 8); hipMalloc(&gB, 1024 * 64 * 16); hipMemset(gB, 0, 1024 * 64 * 16); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int grid = 256; grid <= 512; grid += 256)
  for (int v = 0; v < 6; ++v) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL(k<0>, grid, 256, 0, 0, out, gB, iters, clk);
      if (v == 1) hipLaunchKernelGGL(k<1>, grid, 256, 0, 0, out, gB, iters, clk);
      if (v == 2) hipLaunchKernelGGL(k<2>, grid, 256, 0, 0, out, gB, iters, clk);
      if (v == 3) hipLaunchKernelGGL(k<3>, grid, 256, 0, 0, out, gB, iters, clk);
      if (v == 4) hipLaunchKernelGGL(k<4>, grid, 256, 0, 0, out, gB, iters, clk);
      if (v == 5) hipLaunchKernelGGL(k<5>, grid, 256, 0, 0, out, gB, iters, clk);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double nm = (double)iters * 128;
    printf("grid %d variant %d: %.3f ms  %.2f TF/s   s_memtime ticks/MFMA %.2f   wall ns/MFMA %.3f  memtime/realtime ratio %.3f (x100MHz)\n", grid, v, ms,
           nm * 512 * 4 * grid / ms * 1e-9, (double)h[0] / nm, ms * 1e6 / nm, (double)h[0] / (double)h[1]);
  }
}
