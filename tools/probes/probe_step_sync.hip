// What does ONE step of a persistent multi-workgroup elimination cost in hand-over latency?  G resident workgroups; per step every workgroup
// waits for the step's counter (nb + 1 publishers), acquires, reads three 32-KB tiles, "computes" (a timed busy loop), and the step's nb + 1
// publishers store a 32-KB tile, release and bump the counter.  Build: hipcc --offload-arch=gfx950 -O3 tools/probes/probe_step_sync.hip -o /tmp/probe_step_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void busy(long cycles) {
  const long t0 = clock64();
  while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(1);
}
__global__ __launch_bounds__(256) void k_steps(double* __restrict__ tiles, unsigned* __restrict__ counters, int nb, int steps, long work_cycles,
                                               long diag_cycles, double* __restrict__ sink, int* __restrict__ fail) {
  const int wg = blockIdx.x, tid = threadIdx.x;
  double acc = 0.0;
  for (int s = 0; s < steps; ++s) {
    if (s > 0) {
      if (tid == 0) {
        int spin = 0;
        while (__hip_atomic_load(counters + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(nb + 1)) {
          __builtin_amdgcn_s_sleep(1);
          if (++spin > (1 << 22)) { *fail = s; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
    }
    // three 32-KB tiles of the step (W, raw rows): tile ids (s, 0), (s, wg % (nb + 1)), (s, (wg / 7) % (nb + 1))
    const double* t0 = tiles + ((size_t)s * (nb + 1) + 0) * 4096;
    const double* t1 = tiles + ((size_t)s * (nb + 1) + wg % (nb + 1)) * 4096;
    const double* t2 = tiles + ((size_t)s * (nb + 1) + (wg / 7) % (nb + 1)) * 4096;
    for (int e = tid; e < 4096; e += 256) acc += t0[e] + t1[e] + t2[e];
    busy(work_cycles);
    if (wg <= nb && s + 1 < steps) {  // a publisher of step s + 1
      if (wg == 0) busy(diag_cycles);  // the diagonal block's factorisation
      double* o = tiles + ((size_t)(s + 1) * (nb + 1) + wg) * 4096;
      for (int e = tid; e < 4096; e += 256) o[e] = acc * 1e-300 + (double)e;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counters + s + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (tid == 0) sink[wg] = acc;
}

int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 32;
  const int G = (nb + 1) * (nb + 2) / 2 - 1, steps = nb;
  double *tiles, *sink; unsigned* counters; int* fail;
  CHECK(hipMalloc(&tiles, (size_t)(steps + 1) * (nb + 1) * 4096 * 8));
  CHECK(hipMemset(tiles, 0, (size_t)(steps + 1) * (nb + 1) * 4096 * 8));
  CHECK(hipMalloc(&sink, G * 8)); CHECK(hipMalloc(&counters, (steps + 2) * 4)); CHECK(hipMalloc(&fail, 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  int dev = 0, coop = 0, nsm = 0, perSm = 0;
  CHECK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev));
  CHECK(hipDeviceGetAttribute(&nsm, hipDeviceAttributeMultiprocessorCount, dev));
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perSm, k_steps, 256, 0));
  printf("nb = %d: %d workgroups, cooperative launch %d, %d CUs x %d resident workgroups\n", nb, G, coop, nsm, perSm);
  for (long diag : {0L, 33600L}) for (long work : {0L, 8000L}) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CHECK(hipMemset(counters, 0, (steps + 2) * 4)); CHECK(hipMemset(fail, 0, 4));
      int a_nb = nb, a_steps = steps; long a_work = work, a_diag = diag;
      void* args[] = {&tiles, &counters, &a_nb, &a_steps, &a_work, &a_diag, &sink, &fail};
      CHECK(hipEventRecord(e0));
      CHECK(hipLaunchCooperativeKernel((void*)k_steps, dim3(G), dim3(256), args, 0, 0));
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
      int f; CHECK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
      if (f) { printf("  wait expired at step %d\n", f); return 2; }
    }
    printf("  diag %6ld cycles, work %5ld cycles: %8.1f us for %d steps = %6.2f us a step (busy part %.2f us)\n", diag, work, 1e3 * best, steps,
           1e3 * best / steps, (diag + work) / 100.0 / 1.0e0 * 0.01 * 100.0 / 100.0);
  }
  return 0;
}
