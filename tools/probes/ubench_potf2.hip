// ubench_potf2.hip -- latency of a 64x64 in-register Cholesky (one wave, lane = row) on gfx950: which pivot
// arithmetic and which broadcast mechanism make the serial chain of kernels_chol.hip shortest?
//   V0  sqrt + IEEE divide, v_readlane broadcast                  (first version of k_chol_update's diagonal role)
//   V1  v_rsq_f64 + 2 Newton steps, v_readlane broadcast
//   V2  V1 pivots, broadcast through LDS (one ds_write_b64, uniform-address ds_read_b128)
//   V3  V1 + the inverse of the factor formed in the same sweep (rows of L^-1 beside rows of L)
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/ubench_potf2.hip -o tools/probes/ubench_potf2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

constexpr int CB = 64;

__device__ __forceinline__ double lane_bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(x) to ~1 ulp: hardware estimate (2^-27..) + two Newton-Raphson steps in FMA form
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double e = __builtin_fma(-x * y, y, 1.0);
  y = __builtin_fma(y * 0.5, e, y);
  e = __builtin_fma(-x * y, y, 1.0);
  y = __builtin_fma(y * 0.5, e, y);
  return y;
}

template <int V>
__device__ __forceinline__ void potf2(double (&a)[CB], double (&w)[CB], int lane, double* lds) {
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const double piv = lane_bcast(a[j], j);
    double s, inv;
    if (V == 0) {
      s = sqrt(piv);
      inv = 1.0 / s;
    } else {
      inv = rsqrt_nr(piv);
      s = piv * inv;
      s = __builtin_fma(__builtin_fma(-s, s, piv), 0.5 * inv, s);  // one correction step: s ~ sqrt(piv) to 1 ulp
    }
    const double l = lane == j ? s : a[j] * inv;
    a[j] = l;
    if (V == 2) {
      lds[lane] = l;
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
      for (int c = j + 1; c < CB; ++c) a[c] = __builtin_fma(-l, ((volatile double*)lds)[c], a[c]);
      __builtin_amdgcn_wave_barrier();
    } else if (V == 4) {  // ds_bpermute (LDS crossbar, result stays in VGPRs)
#pragma unroll
      for (int c = j + 1; c < CB; ++c) a[c] = __builtin_fma(-l, __shfl(l, c), a[c]);
    } else if (V == 5) {  // readlane in groups of 8 columns, fenced so that the scheduler cannot hoist them all
#pragma unroll
      for (int c0 = j + 1; c0 < CB; c0 += 8) {
#pragma unroll
        for (int c = c0; c < c0 + 8 && c < CB; ++c) a[c] = __builtin_fma(-l, lane_bcast(l, c), a[c]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (V == 6) {  // LDS broadcast with plain (non-volatile) 16-byte reads between compiler fences
      lds[lane] = l;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const double2* l2 = reinterpret_cast<const double2*>(lds);
#pragma unroll
      for (int c2 = (j + 1) / 2; c2 < CB / 2; ++c2) {
        const double2 v = l2[c2];
        if (2 * c2 > j) a[2 * c2] = __builtin_fma(-l, v.x, a[2 * c2]);
        a[2 * c2 + 1] = __builtin_fma(-l, v.y, a[2 * c2 + 1]);
      }
      asm volatile("" ::: "memory");
    } else {
#pragma unroll
      for (int c = j + 1; c < CB; ++c) a[c] = __builtin_fma(-l, lane_bcast(l, c), a[c]);
    }
    if (V == 3) {
      // rows of W = L^-1: row j is final once scaled; lanes below it subtract L[i][j] * W[j][:]
      if (lane == j) w[j] = 1.0;
      if (lane == j) {
#pragma unroll
        for (int m = 0; m <= j; ++m) w[m] *= inv;
      }
#pragma unroll
      for (int m = 0; m <= j; ++m) {
        const double wj = lane_bcast(w[m], j);
        if (lane > j) w[m] = __builtin_fma(-l, wj, w[m]);
      }
    }
  }
}

template <int V>
__global__ __launch_bounds__(64) void k_bench(const double* __restrict__ A, double* __restrict__ L, double* __restrict__ W, int reps) {
  __shared__ __attribute__((aligned(16))) double lds[CB];
  const int lane = threadIdx.x;
  double a[CB], w[CB];
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      a[c] = A[(size_t)c * CB + lane];
      w[c] = 0.0;
    }
    potf2<V>(a, w, lane, lds);
#pragma unroll
    for (int c = 0; c < CB; ++c)
      if (lane >= c) L[(size_t)c * CB + lane] = a[c];
    if (V == 3) {
#pragma unroll
      for (int c = 0; c < CB; ++c)
        if (lane >= c) W[(size_t)c * CB + lane] = w[c];
    }
  }
}

template <int V>
void run(const char* name, const double* dA, double* dL, double* dW, const std::vector<double>& Lref, const std::vector<double>& A) {
  const int reps = 200;
  hipMemset(dL, 0, CB * CB * 8);
  hipMemset(dW, 0, CB * CB * 8);
  hipLaunchKernelGGL(k_bench<V>, dim3(1), 64, 0, 0, dA, dL, dW, 2);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_bench<V>, dim3(1), 64, 0, 0, dA, dL, dW, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<double> L(CB * CB), W(CB * CB);
  hipMemcpy(L.data(), dL, CB * CB * 8, hipMemcpyDeviceToHost);
  hipMemcpy(W.data(), dW, CB * CB * 8, hipMemcpyDeviceToHost);
  double err = 0, werr = 0;
  for (int c = 0; c < CB; ++c)
    for (int r = c; r < CB; ++r) err = fmax(err, fabs(L[c * CB + r] - Lref[c * CB + r]));
  if (V == 3) {  // || W L - I ||_max
    for (int i = 0; i < CB; ++i)
      for (int j = 0; j <= i; ++j) {
        double s = 0;
        for (int m = j; m <= i; ++m) s += W[m * CB + i] * Lref[j * CB + m];
        werr = fmax(werr, fabs(s - (i == j ? 1.0 : 0.0)));
      }
  }
  printf("%-44s %8.2f us / block   max|L - Lref| = %.2e   max|W L - I| = %.2e\n", name, ms * 1e3 / reps, err, werr);
}

int main() {
  // SPD test block: Gaussian kernel matrix of 64 random points + nugget (the kind of block the fit factors)
  std::vector<double> A(CB * CB), Lref(CB * CB, 0.0), x(CB * 3);
  unsigned s = 12345;
  for (auto& v : x) { s = s * 1664525u + 1013904223u; v = (s >> 8) / 16777216.0 * 4.0; }
  for (int i = 0; i < CB; ++i)
    for (int j = 0; j < CB; ++j) {
      double d2 = 0;
      for (int k = 0; k < 3; ++k) d2 += (x[i * 3 + k] - x[j * 3 + k]) * (x[i * 3 + k] - x[j * 3 + k]);
      A[j * CB + i] = exp(-0.5 * d2) + (i == j ? 1e-6 : 0.0);
    }
  std::vector<double> T = A;  // reference: column Cholesky in long double accumulation
  for (int j = 0; j < CB; ++j) {
    long double d = T[j * CB + j];
    for (int m = 0; m < j; ++m) d -= (long double)Lref[m * CB + j] * Lref[m * CB + j];
    const double ljj = (double)sqrtl(d);
    Lref[j * CB + j] = ljj;
    for (int i = j + 1; i < CB; ++i) {
      long double v = T[j * CB + i];
      for (int m = 0; m < j; ++m) v -= (long double)Lref[m * CB + i] * Lref[m * CB + j];
      Lref[j * CB + i] = (double)(v / ljj);
    }
  }
  double *dA, *dL, *dW;
  hipMalloc((void**)&dA, CB * CB * 8);
  hipMalloc((void**)&dL, CB * CB * 8);
  hipMalloc((void**)&dW, CB * CB * 8);
  hipMemcpy(dA, A.data(), CB * CB * 8, hipMemcpyHostToDevice);
  run<0>("V0 sqrt + divide, readlane", dA, dL, dW, Lref, A);
  run<1>("V1 rsq + Newton, readlane", dA, dL, dW, Lref, A);
  run<2>("V2 rsq + Newton, LDS broadcast", dA, dL, dW, Lref, A);
  run<3>("V3 rsq + Newton, readlane, + inverse rows", dA, dL, dW, Lref, A);
  run<4>("V4 rsq + Newton, ds_bpermute", dA, dL, dW, Lref, A);
  run<5>("V5 rsq + Newton, readlane in fenced groups", dA, dL, dW, Lref, A);
  run<6>("V6 rsq + Newton, LDS broadcast b128", dA, dL, dW, Lref, A);
  return 0;
}
