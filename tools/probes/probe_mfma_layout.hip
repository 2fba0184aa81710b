// Empirically derive the lane<->element mapping of v_mfma_f64_4x4x4_4b_f64 and v_mfma_f64_16x16x4_f64 on gfx950.
// For every (la, lb): A = one-hot at lane la, B = one-hot at lane lb, C = 0; record which (lane, reg) of D is non-zero.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe4(int* out) {  // out[la*64+lb] = lane index with nonzero D, or -1 ; -2 if several
  int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) out[la * 64 + lb] = m == 0 ? -1 : (__popcll(m) == 1 ? __ffsll((long long)m) - 1 : -2);
    }
}
__global__ void probe16(int* out) {  // out = lane*4+reg of nonzero D
  int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      d4 c = {0, 0, 0, 0};
      d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
      int res = -1;
      for (int r = 0; r < 4; ++r) {
        unsigned long long m = __ballot(d[r] != 0.0);
        if (m) res = (__ffsll((long long)m) - 1) * 4 + r + (__popcll(m) > 1 ? 100000 : 0);
      }
      if (lane == 0) out[la * 64 + lb] = res;
    }
}
int main() {
  int *d, h[4096];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(probe4, 1, 64, 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("== v_mfma_f64_4x4x4_4b_f64: for A-lane la (rows) x B-lane lb: D lane (.. = none)\n");
  for (int la = 0; la < 64; ++la) {
    printf("la=%2d:", la);
    for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb] >= 0) printf(" (lb=%d->%d)", lb, h[la * 64 + lb]); else if (h[la*64+lb]==-2) printf(" (lb=%d->MULTI)", lb);
    printf("\n");
  }
  hipLaunchKernelGGL(probe16, 1, 64, 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("== v_mfma_f64_16x16x4_f64: la x lb -> D lane*4+reg (first 20 A lanes)\n");
  for (int la = 0; la < 20; ++la) {
    printf("la=%2d:", la);
    int c = 0;
    for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb] >= 0 && c++ < 18) printf(" (%d->l%d r%d)", lb, h[la * 64 + lb] / 4, h[la * 64 + lb] % 4);
    printf("\n");
  }
  return 0;
}
