// which way does DPP row_ror:n move data on gfx950?  out[n][lane] = source lane seen by `lane` after row_ror:n
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> __device__ int ror(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x120 | N, 0xf, 0xf, true); }
__global__ void k(int* out) {
  int l = threadIdx.x;
  out[0 * 64 + l] = ror<4>(l);
  out[1 * 64 + l] = ror<8>(l);
  out[2 * 64 + l] = ror<12>(l);
}
int main() {
  int *d, h[192];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int n = 0; n < 3; ++n) { printf("row_ror:%d :", 4 * (n + 1)); for (int l = 0; l < 20; ++l) printf(" %d", h[n * 64 + l]); printf(" ... l=63:%d\n", h[n * 64 + 63]); }
}
