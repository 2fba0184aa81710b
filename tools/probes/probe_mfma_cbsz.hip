// Does v_mfma_f64_4x4x4_4b_f64 honour CBSZ/ABID (broadcast one A block to all 4 blocks) on gfx950?
// For cbsz=2, abid=t: A one-hot at lane la, B one-hot at lane lb -> list D lanes that are non-zero.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ABID>
__device__ unsigned long long run(int lane, int la, int lb) {
  double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 2, ABID, 0);
  return __ballot(d != 0.0);
}
__global__ void probe(unsigned long long* out) {
  int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      unsigned long long m0 = run<0>(lane, la, lb), m1 = run<1>(lane, la, lb), m2 = run<2>(lane, la, lb), m3 = run<3>(lane, la, lb);
      if (lane == 0) { out[((0 * 64 + la) * 64 + lb)] = m0; out[((1 * 64 + la) * 64 + lb)] = m1; out[((2 * 64 + la) * 64 + lb)] = m2; out[((3 * 64 + la) * 64 + lb)] = m3; }
    }
}
int main() {
  unsigned long long* d; static unsigned long long h[4 * 64 * 64];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(probe, 1, 64, 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int t = 0; t < 4; ++t) {
    printf("== cbsz=2 abid=%d: A lanes that matter and where (la: lb->Dlane ...), first 3 hits per la\n", t);
    for (int la = 0; la < 64; ++la) {
      int c = 0;
      for (int lb = 0; lb < 64; ++lb) { unsigned long long m = h[(t * 64 + la) * 64 + lb]; if (m) { if (c == 0) printf("la=%2d:", la); if (c < 6) { printf(" (lb=%d->", lb); for (int l = 0; l < 64; ++l) if (m >> l & 1) printf("%d,", l); printf(")"); } ++c; } }
      if (c) printf("  [%d hits]\n", c);
    }
  }
}
