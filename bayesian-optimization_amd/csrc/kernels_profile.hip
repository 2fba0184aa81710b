// kernels_profile.hip -- bogp_selftest_profile: the radial profile of a correlation function (or one of its building blocks) evaluated on the
// device for an array of arguments, so that tests can hold the special functions of bogp_device.h (K_nu, 1 / Gamma) to committed
// tables one argument at a time instead of through a posterior.  See include/bogp.h.
#include "bogp_device.h"
#include "bogp_handle.h"

using namespace bogp;

namespace {

template <int KERNEL>
__global__ void k_profile(const double* __restrict__ in, double* __restrict__ out, int64_t n, double pexp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = corr_profile<KERNEL>(in[i], pexp);
}
__global__ void k_special(const double* __restrict__ in, double* __restrict__ out, int64_t n, double pexp, int what) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double rg;
  switch (what) {
    case BOGP_SELFTEST_BESSEL_K: out[i] = bessel_k_nu(pexp, in[i]); break;
    case BOGP_SELFTEST_BESSEL_K_PAIRS: out[i] = bessel_k_nu(in[2 * i], in[2 * i + 1]); break;
    case BOGP_SELFTEST_MATERN_NU_PAIRS: out[i] = corr_profile<BOGP_KERNEL_MATERN_NU>(in[2 * i + 1], in[2 * i]); break;
    default: bessel_k_nu(in[i], 1.0, &rg); out[i] = rg; break;
  }
}

}  // namespace

extern "C" int bogp_selftest_profile(bogp_handle* h, int what, int kernel, double pexp, const double* arg, int64_t n, double* out) {
  if (!h) return BOGP_ERR_INVALID;
  if (!arg || !out || n <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_selftest_profile: null pointer or empty array");
  if (what < BOGP_SELFTEST_PROFILE || what > BOGP_SELFTEST_MATERN_NU_PAIRS) FAIL(h, BOGP_ERR_INVALID, "bogp_selftest_profile: unknown selector %d", what);
  if (what == BOGP_SELFTEST_PROFILE && (kernel < 0 || kernel > BOGP_KERNEL_MATERN_NU)) FAIL(h, BOGP_ERR_INVALID, "bogp_selftest_profile: unknown kernel id %d", kernel);
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = h->stream;
  double *din = nullptr, *dout = nullptr;
  const int64_t nin = what >= BOGP_SELFTEST_BESSEL_K_PAIRS ? 2 * n : n;  // the *_PAIRS selectors read (order, argument) pairs
  if (hipMalloc((void**)&din, nin * sizeof(double)) != hipSuccess || hipMalloc((void**)&dout, n * sizeof(double)) != hipSuccess) {
    dfree(din); dfree(dout);
    FAIL(h, BOGP_ERR_HIP, "bogp_selftest_profile: hipMalloc failed");
  }
  hipError_t e = hipMemcpyAsync(din, arg, nin * sizeof(double), hipMemcpyHostToDevice, st);
  const dim3 grid((unsigned)((n + 255) / 256));
  if (e == hipSuccess) {
    if (what != BOGP_SELFTEST_PROFILE) {
      hipLaunchKernelGGL(k_special, grid, 256, 0, st, din, dout, n, pexp, what);
    } else {
      switch (kernel) {
#define BOGP_CASE(K) case K: hipLaunchKernelGGL(k_profile<K>, grid, 256, 0, st, din, dout, n, pexp); break;
        BOGP_CASE(BOGP_KERNEL_SE) BOGP_CASE(BOGP_KERNEL_MATERN12) BOGP_CASE(BOGP_KERNEL_MATERN32) BOGP_CASE(BOGP_KERNEL_MATERN52)
        BOGP_CASE(BOGP_KERNEL_ABSEXP) BOGP_CASE(BOGP_KERNEL_CUBIC) BOGP_CASE(BOGP_KERNEL_GENEXP) BOGP_CASE(BOGP_KERNEL_MATERN_NU)
#undef BOGP_CASE
      }
    }
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, dout, n * sizeof(double), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  dfree(din); dfree(dout);
  if (e != hipSuccess) FAIL(h, BOGP_ERR_HIP, "bogp_selftest_profile: %s", hipGetErrorString(e));
  return BOGP_OK;
}
