// bogp_api.hip -- the C ABI of libbogp.so (include/bogp.h): device state, rocSOLVER/rocBLAS orchestration of the
// fit path, and the chunked posterior/acquisition sweep.  No host fallback exists: every numerical step runs on
// the gfx950 device, and every failure is reported as an error code + message.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bogp.h"
#include "bogp_internal.h"

using namespace bogp;

static std::string g_create_error;

struct bogp_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;  // producer stream: k_corr_chunk of chunk c+1 runs beside k_contract of chunk c
  rocblas_handle blas = nullptr;
  std::string err;

  // training set
  int N = 0, d = 0, Np = 0;
  int ldr = 0;  // leading dimension of dR / dV / dRinv: N rounded up to 64 (identity padding, kernels_chol.hip)
  double *dX = nullptr, *dy = nullptr;

  // factorisation workspace (column-major, ld = ldr)
  double *dR = nullptr, *dV = nullptr, *dU = nullptr, *dT = nullptr, *dRinv = nullptr;  // L, L^-1, L^-T, scratch, R^-1
  double* dones = nullptr;  // N ones (the constant trend basis)
  double* dgemv_scratch = nullptr;  // segment partials of launch_gemv2
  std::vector<double> h_theta, h_sqrt_theta;
  double* ddinv = nullptr;  // ldr x 64: inverses of the diagonal blocks of the running factorisation (kernels_chol.hip)
  double *dyt = nullptr, *dft = nullptr, *drho = nullptr, *dtmp = nullptr;  // N each
  double *dgamma = nullptr, *dw = nullptr;                                  // Np each (zero padded)
  double *dtheta = nullptr, *dsqrt_theta = nullptr;                         // d each
  double* dscal = nullptr;                                                  // small scalar scratch
  rocblas_int* dinfo = nullptr;
  double* dgrad_partial = nullptr;
  size_t grad_partial_cap = 0;
  double* dbatch = nullptr;
  size_t batch_cap = 0;

  // committed state
  bool committed = false;
  int kernel = 0, mode = 0, estimate_trend = 0;
  double beta = 0, G = 0, sigma2 = 0, noise_var = 0, llf = 0, ftft = 0;
  double* dXthT = nullptr;  // [d][Np]
  double2* dVp = nullptr;   // [Np/16][Np/8][64]

  // candidates
  const double* dXs = nullptr;
  double* dXs_owned = nullptr;
  size_t xs_cap = 0;
  double* dbounds = nullptr;
  size_t bounds_cap = 0;
  int64_t M = 0;

  // sweep scratch
  double *drT[2] = {nullptr, nullptr}, *dmu_part[2] = {nullptr, nullptr}, *dw_part[2] = {nullptr, nullptr};
  double* dss_part = nullptr;
  size_t rT_cap[2] = {0, 0}, mu_part_cap[2] = {0, 0}, w_part_cap[2] = {0, 0}, ss_part_cap = 0;
  double *dblk_val = nullptr, *dmu_out = nullptr, *dmse_out = nullptr, *dacq_out = nullptr, *dbest_val = nullptr;
  int64_t *dblk_idx = nullptr, *dbest_idx = nullptr;
  size_t blk_val_cap = 0, blk_idx_cap = 0, mu_out_cap = 0, mse_out_cap = 0, acq_out_cap = 0;

  // timing of the last sweep/predict
  std::vector<hipEvent_t> ev;
  double t_corr_ms = 0, t_contract_ms = 0, t_acq_ms = 0;
  int n_chunks = 0;
};

#define FAIL(h, code, ...)                              \
  do {                                                  \
    char _b[512];                                       \
    snprintf(_b, sizeof(_b), __VA_ARGS__);              \
    (h)->err = _b;                                      \
    return (code);                                      \
  } while (0)
#define HIPCHK(h, expr)                                                                                  \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess) FAIL(h, BOGP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)
#define BLASCHK(h, expr)                                                                          \
  do {                                                                                            \
    rocblas_status _s = (expr);                                                                   \
    if (_s != rocblas_status_success)                                                             \
      FAIL(h, BOGP_ERR_HIP, "%s failed: rocblas_status %d (%s:%d)", #expr, (int)_s, __FILE__, __LINE__); \
  } while (0)

template <typename T>
static int ensure(bogp_handle* h, T** p, size_t* cap, size_t n) {
  if (*cap >= n && *p) return BOGP_OK;
  if (*p) HIPCHK(h, hipFree(*p));
  *p = nullptr;
  *cap = 0;
  HIPCHK(h, hipMalloc((void**)p, n * sizeof(T)));
  *cap = n;
  return BOGP_OK;
}
template <typename T>
static void dfree(T*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

extern "C" int bogp_abi_version(void) { return 1; }

extern "C" const char* bogp_last_error(const bogp_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" int bogp_create(int device, bogp_handle** out) {
  if (!out) return BOGP_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_error = std::string("no HIP device: ") + hipGetErrorString(e);
    return BOGP_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) {
    g_create_error = "device index out of range";
    return BOGP_ERR_INVALID;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    g_create_error = "hipGetDeviceProperties failed";
    return BOGP_ERR_HIP;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_error = std::string("libbogp is built for gfx950 only; device is ") + prop.gcnArchName;
    return BOGP_ERR_NO_DEVICE;
  }
  bogp_handle* h = new bogp_handle();
  h->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) != hipSuccess ||
      rocblas_create_handle(&h->blas) != rocblas_status_success ||
      rocblas_set_stream(h->blas, h->stream) != rocblas_status_success ||
      hipMalloc((void**)&h->dinfo, sizeof(rocblas_int)) != hipSuccess ||
      hipMalloc((void**)&h->dscal, 64 * sizeof(double)) != hipSuccess) {
    g_create_error = "stream / rocBLAS handle creation failed";
    delete h;
    return BOGP_ERR_HIP;
  }
  rocblas_set_pointer_mode(h->blas, rocblas_pointer_mode_host);
  *out = h;
  return BOGP_OK;
}

static void free_train(bogp_handle* h) {
  dfree(h->dX); dfree(h->dy); dfree(h->dR); dfree(h->dV); dfree(h->dU); dfree(h->dT); dfree(h->dRinv); dfree(h->ddinv); dfree(h->dones); dfree(h->dgemv_scratch);
  dfree(h->dyt); dfree(h->dft); dfree(h->drho); dfree(h->dtmp); dfree(h->dgamma); dfree(h->dw);
  dfree(h->dtheta); dfree(h->dsqrt_theta); dfree(h->dXthT); dfree(h->dVp);
  h->committed = false;
}

extern "C" void bogp_destroy(bogp_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  free_train(h);
  (void)hipStreamSynchronize(h->stream2);
  dfree(h->dXs_owned); dfree(h->dss_part); dfree(h->dbounds);
  for (int b = 0; b < 2; ++b) { dfree(h->drT[b]); dfree(h->dmu_part[b]); dfree(h->dw_part[b]); }
  dfree(h->dblk_val); dfree(h->dblk_idx); dfree(h->dmu_out); dfree(h->dmse_out); dfree(h->dacq_out);
  dfree(h->dbest_val); dfree(h->dbest_idx); dfree(h->dinfo); dfree(h->dscal); dfree(h->dgrad_partial); dfree(h->dbatch);
  for (auto e : h->ev) (void)hipEventDestroy(e);
  if (h->blas) rocblas_destroy_handle(h->blas);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  delete h;
}

extern "C" int bogp_set_train(bogp_handle* h, const double* X, const double* y, int N, int d, int n_targets) {
  if (!h) return BOGP_ERR_INVALID;
  if (!X || !y || N <= 0 || d <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_set_train: X, y must be non-null and N, d > 0");
  if (n_targets != 1) FAIL(h, BOGP_ERR_UNSUPPORTED, "bogp_set_train: n_targets = %d; only single-target GPs are built (multi-target y is MOBO-only)", n_targets);
  if (d > 128) FAIL(h, BOGP_ERR_UNSUPPORTED, "bogp_set_train: d = %d > 128: the sweep producer keeps a 64 x d candidate tile in 64 KB of LDS", d);
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  free_train(h);
  h->N = N;
  h->d = d;
  h->Np = ((N + 31) / 32) * 32;
  h->ldr = ((N + 63) / 64) * 64;
  const size_t NN = (size_t)h->ldr * h->ldr;
  HIPCHK(h, hipMalloc((void**)&h->dX, (size_t)N * d * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&h->dy, N * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&h->dR, NN * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&h->ddinv, (size_t)h->ldr * 64 * sizeof(double)));
  HIPCHK(h, launch_pad_identity(h->dR, N, h->ldr, h->stream));
  // V = L^-1 and U = V^T keep exact zeros in their other triangle (set once here; kernels_chol.hip never writes there)
  HIPCHK(h, hipMalloc((void**)&h->dV, NN * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&h->dU, NN * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&h->dT, NN * sizeof(double)));
  HIPCHK(h, hipMemsetAsync(h->dV, 0, NN * sizeof(double), h->stream));
  HIPCHK(h, hipMemsetAsync(h->dU, 0, NN * sizeof(double), h->stream));
  {
    std::vector<double> ones(N, 1.0);
    HIPCHK(h, hipMalloc((void**)&h->dones, N * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dgemv_scratch, gemv2_scratch_doubles(N) * sizeof(double)));
    HIPCHK(h, hipMemcpy(h->dones, ones.data(), N * sizeof(double), hipMemcpyHostToDevice));
  }
  HIPCHK(h, hipMalloc((void**)&h->dyt, N * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&h->dft, N * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&h->drho, N * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&h->dtmp, N * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&h->dgamma, h->Np * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&h->dw, h->Np * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&h->dtheta, d * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&h->dsqrt_theta, d * sizeof(double)));
  HIPCHK(h, hipMemcpyAsync(h->dX, X, (size_t)N * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->dy, y, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return BOGP_OK;
}

// ------------------------------------------------------------------------------------------------------
// factorise at `par` (shared by bogp_nll and bogp_commit)
// ------------------------------------------------------------------------------------------------------
struct FitOut {
  double llf = 0, sigma2 = 0, noise_var = 0, s2t = 0, G = 0, beta = 0, ftyt = 0, ftft = 0;
};

static int factorize(bogp_handle* h, int kernel, int mode, const double* par, int n_par, double noise_var, int trend,
                     int estimate_trend, double beta, bool want_gamma, FitOut* o, std::vector<double>* theta_out) {
  if (!h->dX) FAIL(h, BOGP_ERR_INVALID, "no training set: call bogp_set_train first");
  if (kernel < 0 || kernel > BOGP_KERNEL_ABSEXP) FAIL(h, BOGP_ERR_INVALID, "unknown kernel id %d", kernel);
  if (mode < 0 || mode > 2) FAIL(h, BOGP_ERR_INVALID, "unknown estimation mode %d", mode);
  if (trend != BOGP_TREND_CONSTANT) FAIL(h, BOGP_ERR_UNSUPPORTED, "only the constant trend basis is built (trend id %d)", trend);
  const int N = h->N, d = h->d, ldr = h->ldr;
  const int n_theta = n_par - (mode == BOGP_MODE_NOISELESS ? 0 : 1);
  if (n_theta != d && n_theta != 1) FAIL(h, BOGP_ERR_INVALID, "len(theta) = %d must be 1 or d = %d", n_theta, d);
  std::vector<double>& th = h->h_theta;  // handle-owned: the asynchronous uploads below outlive this scope
  std::vector<double>& sth = h->h_sqrt_theta;
  th.resize(d);
  sth.resize(d);
  for (int k = 0; k < d; ++k) {
    th[k] = par[n_theta == 1 ? 0 : k];
    if (!(th[k] > 0) || !std::isfinite(th[k])) FAIL(h, BOGP_ERR_INVALID, "theta[%d] = %g must be finite and > 0", k, th[k]);
    // coordinates are pre-scaled so that the producer forms (a - b)^2 (radial kernels) or |a - b| (absolute_exponential)
    sth[k] = kernel == BOGP_KERNEL_ABSEXP ? th[k] : std::sqrt(th[k]);
  }
  if (theta_out) *theta_out = th;
  hipStream_t st = h->stream;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(h->dtheta, th.data(), d * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(h->dsqrt_theta, sth.data(), d * sizeof(double), hipMemcpyHostToDevice, st));

  // correlation matrix with the per-mode normalisation (gpr.py:931-969)
  double s2t = 0, alpha = 0, sigma2_par = 0;
  if (mode == BOGP_MODE_NOISELESS) {
    HIPCHK(h, launch_build_R(kernel, h->dX, N, d, h->dtheta, 1.0, 1.0, h->dR, ldr, st));
  } else if (mode == BOGP_MODE_NOISE_ESTIM) {
    alpha = par[n_par - 1];
    HIPCHK(h, launch_build_R(kernel, h->dX, N, d, h->dtheta, alpha, alpha * 1.0 + (1 - alpha) * 1.0, h->dR, ldr, st));
  } else {
    sigma2_par = par[n_par - 1];
    s2t = sigma2_par + noise_var;
    HIPCHK(h, launch_build_R_div(kernel, h->dX, N, d, h->dtheta, sigma2_par, s2t, (sigma2_par * 1.0 + noise_var * 1.0) / s2t,
                                 h->dR, ldr, st));
  }
  // The whole evaluation is queued without a host round trip and read back once:
  //   L = chol(R) (gpr.py:795)                      kernels_chol.hip
  //   V = L^-1, U = L^-T                            every triangular solve of :799-808 / :787-788 / :997 becomes a product
  //   Yt = V y (:799), Ft = V 1 (:803)              one pass over V
  //   rho (:806 / :808), |Ft|, Ft.Yt, rho.rho       k_fit_rho
  //   gamma = U rho (:788 / :996)
  HIPCHK(h, launch_chol_lower(h->dR, ldr, h->ddinv, h->dinfo, st));
  HIPCHK(h, launch_logdet(h->dR, N, ldr, h->dscal, st));
  HIPCHK(h, launch_tri_inverse(h->dR, h->ddinv, h->dV, h->dU, h->dT, ldr, st));
  HIPCHK(h, launch_gemv2(h->dV, ldr, N, 1, h->dy, h->dones, h->dyt, h->dft, h->dgemv_scratch, st));
  HIPCHK(h, launch_fit_rho(h->dyt, h->dft, N, estimate_trend, beta, h->drho, h->dscal, st));
  if (want_gamma) {
    HIPCHK(h, hipMemsetAsync(h->dgamma, 0, h->Np * sizeof(double), st));
    HIPCHK(h, launch_gemv2(h->dU, ldr, N, 2, h->drho, nullptr, h->dgamma, nullptr, h->dgemv_scratch, st));
  }
  rocblas_int info = 0;
  double sc[4] = {0, 0, 0, 0};  // sum(log diag L), |Ft|, Ft.Yt, rho.rho
  HIPCHK(h, hipMemcpyAsync(&info, h->dinfo, sizeof(info), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(sc, h->dscal, sizeof(sc), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  if (info != 0) FAIL(h, BOGP_ERR_NOT_POSDEF, "correlation matrix is not positive definite (potrf info = %d)", (int)info);

  const double logdet = sc[0], rho_ss = sc[3];
  double ftyt = 0, ftft = 0, G = 0, beta_eff = beta;
  if (estimate_trend) {
    // economic QR of the single column Ft: G = -sign(Ft[0]) |Ft|, Ft[0] = 1 / L[0][0] > 0 (:803-806)
    const double nrm = sc[1];
    ftyt = sc[2];
    G = -nrm;
    ftft = nrm * nrm;
    const double qty = ftyt / G;  // Q^T Yt
    beta_eff = qty / G;           // beta = G^-1 Q^T Yt (:785-787)
  }

  const double TWO_PI = 2.0 * 3.141592653589793;
  double llf, sigma2, nv;
  if (mode == BOGP_MODE_NOISELESS) {  // :941-945
    const int k = estimate_trend ? 1 : 0;
    sigma2 = rho_ss / (N - k);
    nv = 0;
    s2t = sigma2;
    llf = -0.5 * (N * std::log(TWO_PI * sigma2) + 2.0 * logdet + N);
  } else if (mode == BOGP_MODE_NOISE_ESTIM) {  // :954-958
    s2t = rho_ss / N;
    sigma2 = alpha * s2t;
    nv = (1 - alpha) * s2t;
    llf = -0.5 * (N * std::log(TWO_PI * s2t) + 2.0 * logdet + N);
  } else {  // :973-977
    sigma2 = sigma2_par;
    nv = noise_var;
    llf = -0.5 * (N * std::log(TWO_PI * s2t) + 2.0 * logdet + rho_ss / s2t);
  }
  if (!std::isfinite(llf)) FAIL(h, BOGP_ERR_NOT_POSDEF, "log-likelihood is not finite (%g): degenerate factorisation", llf);
  o->llf = llf; o->sigma2 = sigma2; o->noise_var = nv; o->s2t = s2t; o->G = G; o->beta = beta_eff; o->ftyt = ftyt; o->ftft = ftft;
  if (llf > 0) FAIL(h, BOGP_ERR_LLF_POSITIVE, "log-likelihood %g > 0 is rejected by the reference (gpr.py:981-982)", llf);

  return BOGP_OK;
}

extern "C" int bogp_nll(bogp_handle* h, int kernel, int mode, const double* par, int n_par, double noise_var, int trend,
                        int estimate_trend, double beta, double* llf, double* grad) {
  if (!h) return BOGP_ERR_INVALID;
  if (!par || !llf || n_par <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_nll: par/llf must be non-null");
  h->committed = false;  // the factor buffers are about to be overwritten
  FitOut o;
  int rc = factorize(h, kernel, mode, par, n_par, noise_var, trend, estimate_trend, beta, grad != nullptr, &o, nullptr);
  *llf = o.llf;
  if (rc != BOGP_OK) return rc;
  if (!grad) return BOGP_OK;

  const int N = h->N, d = h->d;
  const int n_theta = n_par - (mode == BOGP_MODE_NOISELESS ? 0 : 1);
  if (n_theta != d) FAIL(h, BOGP_ERR_UNSUPPORTED, "gradient with isotropic theta (len 1, d = %d) is not built: the reference's own gradient is inconsistent there (gpr.py:1001-1037 index the (N,N,d) tensor by parameter)", d);
  hipStream_t st = h->stream;
  // R^-1 = cho_solve(L, I) (:997) via potri on a copy of L
  const int ldr = h->ldr;
  if (!h->dRinv) HIPCHK(h, hipMalloc((void**)&h->dRinv, (size_t)UUT_PARTS * ldr * ldr * sizeof(double)));
  HIPCHK(h, launch_uut(h->dU, h->dRinv, ldr, st));  // R^-1 = L^-T L^-1, lower triangle
  const int nblk = grad_contract_blocks(N);
  int e = ensure(h, &h->dgrad_partial, &h->grad_partial_cap, (size_t)nblk * (d + 1) + (d + 3));
  if (e) return e;
  const double c1 = 1.0 / (mode == BOGP_MODE_NOISELESS ? o.sigma2 : o.s2t);
  HIPCHK(h, launch_grad_contract(kernel, h->dX, N, d, h->dtheta, h->dgamma, c1, h->dRinv, ldr, UUT_PARTS, (size_t)ldr * ldr, h->dgrad_partial, nblk, st));
  double* dS = h->dgrad_partial + (size_t)nblk * (d + 1);
  HIPCHK(h, launch_grad_reduce(h->dgrad_partial, nblk, d + 1, dS, st));
  std::vector<double> S(d + 3);
  if (mode == BOGP_MODE_NOISY) HIPCHK(h, launch_trace_gg(h->dRinv, ldr, UUT_PARTS, (size_t)ldr * ldr, N, h->dgamma, dS + d + 1, st));
  HIPCHK(h, hipMemcpyAsync(S.data(), dS, (d + 3) * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  const double tr = S[d + 1], gg = S[d + 2];
  if (mode == BOGP_MODE_NOISELESS) {
    for (int k = 0; k < d; ++k) grad[k] = S[k];
  } else if (mode == BOGP_MODE_NOISE_ESTIM) {
    const double alpha = par[n_par - 1];
    for (int k = 0; k < d; ++k) grad[k] = alpha * S[k];
    grad[d] = S[d];
  } else {
    for (int k = 0; k < d; ++k) grad[k] = S[k];
    grad[d] = -0.5 * (tr / o.s2t - gg / (o.s2t * o.s2t)) + S[d] / o.s2t;
  }
  return BOGP_OK;
}

extern "C" int bogp_commit(bogp_handle* h, int kernel, int mode, const double* par, int n_par, double noise_var, int trend,
                           int estimate_trend, double beta, double* llf) {
  if (!h) return BOGP_ERR_INVALID;
  if (!par || n_par <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_commit: par must be non-null");
  h->committed = false;
  FitOut o;
  std::vector<double> th;
  int rc = factorize(h, kernel, mode, par, n_par, noise_var, trend, estimate_trend, beta, true, &o, &th);
  if (llf) *llf = o.llf;
  if (rc != BOGP_OK) return rc;
  const int N = h->N, d = h->d, Np = h->Np;
  hipStream_t st = h->stream;
  // V = L^-1 (the triangular solve of gpr.py:494 becomes a triangular GEMM against V)
  const int ldr = h->ldr;
  if (!h->dVp) HIPCHK(h, hipMalloc((void**)&h->dVp, (size_t)Np * Np * sizeof(double)));
  HIPCHK(h, launch_pack_V(h->dV, N, ldr, Np, h->dVp, st));
  // w = L^-T Ft  (so that Ft^T L^-1 r = w . r, gpr.py:496-498)
  HIPCHK(h, hipMemsetAsync(h->dw, 0, Np * sizeof(double), st));
  if (estimate_trend) {
    HIPCHK(h, launch_gemv2(h->dU, ldr, N, 2, h->dft, nullptr, h->dw, nullptr, h->dgemv_scratch, st));
  }
  if (!h->dXthT) HIPCHK(h, hipMalloc((void**)&h->dXthT, (size_t)d * Np * sizeof(double)));
  HIPCHK(h, launch_scale_transpose(h->dX, N, d, Np, h->dsqrt_theta, h->dXthT, st));
  HIPCHK(h, hipStreamSynchronize(st));
  h->kernel = kernel; h->mode = mode; h->estimate_trend = estimate_trend;
  h->beta = o.beta; h->G = o.G; h->sigma2 = o.sigma2; h->noise_var = o.noise_var; h->llf = o.llf; h->ftft = o.ftft;
  h->committed = true;
  return BOGP_OK;
}

extern "C" int bogp_get_state(bogp_handle* h, double* C, double* gamma, double* rho, double* Yt, double* Ft, double* Q,
                              double* G, double* beta, double* sigma2, double* noise_var) {
  if (!h) return BOGP_ERR_INVALID;
  if (!h->committed) FAIL(h, BOGP_ERR_INVALID, "bogp_get_state: no committed state");
  const int N = h->N;
  hipStream_t st = h->stream;
  HIPCHK(h, hipSetDevice(h->device));
  if (C) {
    if (!h->dRinv) HIPCHK(h, hipMalloc((void**)&h->dRinv, (size_t)UUT_PARTS * h->ldr * h->ldr * sizeof(double)));
    HIPCHK(h, launch_copy_lower(h->dR, N, h->ldr, h->dRinv, st));
    HIPCHK(h, hipMemcpyAsync(C, h->dRinv, (size_t)N * N * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  if (gamma) HIPCHK(h, hipMemcpyAsync(gamma, h->dgamma, N * sizeof(double), hipMemcpyDeviceToHost, st));
  if (rho) HIPCHK(h, hipMemcpyAsync(rho, h->drho, N * sizeof(double), hipMemcpyDeviceToHost, st));
  if (Yt) HIPCHK(h, hipMemcpyAsync(Yt, h->dyt, N * sizeof(double), hipMemcpyDeviceToHost, st));
  if (h->estimate_trend) {
    if (Ft) HIPCHK(h, hipMemcpyAsync(Ft, h->dft, N * sizeof(double), hipMemcpyDeviceToHost, st));
    if (Q) HIPCHK(h, hipMemcpyAsync(Q, h->dft, N * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  HIPCHK(h, hipStreamSynchronize(st));
  if (h->estimate_trend && Q)
    for (int i = 0; i < N; ++i) Q[i] /= h->G;
  if (G) *G = h->G;
  if (beta) *beta = h->beta;
  if (sigma2) *sigma2 = h->sigma2;
  if (noise_var) *noise_var = h->noise_var;
  return BOGP_OK;
}

// ------------------------------------------------------------------------------------------------------
// candidates
// ------------------------------------------------------------------------------------------------------
extern "C" int bogp_candidates_upload(bogp_handle* h, const double* Xs, int64_t M) {
  if (!h) return BOGP_ERR_INVALID;
  if (!h->dX) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_upload: call bogp_set_train first (d is unknown)");
  if (!Xs || M <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_upload: Xs must be non-null and M > 0");
  HIPCHK(h, hipSetDevice(h->device));
  int e = ensure(h, &h->dXs_owned, &h->xs_cap, (size_t)M * h->d);
  if (e) return e;
  HIPCHK(h, hipMemcpyAsync(h->dXs_owned, Xs, (size_t)M * h->d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->dXs = h->dXs_owned;
  h->M = M;
  return BOGP_OK;
}

extern "C" int bogp_candidates_generate(bogp_handle* h, const double* lo, const double* hi, int64_t M, uint64_t seed,
                                        int64_t first_row) {
  if (!h) return BOGP_ERR_INVALID;
  if (!h->dX) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_generate: call bogp_set_train first (d is unknown)");
  if (!lo || !hi || M <= 0 || first_row < 0) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_generate: bounds must be non-null, M > 0, first_row >= 0");
  const int d = h->d;
  for (int k = 0; k < d; ++k)
    if (!(std::isfinite(lo[k]) && std::isfinite(hi[k]) && lo[k] <= hi[k])) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_generate: bad bounds in dimension %d", k);
  HIPCHK(h, hipSetDevice(h->device));
  int e = ensure(h, &h->dXs_owned, &h->xs_cap, (size_t)M * d);
  if (e) return e;
  if ((e = ensure(h, &h->dbounds, &h->bounds_cap, (size_t)2 * d))) return e;
  HIPCHK(h, hipMemcpyAsync(h->dbounds, lo, d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->dbounds + d, hi, d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, launch_generate_uniform(h->dXs_owned, M * d, d, h->dbounds, h->dbounds + d, seed, (uint64_t)first_row * (uint64_t)d, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));  // lo / hi are caller memory
  h->dXs = h->dXs_owned;
  h->M = M;
  return BOGP_OK;
}

extern "C" int bogp_candidates_read(bogp_handle* h, const int64_t* rows, int n, double* out) {
  if (!h) return BOGP_ERR_INVALID;
  if (!h->dXs || h->M <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_read: no candidates");
  if (!rows || !out || n < 0) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_read: null pointer");
  HIPCHK(h, hipSetDevice(h->device));
  const int d = h->d;
  for (int i = 0; i < n; ++i) {
    if (rows[i] < 0 || rows[i] >= h->M) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_read: row %lld outside [0, %lld)", (long long)rows[i], (long long)h->M);
    HIPCHK(h, hipMemcpyAsync(out + (size_t)i * d, h->dXs + (size_t)rows[i] * d, d * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return BOGP_OK;
}

extern "C" int bogp_candidates_bind(bogp_handle* h, const void* d_Xs, int64_t M) {
  if (!h) return BOGP_ERR_INVALID;
  if (!d_Xs || M <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_bind: pointer must be non-null and M > 0");
  h->dXs = (const double*)d_Xs;
  h->M = M;
  return BOGP_OK;
}

// ------------------------------------------------------------------------------------------------------
// posterior + acquisition sweep
// ------------------------------------------------------------------------------------------------------
static hipEvent_t get_event(bogp_handle* h, size_t i) {
  while (h->ev.size() <= i) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    h->ev.push_back(e);
  }
  return h->ev[i];
}

static int run_sweep(bogp_handle* h, bool want_out, int q, const int* acq_id, const double* acq_par, double plugin,
                     int minimize, bool want_acq_out) {
  if (!h->committed) FAIL(h, BOGP_ERR_INVALID, "no committed model: call bogp_commit first");
  if (!h->dXs || h->M <= 0) FAIL(h, BOGP_ERR_INVALID, "no candidates: call bogp_candidates_upload/bind first");
  if (q < 0 || q > BOGP_MAX_Q) FAIL(h, BOGP_ERR_INVALID, "q = %d outside [0, %d]", q, BOGP_MAX_Q);
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const int Np = h->Np, d = h->d;
  const int64_t M = h->M;
  const int64_t Mpad = ((M + 63) / 64) * 64;
  size_t chunk_bytes = (size_t)1 << 30;
  if (const char* env = getenv("BOGP_CHUNK_MB")) chunk_bytes = (size_t)std::max(1, atoi(env)) << 20;
  int64_t Mc = (int64_t)(chunk_bytes / ((size_t)Np * sizeof(double)) / 64) * 64;
  Mc = std::max<int64_t>(64, std::min<int64_t>(Mc, Mpad));
  const int nblk32 = Np / 32;
  // the training set is sliced into groups of 8 x 32 rows per producer workgroup: a function of N only, so that the
  // grouping of the partial sums of mu (hence every output bit) does not depend on the chunk size
  const int nblk_per_split = 8;
  const int S = (nblk32 + nblk_per_split - 1) / nblk_per_split;
  const int NJ16 = Np / 16;
  const int cols = contract_cols_per_group();
  const int nJ = (Np + cols - 1) / cols;
  const int64_t nchunk = (M + Mc - 1) / Mc;
  const int64_t nblk_total = (M + 255) / 256 + nchunk;  // per-chunk block counts are rounded up

  // Small batches (the reference's one-point-per-call usage through L-BFGS-B): the tiled contraction would leave one
  // workgroup walking all N columns alone (~0.3 ms at N = 2048).  For M <= BOGP_SMALL_M the posterior is instead
  // r -> rt = V r (rocBLAS dtrmm with M right-hand sides) -> column reductions, feeding the same acquisition kernel.
  int small_m = 32;
  if (const char* env = getenv("BOGP_SMALL_M")) small_m = atoi(env);
  if (M <= small_m) {
    const int B = (int)M, N = h->N;
    int e2;
    if ((e2 = ensure(h, &h->dbatch, &h->batch_cap, (size_t)3 * N * B + 3 * (size_t)B))) return e2;
    double* dr = h->dbatch;
    double* ds2 = dr + (size_t)N * B;
    double* drt = ds2 + (size_t)N * B;
    double* dred = drt + (size_t)N * B;  // mu[B], wd[B], ss[B]
    if (q > 0) {
      if ((e2 = ensure(h, &h->dblk_val, &h->blk_val_cap, (size_t)q * 2))) return e2;
      if ((e2 = ensure(h, &h->dblk_idx, &h->blk_idx_cap, (size_t)q * 2))) return e2;
      if (!h->dbest_val) HIPCHK(h, hipMalloc((void**)&h->dbest_val, BOGP_MAX_Q * sizeof(double)));
      if (!h->dbest_idx) HIPCHK(h, hipMalloc((void**)&h->dbest_idx, BOGP_MAX_Q * sizeof(int64_t)));
    }
    if (want_out) {
      if ((e2 = ensure(h, &h->dmu_out, &h->mu_out_cap, (size_t)M))) return e2;
      if ((e2 = ensure(h, &h->dmse_out, &h->mse_out_cap, (size_t)M))) return e2;
    }
    if (want_acq_out)
      if ((e2 = ensure(h, &h->dacq_out, &h->acq_out_cap, (size_t)q * M))) return e2;
    HIPCHK(h, launch_batch_corr(h->kernel, h->dX, N, d, h->dtheta, h->dXs, B, dr, ds2, st));
    const double one = 1.0;
    BLASCHK(h, rocblas_dtrmm(h->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, N, B, &one, h->dV, h->ldr, dr, N, drt, N));
    HIPCHK(h, launch_col_reduce(dr, drt, N, B, h->dgamma, h->dw, dred, dred + B, dred + 2 * B, st));
    AcqArgs aa;
    memset(&aa, 0, sizeof(aa));
    aa.mu_part = dred; aa.w_part = dred + B; aa.ss_part = dred + 2 * B; aa.S = 1; aa.nJ = 1; aa.Mc = B;
    aa.mcount = B; aa.m0 = 0; aa.beta = h->beta; aa.G = h->G; aa.estimate_trend = h->estimate_trend;
    aa.sigma2 = h->sigma2; aa.mu_out = want_out ? h->dmu_out : nullptr; aa.mse_out = want_out ? h->dmse_out : nullptr;
    aa.q = q;
    for (int i = 0; i < q; ++i) { aa.acq_id[i] = acq_id[i]; aa.acq_par[i] = acq_par ? acq_par[i] : 0.0; }
    aa.plugin = plugin; aa.minimize = minimize; aa.acq_out = want_acq_out ? h->dacq_out : nullptr; aa.M = M;
    aa.blk_val = h->dblk_val; aa.blk_idx = h->dblk_idx; aa.blk_offset = 0; aa.nblk_total = 1;
    HIPCHK(h, launch_acquisition(aa, st));
    if (q > 0) HIPCHK(h, launch_argmax_final(h->dblk_val, h->dblk_idx, 1, 1, q, h->dbest_val, h->dbest_idx, st));
    HIPCHK(h, hipStreamSynchronize(st));
    h->t_corr_ms = h->t_contract_ms = h->t_acq_ms = 0;
    h->n_chunks = 0;
    return BOGP_OK;
  }

  // Optional two-stream mode (BOGP_OVERLAP=1): the correlation producer of chunk c+1 (FP64 VALU) runs beside the
  // contraction of chunk c (FP64 MFMA), everything the producer writes double buffered.  Measured on MI355X (r01,
  // C3): the kernels do overlap (contract 75.9 -> 81.9 ms, corr 6.8 -> 11.7 ms) but the step time is unchanged
  // (83.3 -> 83.0 ms): the DP pipe is the shared resource.  Off by default: it costs a second 1-GiB chunk buffer.
  const bool overlap = nchunk > 1 && getenv("BOGP_OVERLAP") && atoi(getenv("BOGP_OVERLAP")) == 1;
  hipStream_t stP = overlap ? h->stream2 : st;
  const int nbuf = overlap ? 2 : 1;
  int e;
  for (int b = 0; b < nbuf; ++b) {
    if ((e = ensure(h, &h->drT[b], &h->rT_cap[b], (size_t)Np * Mc))) return e;
    if ((e = ensure(h, &h->dmu_part[b], &h->mu_part_cap[b], (size_t)S * Mc))) return e;
    if ((e = ensure(h, &h->dw_part[b], &h->w_part_cap[b], (size_t)S * Mc))) return e;
  }
  if ((e = ensure(h, &h->dss_part, &h->ss_part_cap, (size_t)nJ * Mc))) return e;
  if (q > 0) {
    if ((e = ensure(h, &h->dblk_val, &h->blk_val_cap, (size_t)q * nblk_total))) return e;
    if ((e = ensure(h, &h->dblk_idx, &h->blk_idx_cap, (size_t)q * nblk_total))) return e;
    if (!h->dbest_val) HIPCHK(h, hipMalloc((void**)&h->dbest_val, BOGP_MAX_Q * sizeof(double)));
    if (!h->dbest_idx) HIPCHK(h, hipMalloc((void**)&h->dbest_idx, BOGP_MAX_Q * sizeof(int64_t)));
  }
  if (want_out) {
    if ((e = ensure(h, &h->dmu_out, &h->mu_out_cap, (size_t)M))) return e;
    if ((e = ensure(h, &h->dmse_out, &h->mse_out_cap, (size_t)M))) return e;
  }
  if (want_acq_out)
    if ((e = ensure(h, &h->dacq_out, &h->acq_out_cap, (size_t)q * M))) return e;

  // events per chunk: [0] corr start, [1] corr end (producer stream); [2] contract start, [3] contract end,
  // [4] acquisition end = chunk done (main stream)
  constexpr int EPC = 5;
  for (int64_t c = 0; c < nchunk; ++c)
    for (int k = 0; k < EPC; ++k)
      if (!get_event(h, (size_t)(c * EPC + k))) FAIL(h, BOGP_ERR_HIP, "hipEventCreate failed");
  hipEvent_t ev_begin = get_event(h, (size_t)(nchunk * EPC));
  if (!ev_begin) FAIL(h, BOGP_ERR_HIP, "hipEventCreate failed");
  if (overlap) {  // the producer stream must see everything queued on the main stream so far (commit, uploads)
    HIPCHK(h, hipEventRecord(ev_begin, st));
    HIPCHK(h, hipStreamWaitEvent(stP, ev_begin, 0));
  }

  int64_t blk_offset = 0;
  for (int64_t c = 0; c < nchunk; ++c) {
    const int b = overlap ? (int)(c & 1) : 0;
    hipEvent_t* ev = &h->ev[(size_t)(c * EPC)];
    const int64_t m0 = c * Mc;
    const int64_t mcount = std::min<int64_t>(Mc, M - m0);
    const int64_t Mc_eff = ((mcount + 63) / 64) * 64;  // rows actually launched; array stride stays Mc
    CorrArgs ca;
    ca.Xs = h->dXs; ca.M = M; ca.m0 = m0; ca.Mc = Mc; ca.d = d; ca.Np = Np; ca.nblk_per_split = nblk_per_split;
    ca.sqrt_theta = h->dsqrt_theta; ca.XthT = h->dXthT; ca.gamma = h->dgamma; ca.wvec = h->dw;
    ca.rT = h->drT[b]; ca.mu_part = h->dmu_part[b]; ca.w_part = h->dw_part[b];
    ContractArgs ka;
    ka.rT = h->drT[b]; ka.Vp = h->dVp; ka.ss_part = h->dss_part; ka.Mc = Mc; ka.nMt = (int)(Mc_eff / 64); ka.nJ = nJ;
    ka.NJ16 = NJ16; ka.NKP = Np / 8;
    // producer: may reuse buffer b only after chunk c-2 (its previous user) is completely done
    if (overlap && c >= 2) HIPCHK(h, hipStreamWaitEvent(stP, h->ev[(size_t)((c - 2) * EPC + 4)], 0));
    HIPCHK(h, hipEventRecord(ev[0], stP));
    HIPCHK(h, launch_corr_chunk(h->kernel, ca, (int)(Mc_eff / 64), S, stP));
    HIPCHK(h, hipEventRecord(ev[1], stP));
    if (overlap) HIPCHK(h, hipStreamWaitEvent(st, ev[1], 0));
    HIPCHK(h, hipEventRecord(ev[2], st));
    HIPCHK(h, launch_contract(ka, st));
    HIPCHK(h, hipEventRecord(ev[3], st));
    AcqArgs aa;
    memset(&aa, 0, sizeof(aa));
    aa.mu_part = h->dmu_part[b]; aa.w_part = h->dw_part[b]; aa.ss_part = h->dss_part; aa.S = S; aa.nJ = nJ; aa.Mc = Mc;
    aa.mcount = mcount; aa.m0 = m0; aa.beta = h->beta; aa.G = h->G; aa.estimate_trend = h->estimate_trend;
    aa.sigma2 = h->sigma2; aa.mu_out = want_out ? h->dmu_out : nullptr; aa.mse_out = want_out ? h->dmse_out : nullptr;
    aa.q = q;
    for (int i = 0; i < q; ++i) { aa.acq_id[i] = acq_id[i]; aa.acq_par[i] = acq_par ? acq_par[i] : 0.0; }
    aa.plugin = plugin; aa.minimize = minimize; aa.acq_out = want_acq_out ? h->dacq_out : nullptr; aa.M = M;
    aa.blk_val = h->dblk_val; aa.blk_idx = h->dblk_idx; aa.blk_offset = blk_offset; aa.nblk_total = nblk_total;
    HIPCHK(h, launch_acquisition(aa, st));
    HIPCHK(h, hipEventRecord(ev[4], st));
    blk_offset += (mcount + 255) / 256;
  }
  if (q > 0) HIPCHK(h, launch_argmax_final(h->dblk_val, h->dblk_idx, blk_offset, nblk_total, q, h->dbest_val, h->dbest_idx, st));
  HIPCHK(h, hipStreamSynchronize(st));
  if (overlap) HIPCHK(h, hipStreamSynchronize(stP));
  h->t_corr_ms = h->t_contract_ms = h->t_acq_ms = 0;
  for (int64_t c = 0; c < nchunk; ++c) {
    float a = 0, b2 = 0, c2 = 0;
    hipEvent_t* ev = &h->ev[(size_t)(c * EPC)];
    (void)hipEventElapsedTime(&a, ev[0], ev[1]);
    (void)hipEventElapsedTime(&b2, ev[2], ev[3]);
    (void)hipEventElapsedTime(&c2, ev[3], ev[4]);
    h->t_corr_ms += a; h->t_contract_ms += b2; h->t_acq_ms += c2;
  }
  h->n_chunks = (int)nchunk;
  return BOGP_OK;
}

extern "C" int bogp_predict(bogp_handle* h, double* mu, double* mse) {
  if (!h) return BOGP_ERR_INVALID;
  if (!mu) FAIL(h, BOGP_ERR_INVALID, "bogp_predict: mu must be non-null");
  int rc = run_sweep(h, true, 0, nullptr, nullptr, 0.0, 1, false);
  if (rc) return rc;
  HIPCHK(h, hipMemcpy(mu, h->dmu_out, (size_t)h->M * sizeof(double), hipMemcpyDeviceToHost));
  if (mse) HIPCHK(h, hipMemcpy(mse, h->dmse_out, (size_t)h->M * sizeof(double), hipMemcpyDeviceToHost));
  return BOGP_OK;
}

extern "C" int bogp_sweep(bogp_handle* h, int q, const int* acq_id, const double* acq_par, double plugin, int minimize,
                          double* best_val, int64_t* best_idx, double* acq_out) {
  if (!h) return BOGP_ERR_INVALID;
  if (q <= 0 || !acq_id || !best_val || !best_idx) FAIL(h, BOGP_ERR_INVALID, "bogp_sweep: q > 0 and non-null acq_id/best_val/best_idx required");
  for (int i = 0; i < q; ++i) {
    if (acq_id[i] < 0 || acq_id[i] > 3) FAIL(h, BOGP_ERR_INVALID, "unknown acquisition id %d", acq_id[i]);
    const bool zero_ok = acq_id[i] == BOGP_ACQ_EPSILON_PI;  // epsilon = 0 is plain PI
    if (acq_id[i] != BOGP_ACQ_EI && (!acq_par || !(acq_par[i] > 0 || (zero_ok && acq_par[i] == 0))))
      FAIL(h, BOGP_ERR_INVALID, "acquisition parameter %d must be > 0 (the reference asserts alpha/epsilon/t > 0)", i);
  }
  int rc = run_sweep(h, false, q, acq_id, acq_par, plugin, minimize, acq_out != nullptr);
  if (rc) return rc;
  HIPCHK(h, hipMemcpy(best_val, h->dbest_val, q * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(best_idx, h->dbest_idx, q * sizeof(int64_t), hipMemcpyDeviceToHost));
  if (acq_out) HIPCHK(h, hipMemcpy(acq_out, h->dacq_out, (size_t)q * h->M * sizeof(double), hipMemcpyDeviceToHost));
  return BOGP_OK;
}

extern "C" int bogp_sweep_topk(bogp_handle* h, int q, const int* acq_id, const double* acq_par, double plugin, int minimize,
                               int k, double* best_val, int64_t* best_idx) {
  if (!h) return BOGP_ERR_INVALID;
  if (k <= 0 || k > BOGP_MAX_TOPK) FAIL(h, BOGP_ERR_INVALID, "bogp_sweep_topk: k = %d outside [1, %d]", k, BOGP_MAX_TOPK);
  if (q <= 0 || !acq_id || !best_val || !best_idx) FAIL(h, BOGP_ERR_INVALID, "bogp_sweep_topk: q > 0 and non-null acq_id/best_val/best_idx required");
  for (int i = 0; i < q; ++i) {
    if (acq_id[i] < 0 || acq_id[i] > 3) FAIL(h, BOGP_ERR_INVALID, "unknown acquisition id %d", acq_id[i]);
    const bool zero_ok = acq_id[i] == BOGP_ACQ_EPSILON_PI;
    if (acq_id[i] != BOGP_ACQ_EI && (!acq_par || !(acq_par[i] > 0 || (zero_ok && acq_par[i] == 0))))
      FAIL(h, BOGP_ERR_INVALID, "acquisition parameter %d must be > 0", i);
  }
  int rc = run_sweep(h, false, q, acq_id, acq_par, plugin, minimize, true);  // keeps the q x M values on the device
  if (rc) return rc;
  const int64_t M = h->M;
  const int64_t nblk = (M + 255) / 256;
  hipStream_t st = h->stream;
  // rank 0 is the sweep's own argmax; ranks 1..k-1 repeat the argmax with the winners so far masked out
  std::vector<int64_t> taken(k);
  for (int c = 0; c < q; ++c) {
    for (int r = 0; r < k; ++r) {
      if ((int64_t)r >= M) {  // fewer candidates than k: pad with (-inf, -1)
        best_val[c * k + r] = -INFINITY;
        best_idx[c * k + r] = -1;
        continue;
      }
      HIPCHK(h, launch_block_argmax_excl(h->dacq_out + (size_t)c * M, M, taken.data(), r, h->dblk_val, h->dblk_idx, st));
      HIPCHK(h, launch_argmax_final(h->dblk_val, h->dblk_idx, nblk, nblk, 1, h->dbest_val, h->dbest_idx, st));
      HIPCHK(h, hipMemcpyAsync(&best_val[c * k + r], h->dbest_val, sizeof(double), hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipMemcpyAsync(&best_idx[c * k + r], h->dbest_idx, sizeof(int64_t), hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
      taken[r] = best_idx[c * k + r];
    }
  }
  return BOGP_OK;
}

extern "C" int bogp_last_timing(bogp_handle* h, double* corr_ms, double* contract_ms, double* acquisition_ms, int* n_chunks) {
  if (!h) return BOGP_ERR_INVALID;
  if (corr_ms) *corr_ms = h->t_corr_ms;
  if (contract_ms) *contract_ms = h->t_contract_ms;
  if (acquisition_ms) *acquisition_ms = h->t_acq_ms;
  if (n_chunks) *n_chunks = h->n_chunks;
  return BOGP_OK;
}

extern "C" double bogp_flops_per_candidate(const bogp_handle* h) {
  if (!h || !h->committed) return 0.0;
  const double N = h->N, d = h->d, p = h->estimate_trend ? 1 : 0;
  return N * N + N * (3 * d + 5 + 2 * p);
}

// ------------------------------------------------------------------------------------------------------
// gradient of the posterior at one point (gpr.py:537-576)
// ------------------------------------------------------------------------------------------------------
extern "C" int bogp_gradient(bogp_handle* h, const double* x, double* dmu, double* dmse) {
  if (!h) return BOGP_ERR_INVALID;
  if (!h->committed) FAIL(h, BOGP_ERR_INVALID, "bogp_gradient: no committed model");
  if (!x || !dmu || !dmse) FAIL(h, BOGP_ERR_INVALID, "bogp_gradient: null pointer");
  const int N = h->N, d = h->d;
  hipStream_t st = h->stream;
  HIPCHK(h, hipSetDevice(h->device));
  int e = ensure(h, &h->dgrad_partial, &h->grad_partial_cap, (size_t)N * (d + 2) + 4 * d + 8);
  if (e) return e;
  double* dr = h->dgrad_partial;            // N
  double* drdx = dr + N;                    // d x N (column k = dr/dx_k)
  double* dz = drdx + (size_t)N * d;        // N
  double* dx = dz + N;                      // d
  double* dout = dx + d;                    // 3 d
  HIPCHK(h, hipMemcpyAsync(dx, x, d * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(h, launch_point_corr(h->kernel, h->dX, N, d, h->dtheta, dx, dr, drdx, st));
  // z = L^-T L^-1 r = V^T (V r) with the explicit V = L^-1 kept from the commit: two triangular matrix-vector
  // products (bandwidth bound, ~50 us at N = 2048) instead of two dependent triangular solves (~350 us each)
  HIPCHK(h, hipMemcpyAsync(dz, dr, N * sizeof(double), hipMemcpyDeviceToDevice, st));
  BLASCHK(h, rocblas_dtrmv(h->blas, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, N, h->dV, h->ldr, dz, 1));
  BLASCHK(h, rocblas_dtrmv(h->blas, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, N, h->dV, h->ldr, dz, 1));
  const double one = 1.0, zero = 0.0;
  BLASCHK(h, rocblas_dgemv(h->blas, rocblas_operation_transpose, N, d, &one, drdx, N, h->dgamma, 1, &zero, dout, 1));
  BLASCHK(h, rocblas_dgemv(h->blas, rocblas_operation_transpose, N, d, &one, drdx, N, dz, 1, &zero, dout + d, 1));
  double wr = 0;
  if (h->estimate_trend) {
    BLASCHK(h, rocblas_dgemv(h->blas, rocblas_operation_transpose, N, d, &one, drdx, N, h->dw, 1, &zero, dout + 2 * d, 1));
    BLASCHK(h, rocblas_ddot(h->blas, N, h->dw, 1, dr, 1, &wr));
  }
  std::vector<double> out(3 * d, 0.0);
  HIPCHK(h, hipMemcpyAsync(out.data(), dout, (h->estimate_trend ? 3 : 2) * d * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  for (int k = 0; k < d; ++k) {
    dmu[k] = out[k];  // beta^T f_dx = 0 for the constant basis
    double m = -1.0 * out[d + k];
    if (h->estimate_trend) m += (wr - 1.0) * (1.0 / h->ftft) * out[2 * d + k];
    dmse[k] = 2.0 * h->sigma2 * m;
  }
  return BOGP_OK;
}

// Batched flavour: B points, one pair of triangular solves with B right-hand sides
// (rocBLAS dtrsm = the reference's solve_triangular twice) and one reduction kernel; dmu, dmse are B x d row-major.
extern "C" int bogp_gradient_batch(bogp_handle* h, const double* Xb, int B, double* dmu, double* dmse) {
  if (!h) return BOGP_ERR_INVALID;
  if (!h->committed) FAIL(h, BOGP_ERR_INVALID, "bogp_gradient_batch: no committed model");
  if (!Xb || !dmu || !dmse || B <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_gradient_batch: null pointer or B <= 0");
  const int N = h->N, d = h->d;
  hipStream_t st = h->stream;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t nout = (size_t)B * (3 * d + 1);
  int e = ensure(h, &h->dbatch, &h->batch_cap, (size_t)3 * N * B + (size_t)B * d + nout);
  if (e) return e;
  double* dr = h->dbatch;               // N x B (column b = r of point b)
  double* ds2 = dr + (size_t)N * B;     // N x B
  double* dZ = ds2 + (size_t)N * B;     // N x B
  double* dXb = dZ + (size_t)N * B;     // B x d
  double* dout = dXb + (size_t)B * d;   // B x (3d + 1)
  HIPCHK(h, hipMemcpyAsync(dXb, Xb, (size_t)B * d * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(h, launch_batch_corr(h->kernel, h->dX, N, d, h->dtheta, dXb, B, dr, ds2, st));
  HIPCHK(h, hipMemcpyAsync(dZ, dr, (size_t)N * B * sizeof(double), hipMemcpyDeviceToDevice, st));
  const double one = 1.0;
  BLASCHK(h, rocblas_dtrsm(h->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, N, B, &one, h->dR, h->ldr, dZ, N));
  BLASCHK(h, rocblas_dtrsm(h->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, N, B, &one, h->dR, h->ldr, dZ, N));
  HIPCHK(h, launch_batch_grad(h->kernel, h->dX, N, d, h->dtheta, dXb, B, dr, ds2, dZ, h->dgamma, h->dw, dout, st));
  std::vector<double> out(nout);
  HIPCHK(h, hipMemcpyAsync(out.data(), dout, nout * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  for (int b = 0; b < B; ++b) {
    const double* o = &out[(size_t)b * (3 * d + 1)];
    const double wr = o[3 * d];
    for (int k = 0; k < d; ++k) {
      dmu[(size_t)b * d + k] = o[k];
      double m = -1.0 * o[d + k];
      if (h->estimate_trend) m += (wr - 1.0) * (1.0 / h->ftft) * o[2 * d + k];
      dmse[(size_t)b * d + k] = 2.0 * h->sigma2 * m;
    }
  }
  return BOGP_OK;
}

