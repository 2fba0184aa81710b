// bogp_api.hip -- the C ABI of libbogp.so (include/bogp.h): device state, orchestration of the fit path (in-tree kernels
// only: kernels_chol / kernels_fit / kernels_pairs / kernels_gemm -- no rocSOLVER, no rocBLAS), and the chunked posterior / acquisition sweep.  No host fallback exists: every numerical step runs on
// the gfx950 device, and every failure is reported as an error code + message.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bogp.h"
#include "bogp_handle.h"
#include "bogp_internal.h"
#include "bogp_fit.h"

using namespace bogp;

static std::string g_create_error;

extern "C" int bogp_abi_version(void) { return BOGP_ABI_VERSION; }

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
  h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  // The second stream carries work that runs BESIDE the main stream's dependent chain (the look-ahead remainder of the
  // two-level factorisation, the optional producer overlap): lowest priority, so that a freed workgroup slot should go to
  // the chain's small kernels first instead of to the next workgroup of the bulk kernel (measured neutral for the two-level
  // factorisation: 18.3 ms per likelihood + gradient at N = 8192 with and without priorities).
  int prio_least = 0, prio_greatest = 0;
  if (hipSetDevice(device) != hipSuccess || hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess ||
      hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio_greatest) != hipSuccess ||
      hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, prio_least) != hipSuccess ||
      hipMalloc((void**)&h->dscal, 64 * sizeof(double)) != hipSuccess) {
    g_create_error = "stream creation failed";
    delete h;
    return BOGP_ERR_HIP;
  }
  if (hipHostMalloc((void**)&h->hfit, 4096 * sizeof(double), hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&h->hfit_dev, h->hfit, 0) != hipSuccess) {
    g_create_error = "pinned host buffer allocation failed";
    delete h;
    return BOGP_ERR_HIP;
  }
  memset(h->hfit, 0, 4096 * sizeof(double));
  if (hipMalloc((void**)&h->dfin_ticket, sizeof(unsigned int)) != hipSuccess || hipMemset(h->dfin_ticket, 0, sizeof(unsigned int)) != hipSuccess) {
    g_create_error = "device allocation failed";
    delete h;
    return BOGP_ERR_HIP;
  }
  // the factorisation's info word lives in the same block as its scalars (doubles 62-63): ONE read-back fetches both
  h->dinfo = reinterpret_cast<int*>(h->dscal + 62);
  if (const char* e = getenv("BOGP_CHOL_RESERVE_CU")) {
    // experiment (tools/ab/ab_big_chol_cumask.sh): the look-ahead update of the two-level factorisation on a stream that may
    // not use the last n CUs (mask bit i -> XCD i % 8, so n / 8 CUs per XCD stay free for the panel chain on the main stream)
    const int n = atoi(e);
    if (n > 0 && n < h->n_cu) {
      uint32_t mask[16] = {0};
      for (int i = 0; i < h->n_cu - n && i < 512; ++i) mask[i >> 5] |= 1u << (i & 31);
      if (hipExtStreamCreateWithCUMask(&h->stream_upd, (uint32_t)((h->n_cu + 31) / 32), mask) != hipSuccess) {
        g_create_error = "hipExtStreamCreateWithCUMask failed";
        delete h;
        return BOGP_ERR_HIP;
      }
    }
  }
  if (hipEventCreateWithFlags(&h->ev_chol[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_chol[1], hipEventDisableTiming) != hipSuccess) {
    g_create_error = "event creation failed";
    delete h;
    return BOGP_ERR_HIP;
  }
  *out = h;
  return BOGP_OK;
}

// BOGP_TREND_ROWS=0: polynomial bases with p > 32 columns stay on the r02-r04 tile products (the A/B switch of profiles/r05_trend_timing.txt)
static bool trend_rows_enabled() {
  static const bool on = [] {
    const char* e = getenv("BOGP_TREND_ROWS");
    return !(e && atoi(e) == 0);
  }();
  return on;
}
// smallest basis that takes the path (BOGP_TREND_ROWS_MIN, default 33: p <= 32 stays fused into kernel A, profiles/r05_trend_timing.txt)
static int trend_rows_min() {
  static const int v = [] {
    const char* e = getenv("BOGP_TREND_ROWS_MIN");
    return e ? std::max(2, atoi(e)) : 33;
  }();
  return v;
}

static void free_trend(bogp_handle* h) {
  dfree(h->dF); dfree(h->dFt); dfree(h->dQ1); dfree(h->dQ); dfree(h->dWp); dfree(h->dWpT); dfree(h->dSinvP);
  dfree(h->gsplit.scratch); dfree(h->gsplit.tickets); h->gsplit.cap = 0; h->gsplit.max_tiles = 0;
  for (int b = 0; b < 2; ++b) { dfree(h->dA[b]); dfree(h->dAV[b]); dfree(h->dAU[b]); }
  dfree(h->dAw); dfree(h->dAT); dfree(h->dGinv); dfree(h->dSinv); dfree(h->dbetav); dfree(h->dqty); dfree(h->dinfo2);
  dfree(h->dVpx); h->vpx_cap = 0; dfree(h->dAtx); h->atx_cap = 0; h->vx_Ne = h->vx_Nt = 0;
  h->tr_built = -1; h->tr_p = 0; h->ldp = 0; h->trend = BOGP_TREND_CONSTANT; h->p = 1; h->reml_ftf_basis = -1;
}

static void free_train(bogp_handle* h) {
  dfree(h->dX); dfree(h->dy_base); h->dy = nullptr; dfree(h->dR); dfree(h->dV); dfree(h->dU); dfree(h->dT); dfree(h->dRinv); dfree(h->ddinv); dfree(h->dchain_flags); dfree(h->dones); dfree(h->dgemv_scratch);
  dfree(h->dyt_base); dfree(h->dft); dfree(h->drho_base); dfree(h->dtmp); dfree(h->dgamma_base); dfree(h->dw);
  h->dyt = h->drho = h->dgamma = nullptr;
  h->n_t = 1; h->target = 0;
  dfree(h->dtheta); h->dsqrt_theta = nullptr; dfree(h->dXthT); dfree(h->dXnorm); dfree(h->dVp);
  free_trend(h);
  h->committed = false;
  h->cap_ld = h->cap_d = h->cap_nt = 0;
}

extern "C" void bogp_destroy(bogp_handle* h) {
  if (!h) return;
  for (bogp_handle* a : h->aux) bogp_destroy(a);
  h->aux.clear();
  h->aux_gen.clear();
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  comm_release(h);
  point_release(h);
  batch_release(h);
  free_train(h);
  (void)hipStreamSynchronize(h->stream2);
  if (h->stream_upd) (void)hipStreamSynchronize(h->stream_upd);
  dfree(h->dXs_owned); dfree(h->dss_part); dfree(h->dbounds); dfree(h->dsobol); dfree(h->dxform);
  for (int b = 0; b < 2; ++b) { dfree(h->drT[b]); dfree(h->dmu_part[b]); dfree(h->dw_part[b]); }
  dfree(h->dblk_val); dfree(h->dblk_idx); dfree(h->dmu_out); dfree(h->dmse_out); dfree(h->dacq_out);
  dfree(h->dbest_val); dfree(h->dbest_idx); dfree(h->dtopk_val); dfree(h->dtopk_idx); dfree(h->dcounter); h->dinfo = nullptr; dfree(h->dscal); dfree(h->dgrad_partial); dfree(h->dbatch);
  dfree(h->dTt); dfree(h->dCS); dfree(h->duu); dfree(h->dmtrend); dfree(h->dtpart[0]); dfree(h->dtpart[1]);
  for (auto e : h->ev) (void)hipEventDestroy(e);
  for (int i = 0; i < 2; ++i)
    if (h->ev_chol[i]) (void)hipEventDestroy(h->ev_chol[i]);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  if (h->stream_upd) (void)hipStreamDestroy(h->stream_upd);
  if (h->stream_copy) (void)hipStreamDestroy(h->stream_copy);
  if (h->ev_copy) (void)hipEventDestroy(h->ev_copy);
  if (h->hfit) (void)hipHostFree(h->hfit);
  if (h->dfin_ticket) (void)hipFree(h->dfin_ticket);
  delete h;
}

// point the per-target views (and the committed sigma2) at target t
static void select_target(bogp_handle* h, int t) {
  h->target = t;
  h->dy = h->dy_base + (size_t)t * h->N;
  h->dyt = h->dyt_base + (size_t)t * h->N;
  h->drho = h->drho_base + (size_t)t * h->N;
  h->dgamma = h->dgamma_base + (size_t)t * h->Np;
  if (h->committed && t < (int)h->sigma2_t.size()) h->sigma2 = h->sigma2_t[t];
}

extern "C" int bogp_set_train(bogp_handle* h, const double* X, const double* y, int N, int d, int n_targets) {
  if (!h) return BOGP_ERR_INVALID;
  if (!X || !y || N <= 0 || d <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_set_train: X, y must be non-null and N, d > 0");
  if (n_targets < 1 || n_targets > BOGP_MAX_TARGETS) FAIL(h, BOGP_ERR_UNSUPPORTED, "bogp_set_train: n_targets = %d outside [1, %d]", n_targets, BOGP_MAX_TARGETS);
  if (d > BOGP_MAX_DIM) FAIL(h, BOGP_ERR_UNSUPPORTED, "bogp_set_train: d = %d > %d: the sweep producer keeps a 64 x d candidate tile in the CU's 160 KB of LDS", d, BOGP_MAX_DIM);
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->dX && d != h->d && (h->dXs || h->hXs_lazy)) {
    // candidates were uploaded / bound / generated as M x (old d): rows of another width are not candidates of this model.  A lazy upload
    // in flight would otherwise be finished later with the NEW d -- (M - lazy_done) * d_new doubles read from a host buffer of M * d_old
    //  -- so it is dropped and the candidate set forgotten: the next sweep says "no candidates" until new ones arrive.
    if (h->hXs_lazy) {
      HIPCHK(h, hipStreamSynchronize(h->stream_copy));
      h->hXs_lazy = nullptr;
    }
    h->dXs = nullptr;
    h->M = 0;
    h->last_q = h->last_topk_q = h->last_topk_k = 0;
  }
  // leading dimension: N rounded up to 64; to 128 from 6144 on, where the inverse and R^-1 work on 128 x 128 tiles
  const int ld_need = N > 6080 ? ((N + 127) / 128) * 128 : ((N + 63) / 64) * 64;
  const bool fits = h->dX && h->dtheta && ld_need <= h->cap_ld && d <= h->cap_d && n_targets <= h->cap_nt;
  if (fits) {
    free_trend(h);  // N x p buffers of a polynomial basis: rebuilt on demand
    h->committed = false;
  } else {
    free_train(h);  // also zeroes cap_*: they are set again only after EVERY allocation below has succeeded, so a failure
                    // half way (HIPCHK returns) can never leave a "fits" state with null buffers behind
    // grow in steps of 256 rows once the set is larger than a block, so that a BO loop reallocates every 256 tell()s
    const int new_cap_ld = ld_need <= 256 ? ld_need : ((ld_need + 255) / 256) * 256;
    const size_t cl = (size_t)new_cap_ld, NNc = cl * cl, ntc = (size_t)n_targets;
    HIPCHK(h, hipMalloc((void**)&h->dX, cl * d * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dy_base, ntc * cl * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dR, NNc * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->ddinv, cl * 64 * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dV, NNc * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dU, NNc * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dT, NNc * sizeof(double)));
    {
      std::vector<double> ones(cl, 1.0);
      HIPCHK(h, hipMalloc((void**)&h->dones, cl * sizeof(double)));
      HIPCHK(h, hipMalloc((void**)&h->dgemv_scratch, gemv2_scratch_doubles((int)cl) * sizeof(double)));
      HIPCHK(h, hipMemcpy(h->dones, ones.data(), cl * sizeof(double), hipMemcpyHostToDevice));
    }
    HIPCHK(h, hipMalloc((void**)&h->dyt_base, ntc * cl * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dft, cl * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->drho_base, ntc * cl * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dtmp, cl * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dgamma_base, ntc * cl * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dw, cl * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dtheta, 2 * (d + 1) * sizeof(double)));  // [theta (d + 1) | sqrt_theta (d + 1)]: one upload
    h->dsqrt_theta = h->dtheta + (d + 1);
    h->cap_ld = new_cap_ld;
    h->cap_d = d;
    h->cap_nt = n_targets;
  }
  h->N = N;
  h->d = d;
  h->n_t = n_targets;
  h->target = 0;
  h->Np = ((N + 31) / 32) * 32;
  h->ldr = ld_need;
  const size_t NN = (size_t)h->ldr * h->ldr;
  const size_t nt = (size_t)n_targets;
  HIPCHK(h, launch_pad_identity(h->dR, N, h->ldr, h->stream));
  // V = L^-1 and U = V^T keep exact zeros in their other triangle (set once here; kernels_chol.hip never writes there)
  HIPCHK(h, hipMemsetAsync(h->dV, 0, NN * sizeof(double), h->stream));
  HIPCHK(h, hipMemsetAsync(h->dU, 0, NN * sizeof(double), h->stream));
  select_target(h, 0);
  HIPCHK(h, hipMemcpyAsync(h->dX, X, (size_t)N * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  std::vector<double> ycols(nt * N);  // y arrives (N, n_targets) row-major; one contiguous column per target here
  for (int i = 0; i < N; ++i)
    for (int t = 0; t < n_targets; ++t) ycols[(size_t)t * N + i] = y[(size_t)i * n_targets + t];
  HIPCHK(h, hipMemcpyAsync(h->dy_base, ycols.data(), nt * N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->h_X.assign(X, X + (size_t)N * d);  // (3 MB at C5: what a helper handle of bogp_nll_batch is fed from)
  h->h_y.assign(y, y + (size_t)N * n_targets);
  ++h->train_gen;
  return BOGP_OK;
}

extern "C" int bogp_select_target(bogp_handle* h, int target) {
  if (!h) return BOGP_ERR_INVALID;
  if (!h->dX) FAIL(h, BOGP_ERR_INVALID, "bogp_select_target: no training set");
  if (target < 0 || target >= h->n_t) FAIL(h, BOGP_ERR_INVALID, "bogp_select_target: target %d outside [0, %d)", target, h->n_t);
  select_target(h, target);
  return BOGP_OK;
}

// ------------------------------------------------------------------------------------------------------
// factorise at `par` (shared by bogp_nll and bogp_commit)
// ------------------------------------------------------------------------------------------------------
// (FitOut / FitPending: bogp_fit.h)
int bogp::trend_size(int trend, int d) {
  return trend == BOGP_TREND_CONSTANT ? 1 : trend == BOGP_TREND_LINEAR ? d + 1 : (d + 1) * (d + 2) / 2;
}
extern "C" int bogp_trend_size(int trend, int d) {
  if (trend < BOGP_TREND_CONSTANT || trend > BOGP_TREND_QUADRATIC || d <= 0) return BOGP_ERR_INVALID;
  return trend_size(trend, d);
}

extern "C" int bogp_set_trend_beta(bogp_handle* h, const double* beta, int p) {
  if (!h) return BOGP_ERR_INVALID;
  if (!beta || p <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_set_trend_beta: beta must be non-null and p > 0");
  h->h_beta_fixed.assign(beta, beta + p);
  return BOGP_OK;
}

// buffers of the p > 1 path, (re)allocated when p changes; F is rebuilt when the basis id changes
// split-K scratch of k_gemm64: tiles x slices <= 512 partial tiles of 64 x 64 (16 MB), 128 zeroed ticket words
static int ensure_gsplit(bogp_handle* h) {
  if (h->gsplit.scratch) return BOGP_OK;
  h->gsplit.max_tiles = 128;
  h->gsplit.cap = (size_t)512 * 64 * 64;
  HIPCHK(h, hipMalloc((void**)&h->gsplit.scratch, h->gsplit.cap * sizeof(double)));
  HIPCHK(h, hipMalloc((void**)&h->gsplit.tickets, 128 * sizeof(unsigned int)));
  HIPCHK(h, hipMemsetAsync(h->gsplit.tickets, 0, 128 * sizeof(unsigned int), h->stream));
  return BOGP_OK;
}

static int ensure_trend(bogp_handle* h, int trend) {
  const int N = h->N, d = h->d, Np = h->Np;
  const int p = trend_size(trend, d);
  if (p > 1024) FAIL(h, BOGP_ERR_UNSUPPORTED, "trend with p = %d basis functions (> 1024)", p);
  if (p > N) FAIL(h, BOGP_ERR_INVALID, "trend with p = %d basis functions needs at least as many training points (N = %d)", p, N);
  hipStream_t st = h->stream;
  if (h->tr_p != p) {
    const bool was_committed = h->committed;
    const int tr0 = h->trend, p0 = h->p;
    free_trend(h);
    h->trend = tr0; h->p = p0;
    if (was_committed && p0 > 1) h->committed = false;  // the committed W / beta lived in the freed buffers
    const int ldp = ((p + 63) / 64) * 64;
    const size_t np_ = (size_t)N * p, pp = (size_t)ldp * ldp;
    HIPCHK(h, hipMalloc((void**)&h->dF, np_ * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dFt, np_ * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dQ1, np_ * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dQ, np_ * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dWp, (size_t)Np * p * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dWpT, (size_t)Np * ((p + 127) / 128 * 128) * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dSinvP, (size_t)((p + 127) / 128 * 128) * ((p + 127) / 128 * 128) * sizeof(double)));
    for (int b = 0; b < 2; ++b) {
      HIPCHK(h, hipMalloc((void**)&h->dA[b], pp * sizeof(double)));
      HIPCHK(h, hipMalloc((void**)&h->dAV[b], pp * sizeof(double)));
      HIPCHK(h, hipMalloc((void**)&h->dAU[b], pp * sizeof(double)));
      HIPCHK(h, hipMemsetAsync(h->dAV[b], 0, pp * sizeof(double), st));  // launch_tri_inverse keeps the other triangle zero
      HIPCHK(h, hipMemsetAsync(h->dAU[b], 0, pp * sizeof(double), st));
    }
    HIPCHK(h, hipMalloc((void**)&h->dAw, (size_t)ldp * 64 * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dAT, pp * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dGinv, (size_t)p * p * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dSinv, (size_t)p * p * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dbetav, p * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dqty, p * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&h->dinfo2, 2 * sizeof(int)));
    { const int eg = ensure_gsplit(h); if (eg) return eg; }
    HIPCHK(h, hipMemsetAsync(h->dinfo2, 0, 2 * sizeof(int), st));
    h->tr_p = p;
    h->ldp = ldp;
    h->tr_built = -1;
  }
  if (h->tr_built != trend) {
    HIPCHK(h, launch_trend_train(trend, h->dX, N, d, h->dF, st));
    h->tr_built = trend;
  }
  return BOGP_OK;
}

// Universal / simple kriging with a p > 1 polynomial basis (gpr.py:799-808 with a matrix F).  On entry V = L^-1, U = L^-T
// are current and Yt = V y is queued.  Everything is queued on the handle's stream; nothing is read back here.
//   Ft = V F;  economic QR of Ft by CholeskyQR2 (two passes of: A = Ft^T Ft, chol, Q = Ft R^-1 -- the second pass restores
//   the orthogonality the first loses to cond(Ft)^2; both small factorisations run through kernels_chol.hip);
//   G = R2 R1 (positive diagonal: LAPACK's Householder QR differs by row signs, which no consumer can see),
//   rho = Yt - Q Q^T Yt,  beta = G^-1 Q^T Yt,  (Ft^T Ft)^-1 = G^-1 G^-T for the variance term u^T u.
static int trend_solve(bogp_handle* h, int trend, int estimate_trend) {
  int e = ensure_trend(h, trend);
  if (e) return e;
  const int N = h->N, p = h->tr_p, ldp = h->ldp, ldr = h->ldr;
  hipStream_t st = h->stream;
  const double one = 1.0, zero = 0.0, mone = -1.0;
  HIPCHK(h, launch_gemm(0, 0, N, p, N, one, h->dV, ldr, h->dF, N, zero, h->dFt, N, st, 1, &h->gsplit));  // V lower, zero upper triangle
  HIPCHK(h, hipMemcpyAsync(h->drho, h->dyt, N * sizeof(double), hipMemcpyDeviceToDevice, st));
  if (!estimate_trend) {
    if ((int)h->h_beta_fixed.size() != p) FAIL(h, BOGP_ERR_INVALID, "trend with p = %d fixed coefficients: call bogp_set_trend_beta first (have %d)", p, (int)h->h_beta_fixed.size());
    HIPCHK(h, hipMemcpyAsync(h->dbetav, h->h_beta_fixed.data(), p * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(h, launch_gemm(0, 0, N, 1, p, mone, h->dFt, N, h->dbetav, p, one, h->drho, N, st, 0, &h->gsplit));  // :808
    return BOGP_OK;
  }
  const double* src = h->dFt;
  for (int pass = 0; pass < 2; ++pass) {
    double* dst = pass == 0 ? h->dQ1 : h->dQ;
    HIPCHK(h, launch_gemm(1, 0, p, p, N, one, src, N, src, N, zero, h->dA[pass], ldp, st, 0, &h->gsplit));
    HIPCHK(h, launch_pad_identity(h->dA[pass], p, ldp, st));
    HIPCHK(h, launch_chol_lower(h->dA[pass], ldp, h->dAw, h->dinfo2 + pass, st, nullptr, nullptr, nullptr, p));
    HIPCHK(h, launch_tri_inverse(h->dA[pass], h->dAw, h->dAV[pass], h->dAU[pass], h->dAT, ldp, st));
    HIPCHK(h, launch_gemm(0, 0, N, p, p, one, src, N, h->dAU[pass], ldp, zero, dst, N, st, 0, &h->gsplit));
    src = dst;
  }
  HIPCHK(h, launch_gemm(0, 0, p, p, p, one, h->dAU[0], ldp, h->dAU[1], ldp, zero, h->dGinv, p, st, 0, &h->gsplit));
  HIPCHK(h, launch_gemm(1, 0, p, 1, N, one, h->dQ, N, h->dyt, N, zero, h->dqty, p, st, 0, &h->gsplit));
  HIPCHK(h, launch_gemm(0, 0, N, 1, p, mone, h->dQ, N, h->dqty, p, one, h->drho, N, st, 0, &h->gsplit));  // :806
  HIPCHK(h, launch_gemm(0, 0, p, 1, p, one, h->dGinv, p, h->dqty, p, zero, h->dbetav, p, st, 0, &h->gsplit));  // :785-787
  HIPCHK(h, launch_gemm(0, 1, p, p, p, one, h->dGinv, p, h->dGinv, p, zero, h->dSinv, p, st, 0, &h->gsplit));
  return BOGP_OK;
}

// what the host half of a factorisation (factorize_finish) needs once info / the device scalars have been read back
int bogp::fit_wait_on(bogp_handle* h, const void* flag_word, unsigned long long seq) {
  volatile const unsigned long long* flag = reinterpret_cast<volatile const unsigned long long*>(flag_word);
  bool seen = false;
  for (int spin = 0; spin < 400000; ++spin) {
    if (*flag == seq) { seen = true; break; }
    __builtin_ia32_pause();
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  if (!seen) HIPCHK(h, hipStreamSynchronize(h->stream));
  return BOGP_OK;
}
static int fit_wait(bogp_handle* h, unsigned long long seq) { return fit_wait_on(h, h->hfit + 3000, seq); }
// The 64 scalars of the evaluation (and nS gradient sums from dS, or none) back on the host: one gather launch into the mapped
// pinned block + a polled sequence word instead of two copy commands into pageable memory + a stream synchronisation (the
// host's API calls, not the GPU, bound an evaluation at the sizes of an ordinary BO run: profiles/r03_bo_loop.txt).
// BOGP_FIT_POLL=0 restores the copies.  Bounded: after ~2 ms of polling the ordinary synchronisation takes over.
static int fit_readback(bogp_handle* h, const double* dS, int nS, double* blk /* 64 */, double* S_out) {
  hipStream_t st = h->stream;
  static const bool poll = [] { const char* e_ = getenv("BOGP_FIT_POLL"); return !(e_ && atoi(e_) == 0); }();
  if (!poll || nS > 512) {
    if (nS > 0) HIPCHK(h, hipMemcpyAsync(S_out, dS, (size_t)nS * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipMemcpyAsync(blk, h->dscal, 64 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    return BOGP_OK;
  }
  const unsigned long long seq = ++h->fit_seq;
  HIPCHK(h, launch_fit_gather(h->dscal, dS, nS, h->hfit_dev + 2048, h->hfit_dev + 2112,
                              reinterpret_cast<unsigned long long*>(h->hfit_dev + 3000), seq, st));
  const int ew = fit_wait(h, seq);
  if (ew) return ew;
  memcpy(blk, h->hfit + 2048, 64 * sizeof(double));
  if (nS > 0) memcpy(S_out, h->hfit + 2112, (size_t)nS * sizeof(double));
  return BOGP_OK;
}

// pend == nullptr: queue the device work, read info + scalars back, finish (ONE host synchronisation).
// pend != nullptr: queue only -- the caller appends its own device work (the likelihood gradient), reads everything back in ONE
// synchronisation and calls factorize_finish itself.
extern "C" int bogp_nll_path(int N, int d, int trend, int n_targets) {
  if (N <= 0 || d <= 0 || trend != BOGP_TREND_CONSTANT || n_targets != 1) return BOGP_NLL_PATH_GENERAL;
  if (getenv("BOGP_NLL_FUSED") && atoi(getenv("BOGP_NLL_FUSED")) == 0) return BOGP_NLL_PATH_GENERAL;
  if (nll_small_fits(N, d)) return BOGP_NLL_PATH_ONE_LAUNCH;
  // 157 <= N <= 3072 (BOGP_NLL_ELIM_MAX; r05: 2048 -> 3072 after the step lost a third of its time -- llf + gradient 1.73 -> 0.95 ms at N = 2112,
  // 2.78 -> 2.28 at 3072, a slot of a batch of ten 1.61 -> 0.93 ms; it loses from ~3500 on: profiles/r05_elim_chain.txt): factor + inverse + solves as one
  // elimination at 64-block granularity (kernels_chol.hip: k_elim_step), one launch a block column; BOGP_NLL_ELIM=0 keeps the
  // Cholesky / recursive-doubling / U U^T kernels
  static const int elim_max = [] { const char* e_ = getenv("BOGP_NLL_ELIM_MAX"); return e_ ? atoi(e_) : 3072; }();
  const int ld = ((N + 63) / 64) * 64;
  if (N <= elim_max && N <= 6080 && ld >= 192 && !(getenv("BOGP_NLL_ELIM") && atoi(getenv("BOGP_NLL_ELIM")) == 0)) return BOGP_NLL_PATH_ELIM;
  return BOGP_NLL_PATH_GENERAL;
}

// fz != nullptr: the caller only wants the likelihood (and its gradient sums), not the factor buffers -- a training set of at most
// 128 points with the constant basis and one target is then evaluated by ONE launch (kernels_nllsmall.hip), `done` says so.
struct FusedNll {
  bool want_grad = false;
  bool done = false;
  bool mid = false;  // 157 <= N <= 3072: k_build_R + k_elim_* left R^-1, gamma, the scalars and the gradient weights; the caller's tail follows
  double S[64 + 3];
};
static int factorize(bogp_handle* h, int kernel, int mode, const double* par, int n_par, double noise_var, int trend,
                     int estimate_trend, double beta, bool want_gamma, FitOut* o, std::vector<double>* theta_out,
                     bool reject_positive = true, FitPending* pend = nullptr, FusedNll* fz = nullptr) {
  if (!h->dX) FAIL(h, BOGP_ERR_INVALID, "no training set: call bogp_set_train first");
  if (kernel < 0 || kernel > BOGP_KERNEL_MATERN_NU) FAIL(h, BOGP_ERR_INVALID, "unknown kernel id %d", kernel);
  if (mode < 0 || mode > 2) FAIL(h, BOGP_ERR_INVALID, "unknown estimation mode %d", mode);
  if (trend < BOGP_TREND_CONSTANT || trend > BOGP_TREND_QUADRATIC) FAIL(h, BOGP_ERR_INVALID, "unknown trend id %d", trend);
  const int ptrend = trend_size(trend, h->d);
  const int N = h->N, d = h->d, ldr = h->ldr;
  int n_theta = n_par - (mode == BOGP_MODE_NOISELESS ? 0 : 1);
  double pexp = 0.0;
  if (kernel == BOGP_KERNEL_MATERN_NU) {  // theta = [theta_1 .. theta_d, nu], or [theta, nu]: the order travels where generalized_exponential's exponent does
    if (n_theta != d + 1 && n_theta != 2) FAIL(h, BOGP_ERR_INVALID, "general-nu matern: len(theta) = %d must be 2 or d + 1 = %d (the last entry is nu)", n_theta, d + 1);
    pexp = par[n_theta - 1];
    if (!(pexp > 0) || !std::isfinite(pexp) || pexp > 60.0) FAIL(h, BOGP_ERR_INVALID, "general-nu matern: nu = %g must be in (0, 60]", pexp);
    n_theta -= 1;
  }
  if (kernel == BOGP_KERNEL_GENEXP) {  // theta = [theta_1 .. theta_d, p], or [theta, p] (kernel.py:369-373)
    if (n_theta != d + 1 && n_theta != 2) FAIL(h, BOGP_ERR_INVALID, "generalized_exponential: len(theta) = %d must be 2 or d + 1 = %d", n_theta, d + 1);
    pexp = par[n_theta - 1];
    if (!(pexp > 0) || !std::isfinite(pexp)) FAIL(h, BOGP_ERR_INVALID, "generalized_exponential: exponent p = %g must be finite and > 0", pexp);
    n_theta -= 1;
  }
  if (n_theta != d && n_theta != 1) FAIL(h, BOGP_ERR_INVALID, "len(theta) = %d must be 1 or d = %d", n_theta, d);
  h->h_theta.resize(2 * (size_t)(d + 1));  // handle-owned: the asynchronous upload below outlives this scope
  double* th = h->h_theta.data();          // [theta (d + 1) | sqrt_theta (d + 1)], uploaded in one copy
  double* sth = th + (d + 1);
  for (int k = 0; k < d; ++k) {
    th[k] = par[n_theta == 1 ? 0 : k];
    if (!(th[k] > 0) || !std::isfinite(th[k])) FAIL(h, BOGP_ERR_INVALID, "theta[%d] = %g must be finite and > 0", k, th[k]);
    // coordinates are pre-scaled so that the producer forms (a - b)^2 (radial kernels), |a - b| (absolute_exponential,
    // cubic) or |a - b|^p (generalized_exponential: theta_k^(1/p))
    sth[k] = (kernel == BOGP_KERNEL_ABSEXP || kernel == BOGP_KERNEL_CUBIC) ? th[k]
             : kernel == BOGP_KERNEL_GENEXP ? std::pow(th[k], 1.0 / pexp) : std::sqrt(th[k]);
  }
  th[d] = sth[d] = pexp;  // entry d of both device arrays: the exponent (read by the generalized_exponential kernels only)
  if (theta_out) theta_out->assign(th, th + d);
  hipStream_t st = h->stream;
  HIPCHK(h, hipSetDevice(h->device));
  const int path = fz ? bogp_nll_path(N, d, trend, h->n_t) : BOGP_NLL_PATH_GENERAL;
  // 157 <= N <= 3072 (BOGP_NLL_ELIM_MAX; slower than the kernels it replaces from ~3500 on): factor + inverse + solves as one elimination at 64-block granularity (kernels_chol.hip: k_elim_step), one
  // launch a block column; BOGP_NLL_ELIM=0 keeps the Cholesky / recursive-doubling / U U^T kernels
  const bool elim = path == BOGP_NLL_PATH_ELIM && (!fz->want_grad || pend);
  const bool mid = elim;
  if (path == BOGP_NLL_PATH_ONE_LAUNCH) {
    NllSmallArgs na;
    na.X = h->dX; na.y = h->dy_base; na.N = N; na.d = d;
    for (int k = 0; k < d; ++k) na.theta[k] = th[k];
    na.pexp = pexp;
    FitPending fp;
    fp.mode = mode; fp.estimate_trend = estimate_trend; fp.ptrend = 1; fp.n_t = 1; fp.N = N;
    fp.beta = beta; fp.alpha = 0; fp.sigma2_par = 0; fp.noise_var = noise_var; fp.s2t = 0;
    if (mode == BOGP_MODE_NOISELESS) {
      h->R_div = false; h->R_a = 1.0; h->R_b = 1.0; h->R_diag = 1.0;
    } else if (mode == BOGP_MODE_NOISE_ESTIM) {
      fp.alpha = par[n_par - 1];
      h->R_div = false; h->R_a = fp.alpha; h->R_b = 1.0; h->R_diag = fp.alpha * 1.0 + (1 - fp.alpha) * 1.0;
    } else {
      fp.sigma2_par = par[n_par - 1];
      fp.s2t = fp.sigma2_par + noise_var;
      h->R_div = true; h->R_a = fp.sigma2_par; h->R_b = fp.s2t; h->R_diag = (fp.sigma2_par * 1.0 + noise_var * 1.0) / fp.s2t;
    }
    na.a = h->R_a; na.b = h->R_b; na.diag = h->R_diag; na.div = h->R_div ? 1 : 0;
    na.estimate_trend = estimate_trend; na.mode = mode; na.beta = beta; na.s2t_host = fp.s2t;
    na.out_scal = h->hfit_dev + 2048; na.out_S = h->hfit_dev + 2112;
    na.flag = reinterpret_cast<unsigned long long*>(h->hfit_dev + 3000);
    na.seq = ++h->fit_seq;
    HIPCHK(h, launch_nll_small(kernel, fz->want_grad, na, st));
    const int ew = fit_wait(h, na.seq);
    if (ew) return ew;
    double blk[64];
    memcpy(blk, h->hfit + 2048, sizeof(blk));
    if (fz->want_grad) memcpy(fz->S, h->hfit + 2112, (size_t)(d + 3) * sizeof(double));
    fz->done = true;
    int info = 0;
    memcpy(&info, blk + 62, sizeof(info));
    const int info2[2] = {0, 0};
    return factorize_finish(h, fp, info, blk, info2, reject_positive, o);
  }
  h->dsqrt_theta = h->dtheta + (d + 1);  // (the block holds 2 (cap_d + 1) doubles; d may be below the capacity)
  // (through the pinned staging block when it fits: a copy from pageable memory is staged by the runtime, synchronously)
  const double* th_src = th;
  if (2 * (size_t)(d + 1) <= 2048) {
    memcpy(h->hfit, th, 2 * (size_t)(d + 1) * sizeof(double));
    th_src = h->hfit;
  }
  HIPCHK(h, hipMemcpyAsync(h->dtheta, th_src, 2 * (size_t)(d + 1) * sizeof(double), hipMemcpyHostToDevice, st));

  // The identity padding is re-established for EVERY factorisation: a factorisation that broke down (pivots of rounding
  // size -> overflowing inverses -> inf * 0) leaves NaN in the padding rows of the in-place factor, and R is only rebuilt
  // inside its N x N block -- without this, one failed likelihood evaluation made every later one on the handle fail too
  // (found with the near-singular noiseless cubic tables of G25).
  if (!mid) HIPCHK(h, launch_pad_identity(h->dR, N, ldr, st));  // (k_elim_init pads)
  // correlation matrix with the per-mode normalisation (gpr.py:931-969)
  double s2t = 0, alpha = 0, sigma2_par = 0;
  if (mode == BOGP_MODE_NOISELESS) {
    h->R_div = false; h->R_a = 1.0; h->R_b = 1.0; h->R_diag = 1.0;
    HIPCHK(h, launch_build_R(kernel, h->dX, N, d, h->dtheta, 1.0, 1.0, h->dR, ldr, st));
  } else if (mode == BOGP_MODE_NOISE_ESTIM) {
    alpha = par[n_par - 1];
    h->R_div = false; h->R_a = alpha; h->R_b = 1.0; h->R_diag = alpha * 1.0 + (1 - alpha) * 1.0;
    HIPCHK(h, launch_build_R(kernel, h->dX, N, d, h->dtheta, alpha, alpha * 1.0 + (1 - alpha) * 1.0, h->dR, ldr, st));
  } else {
    sigma2_par = par[n_par - 1];
    s2t = sigma2_par + noise_var;
    h->R_div = true; h->R_a = sigma2_par; h->R_b = s2t; h->R_diag = (sigma2_par * 1.0 + noise_var * 1.0) / s2t;
    HIPCHK(h, launch_build_R_div(kernel, h->dX, N, d, h->dtheta, sigma2_par, s2t, (sigma2_par * 1.0 + noise_var * 1.0) / s2t,
                                 h->dR, ldr, st));
  }
  // The whole evaluation is queued without a host round trip and read back once:
  //   L = chol(R) (gpr.py:795)                      kernels_chol.hip
  //   V = L^-1, U = L^-T                            every triangular solve of :799-808 / :787-788 / :997 becomes a product
  //   Yt = V y (:799), Ft = V 1 (:803)              one pass over V
  //   rho (:806 / :808), |Ft|, Ft.Yt, rho.rho       k_fit_rho
  //   gamma = U rho (:788 / :996)
  const int n_t = h->n_t;
  if (elim) {
    if (!h->dRinv) HIPCHK(h, hipMalloc((void**)&h->dRinv, (size_t)UUT_PARTS * h->cap_ld * h->cap_ld * sizeof(double)));
    // scratch behind the first of the UUT_PARTS slices of dRinv (the result goes into that slice): two raw panels, block row nb,
    // Yt, Ft, the log-determinant parts
    double* sc0 = h->dRinv + (size_t)ldr * ldr;
    const int lde = ldr + 64;
    ElimArgs ea;
    ea.E = h->dR; ea.ld = ldr; ea.nb = ldr / 64; ea.N = N;
    double* panels = sc0;
    ea.Eb = panels + (size_t)2 * lde * 64;
    ea.yt = ea.Eb + (size_t)64 * ldr;
    ea.ft = ea.yt + ldr;
    ea.logpart = ea.ft + ldr;
    ea.info = h->dinfo;
    HIPCHK(h, launch_elim(ea, h->dy_base, h->ddinv, panels, h->dRinv, ldr, h->dgamma_base, h->dscal, h->dscal + 4 * BOGP_MAX_TARGETS,
                          estimate_trend, mode, beta, s2t, st));
    fz->mid = true;
  } else {
  if (!h->dchain_flags) HIPCHK(h, hipMalloc((void**)&h->dchain_flags, (size_t)2 * (h->cap_ld / 64 + 1) * sizeof(unsigned int)));
  HIPCHK(h, launch_chol_lower(h->dR, ldr, h->ddinv, h->dinfo, st, h->stream_upd ? h->stream_upd : h->stream2, h->ev_chol, h->dT, N, h->dchain_flags));  // dT: free until the inverse
  const bool logdet_in_rho = trend_size(trend, h->d) == 1;  // constant basis: k_fit_rho of target 0 forms sum(log diag L) too (one launch less)
  if (!logdet_in_rho) HIPCHK(h, launch_logdet(h->dR, N, ldr, h->dscal, st));
  HIPCHK(h, launch_tri_inverse(h->dR, h->ddinv, h->dV, h->dU, h->dT, ldr, st));
  if (n_t > 1 && (ptrend != 1 || estimate_trend))
    FAIL(h, BOGP_ERR_UNSUPPORTED, "multi-target y (%d targets) is built for a FIXED constant trend only: with estimated coefficients the reference raises at gpr.py:787 (beta gets one row per target)", n_t);
  if (ptrend == 1) {
    for (int t = 0; t < n_t; ++t) {  // scal[4 t + 1..3] = |Ft|, Ft.Yt_t, rho_t.rho_t
      HIPCHK(h, launch_gemv2(h->dV, ldr, N, 1, h->dy_base + (size_t)t * N, h->dones, h->dyt_base + (size_t)t * N, h->dft, h->dgemv_scratch, st));
      // (one target and the gradient queued behind: k_grad_coef's two weights come from this kernel too)
      HIPCHK(h, launch_fit_rho(h->dyt_base + (size_t)t * N, h->dft, N, estimate_trend, beta, h->drho_base + (size_t)t * N, h->dscal + 4 * t, st,
                               t == 0 ? h->dR : nullptr, ldr, (pend && n_t == 1) ? h->dscal + 4 * BOGP_MAX_TARGETS : nullptr, mode, s2t));
    }
  } else {
    HIPCHK(h, launch_gemv2(h->dV, ldr, N, 1, h->dy, nullptr, h->dyt, nullptr, h->dgemv_scratch, st));
    int et = trend_solve(h, trend, estimate_trend);
    if (et) return et;
    HIPCHK(h, launch_sumsq(h->drho, N, h->dscal + 3, st));
  }
  if (want_gamma) {
    // (the zero padding matters to the sweeps after a commit and to the several-target sum of squares; a likelihood evaluation of
    // one target reads gamma[0 .. N) only)
    if (!(fz && n_t == 1)) HIPCHK(h, hipMemsetAsync(h->dgamma_base, 0, (size_t)n_t * h->Np * sizeof(double), st));
    for (int t = 0; t < n_t; ++t)
      HIPCHK(h, launch_gemv2(h->dU, ldr, N, 2, h->drho_base + (size_t)t * N, nullptr, h->dgamma_base + (size_t)t * h->Np, nullptr, h->dgemv_scratch, st));
  }
  }  // !mid
  FitPending fp;
  fp.mode = mode; fp.estimate_trend = estimate_trend; fp.ptrend = ptrend; fp.n_t = n_t; fp.N = N;
  fp.beta = beta; fp.alpha = alpha; fp.sigma2_par = sigma2_par; fp.noise_var = noise_var; fp.s2t = s2t;
  if (pend) {
    *pend = fp;
    return BOGP_OK;
  }
  double blk[64];  // [0 .. 4 n_t): sum(log diag L), |Ft|, Ft.Yt, rho.rho (the last three per target); [62]: the info word
  const double* sc = blk;
  int info2[2] = {0, 0};
  if (ptrend > 1 && estimate_trend) {
    HIPCHK(h, hipMemcpyAsync(blk, h->dscal, sizeof(blk), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipMemcpyAsync(info2, h->dinfo2, sizeof(info2), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
  } else {
    const int er = fit_readback(h, nullptr, 0, blk, nullptr);
    if (er) return er;
  }
  int info = 0;
  memcpy(&info, blk + 62, sizeof(info));
  return factorize_finish(h, fp, (int)info, sc, info2, reject_positive, o);
}

int bogp::factorize_finish(bogp_handle* h, const FitPending& fp, int info, const double* sc, const int* info2,
                           bool reject_positive, FitOut* o) {
  const int mode = fp.mode, estimate_trend = fp.estimate_trend, ptrend = fp.ptrend, n_t = fp.n_t, N = fp.N;
  const double beta = fp.beta, alpha = fp.alpha, sigma2_par = fp.sigma2_par, noise_var = fp.noise_var;
  double s2t = fp.s2t;
  if (info < 0) FAIL(h, BOGP_ERR_HIP, "factorisation: a hand-over between the diagonal chain and the block-column kernels timed out (info = %d)", (int)info);
  if (info != 0) FAIL(h, BOGP_ERR_NOT_POSDEF, "correlation matrix is not positive definite (potrf info = %d)", (int)info);
  if (info2[0] != 0 || info2[1] != 0) FAIL(h, BOGP_ERR_NOT_POSDEF, "trend basis is rank deficient after whitening (Ft^T Ft not positive definite, info = %d / %d)", (int)info2[0], (int)info2[1]);

  const double logdet = sc[0], rho_ss = sc[3];
  double ftyt = 0, ftft = 0, G = 0, beta_eff = beta;
  if (estimate_trend && ptrend == 1) {
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
    const int k = estimate_trend ? ptrend : 0;  // rank(Q Q^T) (:941), full column rank assumed
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
  o->logdet = logdet; o->rho_ss = rho_ss;
  o->llf = llf; o->sigma2 = sigma2; o->noise_var = nv; o->s2t = s2t; o->G = G; o->beta = beta_eff; o->ftyt = ftyt; o->ftft = ftft;
  o->sigma2_t[0] = sigma2; o->s2t_t[0] = s2t; o->nv_t[0] = nv;
  bool positive = llf > 0;
  for (int t = 1; t < n_t; ++t) {  // the same three formulas per target; the reference sums them (:1040) and rejects
    const double rss = sc[4 * t + 3];  // when ANY target's value is positive (:981)
    double l_t, s_t, st_t, nv_t;
    if (mode == BOGP_MODE_NOISELESS) {
      s_t = rss / N; nv_t = 0; st_t = s_t;
      l_t = -0.5 * (N * std::log(TWO_PI * s_t) + 2.0 * logdet + N);
    } else if (mode == BOGP_MODE_NOISE_ESTIM) {
      st_t = rss / N; s_t = alpha * st_t; nv_t = (1 - alpha) * st_t;
      l_t = -0.5 * (N * std::log(TWO_PI * st_t) + 2.0 * logdet + N);
    } else {
      s_t = sigma2_par; nv_t = noise_var; st_t = s2t;
      l_t = -0.5 * (N * std::log(TWO_PI * st_t) + 2.0 * logdet + rss / st_t);
    }
    if (!std::isfinite(l_t)) FAIL(h, BOGP_ERR_NOT_POSDEF, "log-likelihood of target %d is not finite (%g)", t, l_t);
    o->sigma2_t[t] = s_t; o->s2t_t[t] = st_t; o->nv_t[t] = nv_t;
    o->llf += l_t;
    positive = positive || l_t > 0;
  }
  if (positive && reject_positive) FAIL(h, BOGP_ERR_LLF_POSITIVE, "log-likelihood %g > 0 is rejected by the reference (gpr.py:981-982)", o->llf);

  return BOGP_OK;
}

// the likelihood gradient from the d + 1 contractions, trace(R^-1) and gamma.gamma (gpr.py:1001-1038)
void bogp::nll_gradient_from_sums(int mode, bool iso, int d, const double* par, int n_par, int n_t, const double* S, double s2t,
                                  double* grad) {
  const double tr = n_t * S[d + 1], gg = S[d + 2];
  if (iso) {
    grad[0] = mode == BOGP_MODE_NOISE_ESTIM ? par[n_par - 1] * S[0] : S[0];
    if (mode == BOGP_MODE_NOISE_ESTIM) grad[1] = S[d];
    if (mode == BOGP_MODE_NOISY) grad[1] = S[1];
  } else if (mode == BOGP_MODE_NOISELESS) {
    for (int k = 0; k < d; ++k) grad[k] = S[k];
  } else if (mode == BOGP_MODE_NOISE_ESTIM) {
    const double alpha = par[n_par - 1];
    for (int k = 0; k < d; ++k) grad[k] = alpha * S[k];
    grad[d] = S[d];
  } else {
    for (int k = 0; k < d; ++k) grad[k] = S[k];
    grad[d] = -0.5 * (tr / s2t - gg / (s2t * s2t)) + S[d] / s2t;
  }
}

extern "C" int bogp_nll(bogp_handle* h, int kernel, int mode, const double* par, int n_par, double noise_var, int trend,
                        int estimate_trend, double beta, double* llf, double* grad) {
  if (!h) return BOGP_ERR_INVALID;
  if (!par || !llf || n_par <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_nll: par/llf must be non-null");
  if (grad && (kernel == BOGP_KERNEL_CUBIC || kernel == BOGP_KERNEL_GENEXP || kernel == BOGP_KERNEL_MATERN_NU))
    FAIL(h, BOGP_ERR_UNSUPPORTED, "bogp_nll: the cubic / generalized_exponential correlation has no theta-derivative (the reference's corr_grad_theta leaves it undefined, gpr.py:763-766: its own likelihood gradient raises UnboundLocalError)");
  h->committed = false;  // the factor buffers are about to be overwritten
  FitOut o;
  // With the constant basis the gradient kernels are queued straight behind the factorisation (their only host-dependent
  // inputs, the per-target weights, are formed on the device by k_grad_coef) and info, the likelihood scalars and the d + 1
  // contractions come back in ONE synchronisation: ~60 us less per evaluation than reading the scalars first (the whole
  // evaluation is 0.15 ms at N <= 64).  A failed factorisation then wastes the queued gradient work -- the rare case.
  const bool deferred = grad != nullptr && trend == BOGP_TREND_CONSTANT && !(getenv("BOGP_NLL_TWO_SYNCS") && atoi(getenv("BOGP_NLL_TWO_SYNCS")) != 0);
  FitPending fp;
  FusedNll fz;
  fz.want_grad = grad != nullptr;
  int rc = factorize(h, kernel, mode, par, n_par, noise_var, trend, estimate_trend, beta, grad != nullptr, &o, nullptr, true,
                     deferred ? &fp : nullptr, &fz);
  if (!deferred || fz.done) *llf = o.llf;
  if (rc != BOGP_OK) return rc;
  if (!grad) return BOGP_OK;

  const int N = h->N, d = h->d;
  const int n_theta = n_par - (mode == BOGP_MODE_NOISELESS ? 0 : 1);
  if (fz.done) {
    nll_gradient_from_sums(mode, n_theta != d, d, par, n_par, 1, fz.S, o.s2t, grad);
    return BOGP_OK;
  }
  // Isotropic theta (len 1, d > 1): corr_grad_theta still returns the (N, N, d) per-dimension tensor (gpr.py:745-770) and the
  // loops of :1001-1037 index it BY PARAMETER, so row 0 is the derivative along dimension 0 only and, in the noisy mode,
  // the "sigma2" row is the derivative along dimension 1 (slice 1 of the d + 1 slices).  That is what the reference's MLE
  // is driven by, so it is reproduced here from the same d + 1 contractions.
  const bool iso = n_theta != d;
  hipStream_t st = h->stream;
  // R^-1 = cho_solve(L, I) (:997) via potri on a copy of L
  const int ldr = h->ldr;
  if (!h->dRinv) HIPCHK(h, hipMalloc((void**)&h->dRinv, (size_t)UUT_PARTS * h->cap_ld * h->cap_ld * sizeof(double)));
  int nparts = UUT_PARTS;
  if (fz.mid) nparts = 1;  // (k_elim_finish left R^-1 itself)
  else HIPCHK(h, launch_uut(h->dU, h->dRinv, ldr, st, &nparts));  // R^-1 = L^-T L^-1, lower triangle
  const int nblk = grad_contract_blocks(N);
  int e = ensure(h, &h->dgrad_partial, &h->grad_partial_cap, (size_t)nblk * (d + 1) + (d + 4));
  if (e) return e;
  // Per-target weights of gamma_t gamma_t^T (single target: 1 / sigma2 resp. 1 / sigma2_total).  With several targets the
  // reference sums gamma gamma^T over ALL targets before dividing by each target's variance in the theta rows of the
  // noiseless / noise_estim modes (`_upper`, :999 with :1008-1020), but uses each target's own variance in the alpha row
  // (:1024-1026) and in the noisy mode (:1036); the R^-1 term is counted once per target (`.sum(axis=1)`, :1038).
  const int n_t = h->n_t;
  GradVecs gv;
  gv.v = h->dgamma_base; gv.stride = (size_t)h->Np; gv.n = n_t; gv.c0 = (double)n_t;
  if (deferred) {
    // scal[4 n_t ..]: 16 doubles of weights behind the per-target scalars (dscal holds 64 doubles)
    double* dcoef = h->dscal + 4 * BOGP_MAX_TARGETS;
    if (!fz.mid && n_t > 1) HIPCHK(h, launch_grad_coef(h->dscal, n_t, mode, N, estimate_trend ? 1 : 0, fp.s2t, dcoef, st));  // (one target: k_fit_rho did it)
    gv.dcoef = dcoef;
    for (int t = 0; t < BOGP_MAX_TARGETS; ++t) gv.cA[t] = gv.cB[t] = 0.0;
  } else {
    double inv_sum = 0.0;
    for (int t = 0; t < n_t; ++t) inv_sum += 1.0 / (mode == BOGP_MODE_NOISELESS ? o.sigma2_t[t] : o.s2t_t[t]);
    for (int t = 0; t < n_t; ++t) {
      gv.cB[t] = 1.0 / o.s2t_t[t];
      gv.cA[t] = mode == BOGP_MODE_NOISY ? gv.cB[t] : inv_sum;
    }
  }
  HIPCHK(h, launch_grad_contract(kernel, h->dX, N, d, h->dtheta, gv, nullptr, 0.0, h->dRinv, ldr, nparts, (size_t)ldr * ldr, h->dgrad_partial, nblk, st));
  double* dS = h->dgrad_partial + (size_t)nblk * (d + 1);
  std::vector<double> S(d + 3);
  static const bool fit_poll = [] { const char* e_ = getenv("BOGP_FIT_POLL"); return !(e_ && atoi(e_) == 0); }();
  if (deferred && n_t == 1 && fit_poll && d + 3 <= 512) {
    // the column sums, trace(R^-1) / gamma.gamma and the read-back in ONE launch (k_grad_finish) + the polled sequence word
    const unsigned long long seq = ++h->fit_seq;
    HIPCHK(h, launch_grad_finish(h->dgrad_partial, nblk, d + 1, dS, h->dRinv, ldr, nparts, (size_t)ldr * ldr, N, h->dgamma_base,
                                 mode == BOGP_MODE_NOISY ? 1 : 0, h->dscal, h->hfit_dev + 2048, h->hfit_dev + 2112,
                                 reinterpret_cast<unsigned long long*>(h->hfit_dev + 3000), seq, h->dfin_ticket, st));
    const int ew = fit_wait(h, seq);
    if (ew) return ew;
    double blk[64];
    memcpy(blk, h->hfit + 2048, sizeof(blk));
    memcpy(S.data(), h->hfit + 2112, (size_t)(d + 3) * sizeof(double));
    const int info2[2] = {0, 0};
    int info = 0;
    memcpy(&info, blk + 62, sizeof(info));
    rc = factorize_finish(h, fp, (int)info, blk, info2, true, &o);
    *llf = o.llf;
    if (rc != BOGP_OK) return rc;
    nll_gradient_from_sums(mode, iso, d, par, n_par, n_t, S.data(), o.s2t, grad);
    return BOGP_OK;
  }
  HIPCHK(h, launch_grad_reduce(h->dgrad_partial, nblk, d + 1, dS, st));
  if (mode == BOGP_MODE_NOISY) {
    HIPCHK(h, launch_trace_gg(h->dRinv, ldr, nparts, (size_t)ldr * ldr, N, h->dgamma_base, nullptr, dS + d + 1, st));
    if (n_t > 1) HIPCHK(h, launch_sumsq(h->dgamma_base, n_t * h->Np, dS + d + 2, st));  // sum_t gamma_t . gamma_t (zero padding)
  }
  if (deferred) {
    double blk[64];
    const int info2[2] = {0, 0};
    {
      const int er = fit_readback(h, dS, d + 3, blk, S.data());
      if (er) return er;
    }
    int info = 0;
    memcpy(&info, blk + 62, sizeof(info));
    rc = factorize_finish(h, fp, (int)info, blk, info2, true, &o);
    *llf = o.llf;
    if (rc != BOGP_OK) return rc;
  } else {
    HIPCHK(h, hipMemcpyAsync(S.data(), dS, (d + 3) * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
  }
  nll_gradient_from_sums(mode, iso, d, par, n_par, n_t, S.data(), o.s2t, grad);
  return BOGP_OK;
}

// Restricted likelihood (gpr.py:813-918).  par: noiseless [theta, sigma2]; noisy [theta, sigma2] + the fixed noise_var
// argument; noise_estim [theta, sigma2, noise_var].  The factorisation is the NOISY-mode one (R = (sigma2 R0 + nv I) /
// (sigma2 + nv), :836-839), so the device work is shared with bogp_nll; only the scalar formula and the extra
// (L^-T Q)(L^-T Q)^T term of the gradient differ.  Returns BOGP_ERR_LLF_POSITIVE when exp(llf) > 1 (:868-871) -- with the
// gradient of the finite value filled in, as the reference returns it.
extern "C" int bogp_nll_restricted(bogp_handle* h, int kernel, int mode, const double* par, int n_par, double noise_var,
                                   int trend, int estimate_trend, double beta, double* llf, double* grad) {
  if (!h) return BOGP_ERR_INVALID;
  if (!par || !llf || n_par <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_nll_restricted: par/llf must be non-null");
  if (grad && (kernel == BOGP_KERNEL_CUBIC || kernel == BOGP_KERNEL_GENEXP || kernel == BOGP_KERNEL_MATERN_NU)) FAIL(h, BOGP_ERR_UNSUPPORTED, "bogp_nll_restricted: the cubic / generalized_exponential correlation has no theta-derivative");
  if (mode < 0 || mode > 2) FAIL(h, BOGP_ERR_INVALID, "unknown estimation mode %d", mode);
  if (trend < BOGP_TREND_CONSTANT || trend > BOGP_TREND_QUADRATIC) FAIL(h, BOGP_ERR_INVALID, "unknown trend id %d", trend);
  // several targets: the VALUE as the reference's arithmetic gives it (the scalar terms broadcast over the n_t x n_t matrix rho^T rho and everything
  // summed, gpr.py:861-866); its gradient raises there (a (1, N n_t) by (N, N) product, :875, :896)
  if (h->n_t != 1 && grad) FAIL(h, BOGP_ERR_UNSUPPORTED, "bogp_nll_restricted: no gradient with %d targets (the reference raises ValueError at gpr.py:896)", h->n_t);
  if (h->n_t != 1 && (estimate_trend || trend != BOGP_TREND_CONSTANT))
    FAIL(h, BOGP_ERR_UNSUPPORTED, "bogp_nll_restricted: %d targets need a FIXED constant trend (gpr.py:787)", h->n_t);
  h->committed = false;
  const int n_tail = mode == BOGP_MODE_NOISE_ESTIM ? 2 : 1;
  const int n_theta = n_par - n_tail;
  if (n_theta <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_nll_restricted: %d parameters for mode %d", n_par, mode);
  const double sigma2 = par[n_theta];
  const double nv = mode == BOGP_MODE_NOISELESS ? 0.0 : (mode == BOGP_MODE_NOISY ? noise_var : par[n_theta + 1]);
  if (!(sigma2 > 0) || !(nv >= 0) || !std::isfinite(sigma2) || !std::isfinite(nv)) FAIL(h, BOGP_ERR_INVALID, "bogp_nll_restricted: sigma2 = %g, noise_var = %g", sigma2, nv);
  std::vector<double> p2(par, par + n_theta + 1);  // [theta, sigma2]
  FitOut o;
  *llf = -INFINITY;
  int rc = factorize(h, kernel, BOGP_MODE_NOISY, p2.data(), n_theta + 1, nv, trend, estimate_trend, beta, grad != nullptr, &o, nullptr, false);
  if (rc != BOGP_OK) return rc;
  const int N = h->N, d = h->d, ldr = h->ldr;
  const double tv = sigma2 + nv, TWO_PI = 2.0 * 3.141592653589793;
  const int ptrend = trend_size(trend, d);
  double v;
  if (estimate_trend && ptrend > 1) {
    // p > 1 (:850-860): (N - p) log(2 pi tv) - log det(F^T F) + 2 sum log diag L + log prod diag(G)^2 + rho.rho / tv
    //   det(F^T F): a constant of (training set, basis) -- F^T F on the device, its p x p Cholesky on the host, cached;
    //   diag(G) = diag(R2) diag(R1) of the two CholeskyQR passes (G = R2 R1, both upper triangular)
    hipStream_t st = h->stream;
    const int ldp = h->ldp;
    if (h->reml_ftf_basis != h->tr_built) {
      const double one = 1.0, zero = 0.0;
      HIPCHK(h, launch_gemm(1, 0, ptrend, ptrend, N, one, h->dF, N, h->dF, N, zero, h->dAT, ldp, st, 0, &h->gsplit));
      std::vector<double> a((size_t)ldp * ptrend);
      HIPCHK(h, hipMemcpyAsync(a.data(), h->dAT, a.size() * sizeof(double), hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
      double ld2 = 0.0;  // log det by an unblocked host Cholesky of the p x p Gram matrix (column-major, lower)
      for (int j = 0; j < ptrend; ++j) {
        double dj = a[(size_t)j * ldp + j];
        for (int k = 0; k < j; ++k) dj -= a[(size_t)k * ldp + j] * a[(size_t)k * ldp + j];
        if (!(dj > 0)) FAIL(h, BOGP_ERR_NOT_POSDEF, "trend basis is rank deficient (F^T F not positive definite at column %d)", j);
        const double ljj = std::sqrt(dj);
        a[(size_t)j * ldp + j] = ljj;
        ld2 += 2.0 * std::log(ljj);
        for (int i = j + 1; i < ptrend; ++i) {
          double s_ = a[(size_t)j * ldp + i];
          for (int k = 0; k < j; ++k) s_ -= a[(size_t)k * ldp + i] * a[(size_t)k * ldp + j];
          a[(size_t)j * ldp + i] = s_ / ljj;
        }
      }
      h->reml_logdet_ftf = ld2;
      h->reml_ftf_basis = h->tr_built;
    }
    std::vector<double> dg((size_t)2 * ptrend);
    for (int pass = 0; pass < 2; ++pass)
      HIPCHK(h, hipMemcpy2DAsync(dg.data() + (size_t)pass * ptrend, sizeof(double), h->dA[pass], (size_t)(ldp + 1) * sizeof(double),
                                 sizeof(double), ptrend, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    double lg = 0.0;
    for (double x : dg) lg += std::log(std::fabs(x));
    v = -0.5 * ((N - ptrend) * std::log(TWO_PI * tv) - h->reml_logdet_ftf + 2.0 * o.logdet + 2.0 * lg + o.rho_ss / tv);
  } else if (estimate_trend)  // p = 1: det(F^T F) = N, prod(diag G)^2 = |Ft|^2  (:850-860)
    v = -0.5 * ((N - 1) * std::log(TWO_PI * tv) - std::log((double)N) + 2.0 * o.logdet + std::log(o.ftft) + o.rho_ss / tv);
  else if (h->n_t > 1) {
    // (scalar + rho^T rho / tv).sum() over the n_t x n_t matrix: n_t^2 times the scalar terms + sum_ab rho_a . rho_b = |sum_a rho_a|^2
    const int T = h->n_t;
    std::vector<double> rho((size_t)T * N);
    HIPCHK(h, hipMemcpyAsync(rho.data(), h->drho_base, rho.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double cross = 0.0;  // row by row of rho^T rho, like the matrix the reference sums
    for (int a = 0; a < T; ++a)
      for (int b = 0; b < T; ++b) {
        double s_ = 0.0;
        for (int i = 0; i < N; ++i) s_ += rho[(size_t)a * N + i] * rho[(size_t)b * N + i];
        cross += s_;
      }
    v = -0.5 * ((double)T * T * (N * std::log(TWO_PI * tv) - 2.0 * o.logdet) + cross / tv);
  } else  // the reference SUBTRACTS the log-determinant here (:861-866)
    v = -0.5 * (N * std::log(TWO_PI * tv) - 2.0 * o.logdet + o.rho_ss / tv);
  if (!std::isfinite(v)) FAIL(h, BOGP_ERR_NOT_POSDEF, "restricted log-likelihood is not finite (%g)", v);
  const bool positive = v > 0;  // exp(llf) > 1
  *llf = v;
  if (grad) {
    hipStream_t st = h->stream;
    if (!h->dRinv) HIPCHK(h, hipMalloc((void**)&h->dRinv, (size_t)UUT_PARTS * h->cap_ld * h->cap_ld * sizeof(double)));
    int nparts = UUT_PARTS;
    HIPCHK(h, launch_uut(h->dU, h->dRinv, ldr, st, &nparts));
    const double* qv = nullptr;
    double c2 = 0.0;
    if (estimate_trend && ptrend > 1) {
      // term = (L^-T Q)(L^-T Q)^T = W S W^T with W = L^-T Ft (N x p) and S = (Ft^T Ft)^-1: folded into the first slice of
      // R^-1 as R^-1 - tv W S W^T (two k_gemm64 products with inner dimension p), after which the p = 1 code below applies
      // with no separate q vector: the contraction sees R^-1 - tv term, and its trace is tr(R^-1) - tv tr(term)
      const double one = 1.0, zero = 0.0, mtv = -tv;
      HIPCHK(h, launch_gemm(0, 0, N, ptrend, N, one, h->dU, ldr, h->dFt, N, zero, h->dQ1, N, st, 0, &h->gsplit));
      HIPCHK(h, launch_gemm(0, 0, N, ptrend, ptrend, one, h->dQ1, N, h->dSinv, ptrend, zero, h->dWp, N, st, 0, &h->gsplit));
      HIPCHK(h, launch_gemm(0, 1, N, N, ptrend, mtv, h->dWp, N, h->dQ1, N, one, h->dRinv, ldr, st, 0, &h->gsplit));
    } else if (estimate_trend) {  // q = L^-T Q = (L^-T Ft) / G
      HIPCHK(h, hipMemsetAsync(h->dw, 0, h->Np * sizeof(double), st));
      HIPCHK(h, launch_gemv2(h->dU, ldr, N, 2, h->dft, nullptr, h->dw, nullptr, h->dgemv_scratch, st));
      qv = h->dw;
      c2 = tv / o.ftft;
    }
    const int nblk = grad_contract_blocks(N);
    int e = ensure(h, &h->dgrad_partial, &h->grad_partial_cap, (size_t)nblk * (d + 1) + (d + 4));
    if (e) return e;
    GradVecs gv;
    gv.v = h->dgamma; gv.stride = 0; gv.n = 1; gv.c0 = 1.0; gv.cA[0] = gv.cB[0] = 1.0 / tv;
    HIPCHK(h, launch_grad_contract(kernel, h->dX, N, d, h->dtheta, gv, qv, c2, h->dRinv, ldr, nparts,
                                   (size_t)ldr * ldr, h->dgrad_partial, nblk, st));
    double* dS = h->dgrad_partial + (size_t)nblk * (d + 1);
    HIPCHK(h, launch_grad_reduce(h->dgrad_partial, nblk, d + 1, dS, st));
    HIPCHK(h, launch_trace_gg(h->dRinv, ldr, nparts, (size_t)ldr * ldr, N, h->dgamma, qv, dS + d + 1, st));
    std::vector<double> S(d + 4);
    HIPCHK(h, hipMemcpyAsync(S.data(), dS, (d + 4) * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    const double tr = S[d + 1], gg = S[d + 2], qq = (estimate_trend && ptrend == 1) ? S[d + 3] / o.ftft : 0.0;
    const double diag = -0.5 * (tr / tv - gg / (tv * tv) - qq);  // sum over the diagonal of (Cinv - gamma_ gamma_^T - term)
    if (n_theta == d) {
      for (int k = 0; k < d; ++k) grad[k] = S[k];
      grad[d] = S[d] / tv + diag;                              // d / d sigma2: C_grad = R0 (:883)
      if (mode == BOGP_MODE_NOISE_ESTIM) grad[d + 1] = diag;   // d / d noise_var: C_grad = I (:885-887)
    } else {
      // isotropic theta (one entry for d dimensions): the reference still builds the (N, N, d) tensor of PER-DIMENSION derivatives
      // (corr_grad_theta, :736-770: `diff` has d slices whatever len(theta) is), appends R0 [and I], and reads slice i for parameter i
      // (:889-900) -- so entry 0 is the derivative w.r.t. the FIRST dimension's weight alone, and for d >= 2 the sigma2 entry is the
      // second dimension's slice, not R0's.  Reproduced as it is (as for the concentrated likelihood, G18): slices 0 .. n_par - 1 of
      // [dims 0 .. d - 1 | R0 | I].
      std::vector<double> full((size_t)d + 2);
      for (int k = 0; k < d; ++k) full[k] = S[k];
      full[d] = S[d] / tv + diag;
      full[d + 1] = diag;
      for (int i = 0; i < n_par; ++i) grad[i] = full[i];
    }
  }
  if (positive) FAIL(h, BOGP_ERR_LLF_POSITIVE, "restricted log-likelihood %g > 0 is rejected by the reference (gpr.py:868-871)", v);
  return BOGP_OK;
}

extern "C" int bogp_commit(bogp_handle* h, int kernel, int mode, const double* par, int n_par, double noise_var, int trend,
                           int estimate_trend, double beta, double* llf) {
  if (!h) return BOGP_ERR_INVALID;
  if (!par || n_par <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_commit: par must be non-null");
  h->committed = false;
  FitOut o;
  std::vector<double> th;
  // committing builds a state; rejecting llf > 0 is a rule of the likelihood EVALUATION (bogp_nll), and the REML path
  // commits at parameters whose concentrated value may well be positive
  int rc = factorize(h, kernel, mode, par, n_par, noise_var, trend, estimate_trend, beta, true, &o, &th, false);
  if (llf) *llf = o.llf;
  if (rc != BOGP_OK) return rc;
  const int N = h->N, d = h->d, Np = h->Np;
  hipStream_t st = h->stream;
  const int ldr = h->ldr;
  // One step of iterative refinement of gamma = R^-1 (y - beta 1) (gpr.py:787-788) against R recomputed from X: the factor of
  // the blocked Cholesky applies explicit inverses of its diagonal blocks (conditionally backward stable), which at
  // cond(R) ~ 1e12 left the posterior mean ~100x further from the exact one than a LAPACK solve (profiles/r03_refine_inverse.txt,
  // r03_refine_gamma.txt).  gamma += L^-T L^-1 (b - R gamma): one N^2 d pass + two triangular matrix-vector products, at
  // commit only, every trend basis (b = y - F beta with the committed coefficients); BOGP_REFINE_GAMMA=0 switches it off.
  {
    static const int steps = [] { const char* e = getenv("BOGP_REFINE_GAMMA"); return e ? atoi(e) : 1; }();
    const int pt = trend_size(trend, d);
    if (steps > 0) {
      int e = ensure(h, &h->dbatch, &h->batch_cap, (size_t)4 * N);
      if (e) return e;
      double *dres = h->dbatch, *dt1 = dres + N, *dt2 = dt1 + N, *db = dt2 + N;
      for (int t = 0; t < h->n_t; ++t) {
        double* g = h->dgamma_base + (size_t)t * Np;
        const double* yt_ = h->dy_base + (size_t)t * N;
        if (pt == 1) {
          HIPCHK(h, launch_sub_const(yt_, o.beta, db, N, st));  // b = y - beta 1
        } else {  // b = y - F beta with the committed coefficients (fixed, or the GLS estimate of trend_solve)
          const double one = 1.0, mone = -1.0;
          HIPCHK(h, hipMemcpyAsync(db, yt_, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, st));
          HIPCHK(h, launch_gemm(0, 0, N, 1, pt, mone, h->dF, N, h->dbetav, pt, one, db, N, st, 0, &h->gsplit));
        }
        for (int it = 0; it < steps; ++it) {
          HIPCHK(h, launch_resid_gamma(kernel, h->R_div, h->dX, N, d, h->dtheta, h->R_a, h->R_b, h->R_diag, db, g, dres, st));
          HIPCHK(h, launch_gemv2(h->dV, ldr, N, 1, dres, nullptr, dt1, nullptr, h->dgemv_scratch, st));
          HIPCHK(h, launch_gemv2(h->dU, ldr, N, 2, dt1, nullptr, dt2, nullptr, h->dgemv_scratch, st));
          HIPCHK(h, launch_add_vec(g, dt2, N, st));
        }
      }
    }
  }
  // V = L^-1 (the triangular solve of gpr.py:494 becomes a triangular GEMM against V)
  if (!h->dVp) HIPCHK(h, hipMalloc((void**)&h->dVp, (size_t)h->cap_ld * h->cap_ld * sizeof(double)));
  HIPCHK(h, launch_pack_V(h->dV, N, ldr, Np, h->dVp, st));
  // w = L^-T Ft  (so that Ft^T L^-1 r = w . r, gpr.py:496-498)
  HIPCHK(h, hipMemsetAsync(h->dw, 0, Np * sizeof(double), st));
  const int ptrend = trend_size(trend, d);
  if (estimate_trend && ptrend == 1) {
    HIPCHK(h, launch_gemv2(h->dU, ldr, N, 2, h->dft, nullptr, h->dw, nullptr, h->dgemv_scratch, st));
  }
  if (ptrend > 1) {
    h->h_betav.assign(ptrend, 0.0);
    h->h_Sinv.assign((size_t)ptrend * ptrend, 0.0);
    HIPCHK(h, hipMemcpyAsync(h->h_betav.data(), h->dbetav, ptrend * sizeof(double), hipMemcpyDeviceToHost, st));
    if (estimate_trend) {  // W = L^-T Ft (N x p), zero rows in the padding
      const double one = 1.0, zero = 0.0;
      HIPCHK(h, hipMemsetAsync(h->dWp, 0, (size_t)Np * ptrend * sizeof(double), st));
      HIPCHK(h, launch_gemm(0, 0, N, ptrend, N, one, h->dU, ldr, h->dFt, N, zero, h->dWp, Np, st, 0, &h->gsplit));
      // the column sides of the two per-chunk trend products on k_mm128 (run_sweep): W^T and (Ft^T Ft)^-1, zero padded to 128 columns
      const int pp = (ptrend + 127) / 128 * 128;
      HIPCHK(h, launch_transpose_pad(h->dWp, Np, Np, ptrend, h->dWpT, pp, st));
      HIPCHK(h, hipMemsetAsync(h->dSinvP, 0, (size_t)pp * pp * sizeof(double), st));
      HIPCHK(h, launch_transpose_pad(h->dSinv, ptrend, ptrend, ptrend, h->dSinvP, pp, st));
      HIPCHK(h, hipMemcpyAsync(h->h_Sinv.data(), h->dSinv, (size_t)ptrend * ptrend * sizeof(double), hipMemcpyDeviceToHost, st));
      // more than 32 columns (a quadratic basis; a linear one from d = 32): the u term as p extra rows of the packed factor (k_pack_Vx)
      h->vx_Ne = h->vx_Nt = 0;
      if (trend_rows_enabled() && ptrend >= trend_rows_min()) {
        const int cols = contract_cols_per_group();
        const int Ne = (Np + cols - 1) / cols * cols, Nt = Ne + (ptrend + 31) / 32 * 32;
        int e2;
        if ((e2 = ensure(h, &h->dAtx, &h->atx_cap, (size_t)N * ptrend))) return e2;
        if (h->vpx_cap < (size_t)Nt * Nt / 2 || !h->dVpx) {
          dfree(h->dVpx);
          h->vpx_cap = 0;
          HIPCHK(h, hipMalloc((void**)&h->dVpx, (size_t)Nt * Nt / 2 * sizeof(double2)));
          h->vpx_cap = (size_t)Nt * Nt / 2;
        }
        HIPCHK(h, launch_gemm(0, 0, N, ptrend, ptrend, one, h->dWp, Np, h->dGinv, ptrend, zero, h->dAtx, N, st, 0, &h->gsplit));  // W G^-1
        HIPCHK(h, launch_pack_Vx(h->dV, N, ldr, h->dAtx, N, h->dGinv, ptrend, Ne, Nt, h->dVpx, st));
        h->vx_Ne = Ne; h->vx_Nt = Nt;
      }
    }
  }
  // [d][Np] + two zero rows: k_sweep_small walks the dimensions three at a time
  if (!h->dXthT) HIPCHK(h, hipMalloc((void**)&h->dXthT, (size_t)(h->cap_d + 2) * h->cap_ld * sizeof(double)));
  if (!h->dXnorm) HIPCHK(h, hipMalloc((void**)&h->dXnorm, (size_t)h->cap_ld * sizeof(double)));
  HIPCHK(h, launch_scale_transpose(h->dX, N, d, Np, h->dsqrt_theta, h->dXthT, h->dXnorm, st));
  HIPCHK(h, hipMemsetAsync(h->dXthT + (size_t)d * Np, 0, (size_t)2 * Np * sizeof(double), st));
  HIPCHK(h, hipStreamSynchronize(st));
  h->kernel = kernel; h->mode = mode; h->estimate_trend = estimate_trend;
  h->trend = trend; h->p = ptrend;
  h->beta = o.beta; h->G = o.G; h->sigma2 = o.sigma2; h->noise_var = o.noise_var; h->llf = o.llf; h->ftft = o.ftft;
  h->sigma2_t.assign(o.sigma2_t, o.sigma2_t + h->n_t);
  h->nv_t.assign(o.nv_t, o.nv_t + h->n_t);
  h->committed = true;
  select_target(h, 0);
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
    if (!h->dRinv) HIPCHK(h, hipMalloc((void**)&h->dRinv, (size_t)UUT_PARTS * h->cap_ld * h->cap_ld * sizeof(double)));
    HIPCHK(h, launch_copy_lower(h->dR, N, h->ldr, h->dRinv, st));
    HIPCHK(h, hipMemcpyAsync(C, h->dRinv, (size_t)N * N * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  if (gamma) HIPCHK(h, hipMemcpyAsync(gamma, h->dgamma, N * sizeof(double), hipMemcpyDeviceToHost, st));
  if (rho) HIPCHK(h, hipMemcpyAsync(rho, h->drho, N * sizeof(double), hipMemcpyDeviceToHost, st));
  if (Yt) HIPCHK(h, hipMemcpyAsync(Yt, h->dyt, N * sizeof(double), hipMemcpyDeviceToHost, st));
  if (h->estimate_trend && h->p == 1) {  // p > 1: bogp_get_trend_state
    if (Ft) HIPCHK(h, hipMemcpyAsync(Ft, h->dft, N * sizeof(double), hipMemcpyDeviceToHost, st));
    if (Q) HIPCHK(h, hipMemcpyAsync(Q, h->dft, N * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  HIPCHK(h, hipStreamSynchronize(st));
  if (h->estimate_trend && h->p == 1 && Q)
    for (int i = 0; i < N; ++i) Q[i] /= h->G;
  if (G) *G = h->G;
  if (beta) *beta = h->beta;
  if (sigma2) *sigma2 = h->sigma2;
  if (noise_var) *noise_var = h->target < (int)h->nv_t.size() ? h->nv_t[h->target] : h->noise_var;
  return BOGP_OK;
}

// State of a polynomial trend (p > 1): Ft, Q (N x p, row-major), G (p x p, row-major, upper, positive diagonal), beta (p).
// Works for p = 1 as well.  Any pointer may be NULL.
extern "C" int bogp_get_trend_state(bogp_handle* h, double* Ft, double* Q, double* G, double* beta) {
  if (!h) return BOGP_ERR_INVALID;
  if (!h->committed) FAIL(h, BOGP_ERR_INVALID, "bogp_get_trend_state: no committed state");
  const int N = h->N, pt = h->p;
  hipStream_t st = h->stream;
  HIPCHK(h, hipSetDevice(h->device));
  if (pt == 1) {
    std::vector<double> ft(N, 0.0);
    if (h->estimate_trend) HIPCHK(h, hipMemcpy(ft.data(), h->dft, N * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < N; ++i) {
      if (Ft) Ft[i] = ft[i];
      if (Q) Q[i] = h->estimate_trend ? ft[i] / h->G : 0.0;
    }
    if (G) *G = h->G;
    if (beta) *beta = h->beta;
    return BOGP_OK;
  }
  if (beta) for (int c = 0; c < pt; ++c) beta[c] = h->h_betav[c];
  if (!h->estimate_trend) return BOGP_OK;  // Ft / Q / G exist only when the coefficients are estimated (gpr.py:801-806)
  std::vector<double> tmp((size_t)N * pt);
  for (int which = 0; which < 2; ++which) {
    double* dst = which == 0 ? Ft : Q;
    if (!dst) continue;
    HIPCHK(h, hipMemcpyAsync(tmp.data(), which == 0 ? h->dFt : h->dQ, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    for (int i = 0; i < N; ++i)
      for (int c = 0; c < pt; ++c) dst[(size_t)i * pt + c] = tmp[(size_t)c * N + i];
  }
  if (G) {  // G = R2 R1 with R = L^T of the two CholeskyQR passes (lower triangles of dA[1], dA[0])
    const int ldp = h->ldp;
    std::vector<double> a0((size_t)ldp * ldp), a1((size_t)ldp * ldp);
    HIPCHK(h, hipMemcpyAsync(a0.data(), h->dA[0], a0.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipMemcpyAsync(a1.data(), h->dA[1], a1.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    for (int i = 0; i < pt; ++i)
      for (int j = 0; j < pt; ++j) {
        double acc = 0.0;
        for (int k = i; k <= j; ++k) acc += a1[(size_t)i * ldp + k] * a0[(size_t)k * ldp + j];  // R2[i][k] = L2[k][i], R1[k][j] = L1[j][k]
        G[(size_t)i * pt + j] = j >= i ? acc : 0.0;
      }
  }
  return BOGP_OK;
}

// ------------------------------------------------------------------------------------------------------
// candidates
// ------------------------------------------------------------------------------------------------------
// The winners a sweep left on the device (dbest_* / dtopk_*) refer to rows of the candidate set they were computed on: any
// change of that set -- and a sweep of the other flavour, which overwrites dbest_* -- makes them unusable for
// bogp_exchange_* (stale or out-of-range rows would be packed otherwise).
static void invalidate_sweep_results(bogp_handle* h) { h->last_q = h->last_topk_q = h->last_topk_k = 0; }

// ---- lazy upload: the copy of chunk c + 1 runs beside the kernels of chunk c --------------------------------------------------------
// rows [lazy_done, upto) onto the copy stream (a copy from pageable memory blocks the HOST while the runtime stages it, not the
// device: the kernels queued before it keep running), the event re-recorded behind it
static int lazy_copy_to(bogp_handle* h, int64_t upto) {
  if (!h->hXs_lazy) return BOGP_OK;
  upto = std::min<int64_t>(upto, h->M);
  if (upto <= h->lazy_done) return BOGP_OK;
  const size_t d = (size_t)h->d;
  HIPCHK(h, hipMemcpyAsync(h->dXs_owned + (size_t)h->lazy_done * d, h->hXs_lazy + (size_t)h->lazy_done * d,
                           (size_t)(upto - h->lazy_done) * d * sizeof(double), hipMemcpyHostToDevice, h->stream_copy));
  HIPCHK(h, hipEventRecord(h->ev_copy, h->stream_copy));
  h->lazy_done = upto;
  return BOGP_OK;
}
// `st` waits for every copy enqueued so far
static int lazy_wait(bogp_handle* h, hipStream_t st) {
  if (!h->hXs_lazy || h->lazy_done == 0) return BOGP_OK;
  HIPCHK(h, hipStreamWaitEvent(st, h->ev_copy, 0));
  return BOGP_OK;
}
// everything copied and visible to the main stream; the host rows are not needed any more
static int lazy_finish(bogp_handle* h) {
  if (!h->hXs_lazy) return BOGP_OK;
  int e = lazy_copy_to(h, h->M);
  if (e) return e;
  HIPCHK(h, hipStreamSynchronize(h->stream_copy));
  h->hXs_lazy = nullptr;
  return BOGP_OK;
}
static int lazy_drop(bogp_handle* h) {  // new candidates arrive: pending copies of the old ones must not land later
  if (h->hXs_lazy) {
    HIPCHK(h, hipStreamSynchronize(h->stream_copy));
    h->hXs_lazy = nullptr;
  }
  return BOGP_OK;
}

extern "C" int bogp_candidates_upload_lazy(bogp_handle* h, const double* Xs, int64_t M) {
  if (!h) return BOGP_ERR_INVALID;
  invalidate_sweep_results(h);
  if (!Xs || M <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_upload_lazy: Xs must be non-null and M > 0");
  if (!h->dX) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_upload_lazy: call bogp_set_train first (d is unknown)");
  HIPCHK(h, hipSetDevice(h->device));
  int e = lazy_drop(h);
  if (e) return e;
  if (!h->stream_copy) HIPCHK(h, hipStreamCreateWithFlags(&h->stream_copy, hipStreamNonBlocking));
  if (!h->ev_copy) HIPCHK(h, hipEventCreateWithFlags(&h->ev_copy, hipEventDisableTiming));
  if ((e = ensure(h, &h->dXs_owned, &h->xs_cap, (size_t)M * h->d))) return e;
  // (the buffer may have been re-allocated, and the last sweep may still read the old candidates: the copies start behind it)
  HIPCHK(h, hipEventRecord(h->ev_copy, h->stream));
  HIPCHK(h, hipStreamWaitEvent(h->stream_copy, h->ev_copy, 0));
  h->dXs = h->dXs_owned;
  h->M = M;
  h->hXs_lazy = Xs;
  h->lazy_done = 0;
  // the first 8 MB go now: they are what the first chunk of the next sweep waits for
  return lazy_copy_to(h, std::max<int64_t>(1, ((int64_t)8 << 20) / (int64_t)(h->d * sizeof(double))));
}

extern "C" int bogp_candidates_upload(bogp_handle* h, const double* Xs, int64_t M) {
  if (!h) return BOGP_ERR_INVALID;
  invalidate_sweep_results(h);
  if (!h->dX) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_upload: call bogp_set_train first (d is unknown)");
  if (!Xs || M <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_upload: Xs must be non-null and M > 0");
  HIPCHK(h, hipSetDevice(h->device));
  int e = lazy_drop(h);
  if (e) return e;
  if ((e = ensure(h, &h->dXs_owned, &h->xs_cap, (size_t)M * h->d))) return e;
  HIPCHK(h, hipMemcpyAsync(h->dXs_owned, Xs, (size_t)M * h->d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->dXs = h->dXs_owned;
  h->M = M;
  return BOGP_OK;
}

// shared front end of the three on-device generators: validates the box, sizes the candidate buffer, stages lo / hi
static int generate_prepare(bogp_handle* h, const char* who, const double* lo, const double* hi, int64_t M, int64_t first) {
  invalidate_sweep_results(h);
  if (!h->dX) FAIL(h, BOGP_ERR_INVALID, "%s: call bogp_set_train first (d is unknown)", who);
  if (!lo || !hi || M <= 0 || first < 0) FAIL(h, BOGP_ERR_INVALID, "%s: bounds must be non-null, M > 0, first row/index >= 0", who);
  const int d = h->d;
  for (int k = 0; k < d; ++k)
    if (!(std::isfinite(lo[k]) && std::isfinite(hi[k]) && lo[k] <= hi[k])) FAIL(h, BOGP_ERR_INVALID, "%s: bad bounds in dimension %d", who, k);
  HIPCHK(h, hipSetDevice(h->device));
  int e = lazy_drop(h);
  if (e) return e;
  if ((e = ensure(h, &h->dXs_owned, &h->xs_cap, (size_t)M * d))) return e;
  if ((e = ensure(h, &h->dbounds, &h->bounds_cap, (size_t)2 * d))) return e;
  HIPCHK(h, hipMemcpyAsync(h->dbounds, lo, d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->dbounds + d, hi, d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  return BOGP_OK;
}

extern "C" int bogp_candidates_set_transform(bogp_handle* h, const int* scale, const int* precision, const double* lo,
                                             const double* hi) {
  if (!h) return BOGP_ERR_INVALID;
  if (!h->dX) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_set_transform: call bogp_set_train first (d is unknown)");
  HIPCHK(h, hipSetDevice(h->device));
  if (!scale && !precision) {  // back to plain designs
    h->h_xform.clear();
    return BOGP_OK;
  }
  const int d = h->d;
  std::vector<double> spec((size_t)4 * d);
  bool any = false;
  for (int k = 0; k < d; ++k) {
    const int sc = scale ? scale[k] : BOGP_SCALE_LINEAR, pr = precision ? precision[k] : -1;
    if (sc < BOGP_SCALE_LINEAR || sc > BOGP_SCALE_BILOG) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_set_transform: unknown scale id %d in dimension %d", sc, k);
    if (pr > 15) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_set_transform: precision %d in dimension %d (at most 15 decimals)", pr, k);
    if (pr >= 0 && (!lo || !hi || !(lo[k] <= hi[k]))) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_set_transform: rounding needs the variable's bounds (dimension %d)", k);
    spec[4 * k] = sc; spec[4 * k + 1] = pr < 0 ? -1 : pr;
    spec[4 * k + 2] = lo ? lo[k] : 0.0; spec[4 * k + 3] = hi ? hi[k] : 0.0;
    any = any || sc != BOGP_SCALE_LINEAR || pr >= 0;
  }
  if (!any) {
    h->h_xform.clear();
    return BOGP_OK;
  }
  if (!h->dxform) HIPCHK(h, hipMalloc((void**)&h->dxform, (size_t)4 * BOGP_MAX_DIM * sizeof(double)));
  h->h_xform = spec;
  HIPCHK(h, hipMemcpy(h->dxform, spec.data(), spec.size() * sizeof(double), hipMemcpyHostToDevice));
  return BOGP_OK;
}

static int generate_finish(bogp_handle* h, int64_t M) {
  if (!h->h_xform.empty() && (int)h->h_xform.size() == 4 * h->d)
    HIPCHK(h, launch_candidates_transform(h->dXs_owned, M * h->d, h->d, h->dxform, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));  // lo / hi (and sv) are caller memory
  h->dXs = h->dXs_owned;
  h->M = M;
  return BOGP_OK;
}

extern "C" int bogp_candidates_generate(bogp_handle* h, const double* lo, const double* hi, int64_t M, uint64_t seed,
                                        int64_t first_row) {
  if (!h) return BOGP_ERR_INVALID;
  int e = generate_prepare(h, "bogp_candidates_generate", lo, hi, M, first_row);
  if (e) return e;
  const int d = h->d;
  HIPCHK(h, launch_generate_uniform(h->dXs_owned, M * d, d, h->dbounds, h->dbounds + d, seed, (uint64_t)first_row * (uint64_t)d, h->stream));
  return generate_finish(h, M);
}

extern "C" int bogp_candidates_generate_lhs(bogp_handle* h, const double* lo, const double* hi, int64_t M, uint64_t seed,
                                            int64_t first_row, int64_t n_strata) {
  if (!h) return BOGP_ERR_INVALID;
  int e = generate_prepare(h, "bogp_candidates_generate_lhs", lo, hi, M, first_row);
  if (e) return e;
  if (n_strata < first_row + M) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_generate_lhs: rows [%lld, %lld) exceed the %lld strata", (long long)first_row, (long long)(first_row + M), (long long)n_strata);
  const int d = h->d;
  HIPCHK(h, launch_generate_lhs(h->dXs_owned, M * d, d, h->dbounds, h->dbounds + d, seed, (uint64_t)first_row * (uint64_t)d, (uint64_t)n_strata, h->stream));
  return generate_finish(h, M);
}

// largest design the maximin criterion accepts: M^2 d / 2 pair terms per trial design (2^18 points, d = 20: ~0.1 s each)
static constexpr int64_t BOGP_MAXIMIN_MAX_POINTS = (int64_t)1 << 18;

extern "C" int bogp_candidates_min_pdist2(bogp_handle* h, double* min_sq) {
  if (!h) return BOGP_ERR_INVALID;
  if (!h->dXs || h->M <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_min_pdist2: no candidates");
  if (!min_sq) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_min_pdist2: null output");
  if (h->M > BOGP_MAXIMIN_MAX_POINTS) FAIL(h, BOGP_ERR_UNSUPPORTED, "bogp_candidates_min_pdist2: %lld points exceed the %lld-point limit of the O(M^2 d) pair sweep", (long long)h->M, (long long)BOGP_MAXIMIN_MAX_POINTS);
  HIPCHK(h, hipSetDevice(h->device));
  {
    const int e = lazy_finish(h);
    if (e) return e;
  }
  unsigned long long* dout = (unsigned long long*)h->dscal;
  HIPCHK(h, launch_min_pdist2(h->dXs, (int)h->M, h->d, dout, h->stream));
  unsigned long long bits = 0;
  HIPCHK(h, hipMemcpyAsync(&bits, dout, sizeof(bits), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (bits == ~0ull) {
    *min_sq = INFINITY;  // fewer than two points
  } else {
    memcpy(min_sq, &bits, sizeof(double));
  }
  return BOGP_OK;
}

extern "C" int bogp_candidates_generate_lhs_maximin(bogp_handle* h, const double* lo, const double* hi, int64_t M, uint64_t seed,
                                                    int iterations, double* best_min_dist, int* best_iteration) {
  if (!h) return BOGP_ERR_INVALID;
  if (iterations < 1 || iterations > 64) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_generate_lhs_maximin: iterations = %d outside [1, 64]", iterations);
  if (M > BOGP_MAXIMIN_MAX_POINTS) FAIL(h, BOGP_ERR_UNSUPPORTED, "bogp_candidates_generate_lhs_maximin: %lld points exceed the %lld-point limit of the O(M^2 d) pair sweep", (long long)M, (long long)BOGP_MAXIMIN_MAX_POINTS);
  int e = generate_prepare(h, "bogp_candidates_generate_lhs_maximin", lo, hi, M, 0);
  if (e) return e;
  const int d = h->d;
  // trial designs live in the unit cube, un-transformed (pyDOE measures the design before the caller scales it)
  if ((e = ensure(h, &h->dbatch, &h->batch_cap, (size_t)2 * d))) return e;
  std::vector<double> unit((size_t)2 * d, 0.0);
  for (int k = 0; k < d; ++k) unit[d + k] = 1.0;
  HIPCHK(h, hipMemcpyAsync(h->dbatch, unit.data(), (size_t)2 * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  unsigned long long* dout = (unsigned long long*)h->dscal;
  double best = -1.0;
  int best_it = 0;
  for (int it = 0; it < iterations; ++it) {
    const uint64_t s_it = seed + 0x9E3779B97F4A7C15ull * (uint64_t)it;
    HIPCHK(h, launch_generate_lhs(h->dXs_owned, M * d, d, h->dbatch, h->dbatch + d, s_it, 0, (uint64_t)M, h->stream));
    HIPCHK(h, launch_min_pdist2(h->dXs_owned, (int)M, d, dout, h->stream));
    unsigned long long bits = 0;
    HIPCHK(h, hipMemcpyAsync(&bits, dout, sizeof(bits), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double msq = INFINITY;
    if (bits != ~0ull) memcpy(&msq, &bits, sizeof(double));
    const double dist = std::sqrt(msq);  // pyDOE compares the distances: `if maxdist < np.min(d)` keeps the EARLIER design on ties
    if (best < dist) {
      best = dist;
      best_it = it;
    }
  }
  const uint64_t s_best = seed + 0x9E3779B97F4A7C15ull * (uint64_t)best_it;
  HIPCHK(h, launch_generate_lhs(h->dXs_owned, M * d, d, h->dbounds, h->dbounds + d, s_best, 0, (uint64_t)M, h->stream));
  if (best_min_dist) *best_min_dist = best;
  if (best_iteration) *best_iteration = best_it;
  return generate_finish(h, M);
}

extern "C" int bogp_candidates_generate_sobol(bogp_handle* h, const double* lo, const double* hi, int64_t M,
                                              int64_t first_index, const uint64_t* sv, int bits) {
  if (!h) return BOGP_ERR_INVALID;
  int e = generate_prepare(h, "bogp_candidates_generate_sobol", lo, hi, M, first_index);
  if (e) return e;
  if (!sv || bits < 1 || bits > 53) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_generate_sobol: direction numbers must be non-null with 1 <= bits <= 53");
  if (((uint64_t)(first_index + M - 1) >> bits) != 0) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_generate_sobol: index %lld needs more than %d bits", (long long)(first_index + M - 1), bits);
  const int d = h->d;
  if ((e = ensure(h, &h->dsobol, &h->sobol_cap, (size_t)d * bits))) return e;
  HIPCHK(h, hipMemcpyAsync(h->dsobol, sv, (size_t)d * bits * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, launch_generate_sobol(h->dXs_owned, M * d, d, h->dbounds, h->dbounds + d, (const uint64_t*)h->dsobol, bits,
                                  (uint64_t)first_index * (uint64_t)d, h->stream));
  return generate_finish(h, M);
}

extern "C" int bogp_candidates_read(bogp_handle* h, const int64_t* rows, int n, double* out) {
  if (!h) return BOGP_ERR_INVALID;
  if (!h->dXs || h->M <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_read: no candidates");
  if (!rows || !out || n < 0) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_read: null pointer");
  HIPCHK(h, hipSetDevice(h->device));
  {
    const int e = lazy_finish(h);
    if (e) return e;
  }
  const int d = h->d;
  for (int i = 0; i < n; ++i)
    if (rows[i] < 0 || rows[i] >= h->M) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_read: row %lld outside [0, %lld)", (long long)rows[i], (long long)h->M);
  for (int i = 0; i < n;) {  // one copy per run of consecutive rows
    int j = i + 1;
    while (j < n && rows[j] == rows[j - 1] + 1) ++j;
    HIPCHK(h, hipMemcpyAsync(out + (size_t)i * d, h->dXs + (size_t)rows[i] * d, (size_t)(j - i) * d * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    i = j;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return BOGP_OK;
}

extern "C" int bogp_candidates_bind(bogp_handle* h, const void* d_Xs, int64_t M) {
  if (!h) return BOGP_ERR_INVALID;
  if (!d_Xs || M <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_candidates_bind: pointer must be non-null and M > 0");
  invalidate_sweep_results(h);
  {
    const int e = lazy_drop(h);
    if (e) return e;
  }
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

// Event times of the last sweep are read lazily (bogp_last_timing, or the next sweep): reading them needs the events
// to have completed, and an un-synchronised sweep (bogp_sweep without host outputs) must not wait for them.
static void collect_timing(bogp_handle* h) {
  if (!h->timing_pending) return;
  h->timing_pending = false;
  (void)hipSetDevice(h->device);
  if (h->timing_fused) {
    float ms = 0;
    (void)hipEventSynchronize(h->ev[1]);
    (void)hipEventElapsedTime(&ms, h->ev[0], h->ev[1]);
    h->t_corr_ms = 0; h->t_contract_ms = ms; h->t_acq_ms = 0;  // one kernel: reported as the contraction's time
    return;
  }
  constexpr int EPC = 5;
  h->t_corr_ms = h->t_contract_ms = h->t_acq_ms = 0;
  for (int64_t c = 0; c < h->n_chunks; ++c) {
    float a = 0, b2 = 0, c2 = 0;
    hipEvent_t* ev = &h->ev[(size_t)(c * EPC)];
    (void)hipEventSynchronize(ev[4]);
    (void)hipEventElapsedTime(&a, ev[0], ev[1]);
    (void)hipEventElapsedTime(&b2, ev[2], ev[3]);
    (void)hipEventElapsedTime(&c2, ev[3], ev[4]);
    h->t_corr_ms += a; h->t_contract_ms += b2; h->t_acq_ms += c2;
  }
}

static int run_sweep(bogp_handle* h, bool want_out, int q, const int* acq_id, const double* acq_par, double plugin,
                     int minimize, bool want_acq_out, bool need_var = true, bool sync = true) {
  collect_timing(h);  // the events are about to be re-recorded
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
  // trend-rows path (k_pack_Vx): the chunk carries Nt - Np extra rows (the hole up to a whole column group, then -f(x*)), the contraction
  // runs over the extended factor
  const bool vx_model = h->vx_Nt > 0 && h->p >= trend_rows_min() && h->estimate_trend;  // the committed model takes the trend-rows path ...
  const bool vx = vx_model && need_var;                                                     // ... and this call needs the variance
  const int Nrows = vx ? h->vx_Nt : Np;
  int64_t Mc = (int64_t)(chunk_bytes / ((size_t)Nrows * sizeof(double)) / 64) * 64;
  Mc = std::max<int64_t>(64, std::min<int64_t>(Mc, Mpad));
  const int nblk32 = Np / 32;
  // the training set is sliced into groups of 8 x 32 rows per producer workgroup: a function of N only, so that the
  // grouping of the partial sums of mu (hence every output bit) does not depend on the chunk size
  static const int nblk_env = [] { const char* e = getenv("BOGP_CORR_SPLIT"); return e ? std::max(2, atoi(e)) : 8; }();  // A/B switch (profiles/r05_corr_split_ab.txt)
  const int nblk_per_split = nblk_env;
  const int S = (nblk32 + nblk_per_split - 1) / nblk_per_split;
  const int cols = contract_cols_per_group();
  const int NJ16 = Nrows / 16;
  const int nJ_main = (Np + cols - 1) / cols;            // column groups of V: |L^-1 r|^2
  const int nJ = vx ? (Nrows + cols - 1) / cols : nJ_main;  // ... + the groups of the trend rows: |u|^2
  const int64_t nchunk = (M + Mc - 1) / Mc;
  const int64_t nblk_total = (M + 255) / 256 + nchunk;  // per-chunk block counts are rounded up

  // Small batches (the reference's one-point-per-call usage through L-BFGS-B): the tiled contraction would leave one
  // workgroup walking all N columns alone (~0.3 ms at N = 2048).  For M <= BOGP_SMALL_M the posterior is instead
  // r -> rt = V r (k_gemm64 with M right-hand sides) -> column reductions, feeding the same acquisition kernel.
  int small_m = 32;
  if (const char* env = getenv("BOGP_SMALL_M")) small_m = atoi(env);
  const bool one_launch = h->p == 1 && sweep_small_supported(Np, d, h->kernel);
  if (h->hXs_lazy && ((M <= small_m && h->p == 1) || one_launch || nchunk == 1)) {
    // nothing to overlap with: the whole upload first (one launch reads every candidate)
    const int el = lazy_finish(h);
    if (el) return el;
  }
  if (M <= small_m && h->p == 1) {
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
    HIPCHK(h, launch_gemm(0, 0, N, B, N, 1.0, h->dV, h->ldr, dr, N, 0.0, drt, N, st, 1));  // rt = V r, V lower with a zero upper triangle
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
    h->timing_pending = false;
    return BOGP_OK;
  }

  // Small training sets (Np <= 512, d <= 60, constant trend): the whole sweep is ONE launch of k_sweep_small -- producer,
  // triangular contraction, posterior, criteria and argmax fused, r never leaves LDS (kernels_small.hip).
  if (one_launch) {
    const int64_t nblk = std::max<int64_t>(sweep_small_blocks(M, h->n_cu), (M + 15) / 16);
    int e2;
    if (q > 0) {
      if ((e2 = ensure(h, &h->dblk_val, &h->blk_val_cap, (size_t)q * nblk))) return e2;
      if ((e2 = ensure(h, &h->dblk_idx, &h->blk_idx_cap, (size_t)q * nblk))) return e2;
      if (!h->dbest_val) HIPCHK(h, hipMalloc((void**)&h->dbest_val, BOGP_MAX_Q * sizeof(double)));
      if (!h->dbest_idx) HIPCHK(h, hipMalloc((void**)&h->dbest_idx, BOGP_MAX_Q * sizeof(int64_t)));
    }
    if (want_out) {
      if ((e2 = ensure(h, &h->dmu_out, &h->mu_out_cap, (size_t)M))) return e2;
      if ((e2 = ensure(h, &h->dmse_out, &h->mse_out_cap, (size_t)M))) return e2;
    }
    if (want_acq_out)
      if ((e2 = ensure(h, &h->dacq_out, &h->acq_out_cap, (size_t)q * M))) return e2;
    if (!h->dcounter) {
      HIPCHK(h, hipMalloc((void**)&h->dcounter, sizeof(unsigned int)));
      HIPCHK(h, hipMemsetAsync(h->dcounter, 0, sizeof(unsigned int), st));
    }
    SmallArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.Xs = h->dXs; sa.sqrt_theta = h->dsqrt_theta; sa.XthT = h->dXthT; sa.gamma = h->dgamma; sa.wvec = h->dw; sa.Vp = h->dVp;
    sa.M = M; sa.d = d; sa.Np = Np; sa.NJ16 = Np / 16; sa.NKP = Np / 8; sa.need_var = need_var ? 1 : 0;
    sa.beta = h->beta; sa.G = h->G; sa.sigma2 = h->sigma2; sa.plugin = plugin;
    sa.estimate_trend = h->estimate_trend; sa.minimize = minimize; sa.q = q;
    for (int i = 0; i < q; ++i) { sa.acq_id[i] = acq_id[i]; sa.acq_par[i] = acq_par ? acq_par[i] : 0.0; }
    sa.mu_out = want_out ? h->dmu_out : nullptr; sa.mse_out = want_out ? h->dmse_out : nullptr;
    sa.acq_out = want_acq_out ? h->dacq_out : nullptr;
    sa.blk_val = h->dblk_val; sa.blk_idx = h->dblk_idx; sa.nblk = nblk; sa.counter = h->dcounter;
    sa.best_val = h->dbest_val; sa.best_idx = h->dbest_idx;
    const bool stamps = getenv("BOGP_SMALL_STAMPS") && atoi(getenv("BOGP_SMALL_STAMPS"));
    if (stamps) {  // measurement aid: per-phase wave-cycles of this launch, printed on stderr
      if ((e2 = ensure(h, &h->dbatch, &h->batch_cap, (size_t)8))) return e2;
      HIPCHK(h, hipMemsetAsync(h->dbatch, 0, 8 * sizeof(double), st));
      sa.stamps = (long long*)h->dbatch;
    }
    hipEvent_t e0 = get_event(h, 0), e1 = get_event(h, 1);
    if (!e0 || !e1) FAIL(h, BOGP_ERR_HIP, "hipEventCreate failed");
    HIPCHK(h, hipEventRecord(e0, st));
    HIPCHK(h, launch_sweep_small(h->kernel, sa, h->n_cu, st));
    HIPCHK(h, hipEventRecord(e1, st));
    h->n_chunks = 1;
    h->timing_pending = true;
    h->timing_fused = true;
    if (sync || stamps) HIPCHK(h, hipStreamSynchronize(st));
    if (stamps) {
      collect_timing(h);
      long long sv[5] = {0, 0, 0, 0, 0};
      HIPCHK(h, hipMemcpy(sv, h->dbatch, sizeof(sv), hipMemcpyDeviceToHost));
      const double nw = (double)std::max<long long>(1, sv[4]);
      fprintf(stderr, "k_sweep_small M=%lld: %.3f ms; per wave: produce %.0f, contract %.0f, wait-at-barrier %.0f, epilogue %.0f cycles (%lld waves)\n",
              (long long)M, h->t_contract_ms, sv[0] / nw, sv[1] / nw, sv[2] / nw, sv[3] / nw, sv[4]);
    }
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
    if ((e = ensure(h, &h->drT[b], &h->rT_cap[b], (size_t)Nrows * Mc))) return e;
    if ((e = ensure(h, &h->dmu_part[b], &h->mu_part_cap[b], (size_t)S * Mc))) return e;
    if ((e = ensure(h, &h->dw_part[b], &h->w_part_cap[b], (size_t)S * Mc))) return e;
  }
  if ((e = ensure(h, &h->dss_part, &h->ss_part_cap, (size_t)nJ * Mc))) return e;
  if (h->p > 1) {
    if (Mc > 0x7fffffff / 2) FAIL(h, BOGP_ERR_UNSUPPORTED, "chunk of %lld candidates is too large for the trend GEMM (lower BOGP_CHUNK_MB)", (long long)Mc);
    if (!vx_model) {
      if ((e = ensure(h, &h->dTt, &h->Tt_cap, (size_t)Mc * ((h->p + 127) / 128 * 128)))) return e;  // whole 128-column tiles (k_mm128)
      if ((e = ensure(h, &h->dCS, &h->CS_cap, (size_t)Mc * ((h->p + 127) / 128 * 128)))) return e;
    }
    if ((e = ensure(h, &h->duu, &h->uu_cap, (size_t)Mc))) return e;
    if ((e = ensure(h, &h->dmtrend, &h->mtrend_cap, (size_t)Mc))) return e;
  }
  // a polynomial basis of at most 32 columns under universal kriging: T = W^T r is accumulated by the producer itself
  // (k_corr_chunk<K, PV>) and finished by ONE per-candidate launch (k_trend_small); BOGP_TREND_FUSED=0 keeps the tile products
  // (a mean-only call of a trend-rows model keeps the producer the full call uses -- pv = 0 -- so that mu comes out bit-identical)
  const int pv = (!vx_model && h->p > 1 && h->estimate_trend && !(getenv("BOGP_TREND_FUSED") && atoi(getenv("BOGP_TREND_FUSED")) == 0)) ? corr_trend_columns(h->p) : 0;
  if (pv > 0)
    for (int b = 0; b < nbuf; ++b)
      if ((e = ensure(h, &h->dtpart[b], &h->tpart_cap[b], (size_t)S * pv * Mc))) return e;
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
    ca.sqrt_theta = h->dsqrt_theta; ca.XthT = h->dXthT; ca.xnorm = h->dXnorm; ca.gamma = h->dgamma; ca.wvec = h->dw;
    ca.rT = h->drT[b]; ca.mu_part = h->dmu_part[b]; ca.w_part = h->dw_part[b];
    if (pv > 0) {
      ca.pv = pv; ca.Wrow = h->dWpT; ca.wld = (h->p + 127) / 128 * 128; ca.t_part = h->dtpart[b];
    }
    ContractArgs ka;
    ka.rT = h->drT[b]; ka.Vp = vx ? h->dVpx : h->dVp; ka.ss_part = h->dss_part; ka.Mc = Mc; ka.nMt = (int)(Mc_eff / 64); ka.nJ = nJ;
    ka.NJ16 = NJ16; ka.NKP = Nrows / 8;
    // producer: may reuse buffer b only after chunk c-2 (its previous user) is completely done
    if (overlap && c >= 2) HIPCHK(h, hipStreamWaitEvent(stP, h->ev[(size_t)((c - 2) * EPC + 4)], 0));
    // trend-rows models: k_trend_rows (below, on the producer stream) writes h->dmtrend, which is NOT double buffered -- chunk c - 1's
    // k_acquisition on the main stream must have read it first (without this wait chunk c - 1 could get chunk c's means)
    if (overlap && vx && c >= 1) HIPCHK(h, hipStreamWaitEvent(stP, h->ev[(size_t)((c - 1) * EPC + 4)], 0));
    if (h->hXs_lazy) {  // lazily uploaded candidates: this chunk's rows must have arrived (chunk 0: copied here; later ones: below)
      int el = lazy_copy_to(h, m0 + mcount);
      if (el) return el;
      if ((el = lazy_wait(h, stP))) return el;
      if (stP != st && (el = lazy_wait(h, st))) return el;
    }
    HIPCHK(h, hipEventRecord(ev[0], stP));
    HIPCHK(h, launch_corr_chunk(h->kernel, ca, (int)(Mc_eff / 64), S, stP));
    if (vx) {  // rows Np .. Ne - 1 = 0 (their columns of the factor are zero: any FINITE value would do), rows Ne .. = -f(x*), then zeros
      if (h->vx_Ne > Np) HIPCHK(h, hipMemsetAsync(h->drT[b] + (size_t)Np * Mc, 0, (size_t)(h->vx_Ne - Np) * Mc * sizeof(double), stP));
      HIPCHK(h, launch_trend_rows(h->trend, h->dXs, m0, mcount, Mc_eff, d, Mc, h->dbetav, h->drT[b] + (size_t)h->vx_Ne * Mc, h->p,
                                  h->vx_Nt - h->vx_Ne, h->dmtrend, stP));
    }
    HIPCHK(h, hipEventRecord(ev[1], stP));
    if (overlap) HIPCHK(h, hipStreamWaitEvent(st, ev[1], 0));
    HIPCHK(h, hipEventRecord(ev[2], st));
    // predict(X) without eval_MSE (gpr.py:486-491 returns before the triangular solve): the N^2 contraction is skipped
    // and k_acquisition sums zero variance groups (its MSE output is not read)
    if (need_var) HIPCHK(h, launch_contract(ka, st));
    HIPCHK(h, hipEventRecord(ev[3], st));
    AcqArgs aa;
    memset(&aa, 0, sizeof(aa));
    aa.mu_part = h->dmu_part[b]; aa.w_part = h->dw_part[b]; aa.ss_part = h->dss_part; aa.S = S; aa.nJ = need_var ? nJ_main : 0; aa.Mc = Mc;
    aa.nJ_plus = vx ? nJ - nJ_main : 0;
    aa.mcount = mcount; aa.m0 = m0; aa.beta = h->beta; aa.G = h->G; aa.estimate_trend = h->estimate_trend;
    aa.sigma2 = h->sigma2; aa.mu_out = want_out ? h->dmu_out : nullptr; aa.mse_out = want_out ? h->dmse_out : nullptr;
    aa.q = q;
    for (int i = 0; i < q; ++i) { aa.acq_id[i] = acq_id[i]; aa.acq_par[i] = acq_par ? acq_par[i] : 0.0; }
    aa.plugin = plugin; aa.minimize = minimize; aa.acq_out = want_acq_out ? h->dacq_out : nullptr; aa.M = M;
    aa.blk_val = h->dblk_val; aa.blk_idx = h->dblk_idx; aa.blk_offset = blk_offset; aa.nblk_total = nblk_total;
    if (h->p > 1) {
      // polynomial trend: mean f(x*) . beta, and under universal kriging u = G^-T (Ft^T L^-1 r - f(x*)) (gpr.py:496-498):
      // T = r W (Mc x p, a tile product on the chunk that k_contract has just read), c = T - f(x*), u^T u = c^T (Ft^T Ft)^-1 c
      // The two products run on k_mm128 (128 x 128 tiles, kernels_chol.hip) when the chunk is whole tiles -- the default
      // chunk sizes are; rows past Mc_eff of the last tile are computed on stale chunk data and never read -- and on the
      // generic k_gemm64 otherwise.
      const int pt = h->p, pp = (pt + 127) / 128 * 128;
      const bool tiles128 = Mc % 128 == 0 && !(getenv("BOGP_TREND_GEMM64") && atoi(getenv("BOGP_TREND_GEMM64")) != 0);
      const int TI = (int)((Mc_eff + 127) / 128);
      const double one = 1.0, zero = 0.0;
      double* Tt = nullptr;
      if (vx) {
        // (mtrend was written by k_trend_rows beside the chunk's extra rows; |u|^2 comes out of the contraction)
      } else if (pv > 0) {
        HIPCHK(h, launch_trend_small(h->trend, h->dXs, m0, mcount, d, Mc, h->dbetav, h->dtpart[b], S, pv, pt, h->dSinv, h->dmtrend, h->duu, st));
        aa.uu = h->duu;
      } else {
      if (h->estimate_trend && !vx_model) {
        if (tiles128)
          HIPCHK(h, launch_mm128_gen(h->drT[b], (int)Mc, h->dWpT, pp, h->dTt, (int)Mc, TI, pp / 128, Np, st));
        else
          HIPCHK(h, launch_gemm(0, 0, (int)Mc_eff, pt, Np, one, h->drT[b], (int)Mc, h->dWp, Np, zero, h->dTt, (int)Mc, st, 0, &h->gsplit));
        Tt = h->dTt;
      }
      HIPCHK(h, launch_trend_terms(h->trend, h->dXs, m0, mcount, d, Mc, h->dbetav, Tt, h->dmtrend, st));
      if (h->estimate_trend && !vx_model) {
        if (tiles128)
          HIPCHK(h, launch_mm128_gen(h->dTt, (int)Mc, h->dSinvP, pp, h->dCS, (int)Mc, TI, pp / 128, pp, st));
        else
          HIPCHK(h, launch_gemm(0, 0, (int)Mc_eff, pt, pt, one, h->dTt, (int)Mc, h->dSinv, pt, zero, h->dCS, (int)Mc, st, 0, &h->gsplit));
        HIPCHK(h, launch_rowdot(h->dTt, h->dCS, Mc, mcount, pt, h->duu, st));
        aa.uu = h->duu;
      }
      }  // pv == 0
      aa.mtrend = h->dmtrend;
      aa.estimate_trend = 0;  // the scalar w_part path is for the constant basis
    }
    HIPCHK(h, launch_acquisition(aa, st));
    HIPCHK(h, hipEventRecord(ev[4], st));
    blk_offset += (mcount + 255) / 256;
    if (h->hXs_lazy) {  // the next chunk's rows travel while this chunk's kernels (queued above) run
      const int el = lazy_copy_to(h, m0 + mcount + Mc);
      if (el) return el;
    }
  }
  if (h->hXs_lazy) {  // every row is on its way; once the copy stream is idle the caller's buffer is no longer needed
    const int el = lazy_finish(h);
    if (el) return el;
  }
  if (q > 0) HIPCHK(h, launch_argmax_final(h->dblk_val, h->dblk_idx, blk_offset, nblk_total, q, h->dbest_val, h->dbest_idx, st));
  h->n_chunks = (int)nchunk;
  h->timing_pending = true;
  h->timing_fused = false;
  if (sync || overlap) HIPCHK(h, hipStreamSynchronize(st));
  if (overlap) HIPCHK(h, hipStreamSynchronize(stP));
  return BOGP_OK;
}

extern "C" int bogp_predict(bogp_handle* h, double* mu, double* mse) {
  if (!h) return BOGP_ERR_INVALID;
  if (!mu) FAIL(h, BOGP_ERR_INVALID, "bogp_predict: mu must be non-null");
  int rc = run_sweep(h, true, 0, nullptr, nullptr, 0.0, 1, false, mse != nullptr);
  if (rc) return rc;
  HIPCHK(h, hipMemcpy(mu, h->dmu_out, (size_t)h->M * sizeof(double), hipMemcpyDeviceToHost));
  if (mse) HIPCHK(h, hipMemcpy(mse, h->dmse_out, (size_t)h->M * sizeof(double), hipMemcpyDeviceToHost));
  return BOGP_OK;
}

extern "C" int bogp_sweep(bogp_handle* h, int q, const int* acq_id, const double* acq_par, double plugin, int minimize,
                          double* best_val, int64_t* best_idx, double* acq_out) {
  if (!h) return BOGP_ERR_INVALID;
  if (q <= 0 || !acq_id || (!best_val != !best_idx)) FAIL(h, BOGP_ERR_INVALID, "bogp_sweep: q > 0, non-null acq_id, and best_val / best_idx both given or both NULL");
  const bool local = best_val != nullptr;  // NULL outputs: the winners stay on the device for bogp_exchange_argmax
  if (!local && acq_out) FAIL(h, BOGP_ERR_INVALID, "bogp_sweep: acq_out needs best_val / best_idx");
  for (int i = 0; i < q; ++i) {
    if (acq_id[i] < 0 || acq_id[i] > 3) FAIL(h, BOGP_ERR_INVALID, "unknown acquisition id %d", acq_id[i]);
    const bool zero_ok = acq_id[i] == BOGP_ACQ_EPSILON_PI;  // epsilon = 0 is plain PI
    if (acq_id[i] != BOGP_ACQ_EI && (!acq_par || !(acq_par[i] > 0 || (zero_ok && acq_par[i] == 0))))
      FAIL(h, BOGP_ERR_INVALID, "acquisition parameter %d must be > 0 (the reference asserts alpha/epsilon/t > 0)", i);
  }
  invalidate_sweep_results(h);
  int rc = run_sweep(h, false, q, acq_id, acq_par, plugin, minimize, acq_out != nullptr, true, local);
  if (rc) return rc;
  h->last_q = q;  // dbest_val / dbest_idx hold this sweep's winners for bogp_exchange_argmax
  if (!local) return BOGP_OK;  // queued, not waited for: the exchange that follows is ordered behind it on the stream
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
  invalidate_sweep_results(h);  // run_sweep below overwrites dbest_* as well
  int rc = run_sweep(h, false, q, acq_id, acq_par, plugin, minimize, true);  // keeps the q x M values on the device
  if (rc) return rc;
  // rank 0 is the sweep's own argmax; ranks 1..k-1 repeat the argmax with the winners so far masked out -- all q criteria
  // per launch, the winners kept on the device: 2 k queued launches and ONE read-back of q x k (value, index) pairs
  const int64_t M = h->M;
  const int64_t nblk = (M + 255) / 256;
  hipStream_t st = h->stream;
  int e;
  if ((e = ensure(h, &h->dblk_val, &h->blk_val_cap, (size_t)q * (nblk + 1)))) return e;
  if ((e = ensure(h, &h->dblk_idx, &h->blk_idx_cap, (size_t)q * (nblk + 1)))) return e;
  if ((e = ensure(h, &h->dtopk_val, &h->topk_val_cap, (size_t)BOGP_MAX_Q * BOGP_MAX_TOPK))) return e;
  if ((e = ensure(h, &h->dtopk_idx, &h->topk_idx_cap, (size_t)BOGP_MAX_Q * BOGP_MAX_TOPK))) return e;
  HIPCHK(h, launch_topk(h->dacq_out, M, q, k, h->dblk_val, h->dblk_idx, h->dtopk_val, h->dtopk_idx, st));
  HIPCHK(h, hipMemcpyAsync(best_val, h->dtopk_val, (size_t)q * k * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(best_idx, h->dtopk_idx, (size_t)q * k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  h->last_topk_q = q;
  h->last_topk_k = k;
  for (int i = 0; i < q * k; ++i)
    if (best_idx[i] == INT64_MAX) {  // fewer candidates than k: pad with (-inf, -1)
      best_val[i] = -INFINITY;
      best_idx[i] = -1;
    }
  return BOGP_OK;
}

extern "C" int bogp_last_timing(bogp_handle* h, double* corr_ms, double* contract_ms, double* acquisition_ms, int* n_chunks) {
  if (!h) return BOGP_ERR_INVALID;
  collect_timing(h);
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
  if (h->kernel == BOGP_KERNEL_CUBIC || h->kernel == BOGP_KERNEL_GENEXP || h->kernel == BOGP_KERNEL_MATERN_NU) FAIL(h, BOGP_ERR_UNSUPPORTED, "bogp_gradient: the cubic correlation has no input-derivative (corr_dx leaves it undefined in the reference, gpr.py:655-658)");
  if (!x || !dmu || !dmse) FAIL(h, BOGP_ERR_INVALID, "bogp_gradient: null pointer");
  const int N = h->N, d = h->d;
  hipStream_t st = h->stream;
  HIPCHK(h, hipSetDevice(h->device));
  const int pt = h->p;
  if (pt == 1)  // constant basis: k_point_rhs + k_point_tri, no library call (kernels_point.hip)
    return point_eval_host(h, "bogp_gradient", x, 1, 0, nullptr, nullptr, 0.0, 1, nullptr, nullptr, dmu, dmse, nullptr, nullptr);
  if (pt > 1 && h->trend == BOGP_TREND_QUADRATIC)
    FAIL(h, BOGP_ERR_UNSUPPORTED, "bogp_gradient: the quadratic trend has no Jacobian in the reference either (trend.py:138-139)");
  int e = ensure(h, &h->dgrad_partial, &h->grad_partial_cap, (size_t)N * (d + 3) + 4 * d + 8 + (size_t)pt * (d + 1));
  if (e) return e;
  double* dr = h->dgrad_partial;            // N
  double* drdx = dr + N;                    // d x N (column k = dr/dx_k); [r | dr/dx] is one N x (d+1) column-major matrix
  double* dz = drdx + (size_t)N * d;        // N
  double* dx = dz + N;                      // d
  double* dout = dx + d;                    // 3 d
  double* dtw = dout + 3 * d + 8;           // p x (d+1): W^T [r | dr/dx]
  double* dvr = dtw + (size_t)pt * (d + 1);  // N: V r
  HIPCHK(h, hipMemcpyAsync(dx, x, d * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(h, launch_point_corr(h->kernel, h->dX, N, d, h->dtheta, dx, dr, drdx, st));
  // z = L^-T L^-1 r = V^T (V r) with the explicit V = L^-1 kept from the commit: two triangular matrix-vector
  // products (bandwidth bound, ~50 us at N = 2048) instead of two dependent triangular solves (~350 us each)
  HIPCHK(h, launch_gemm(0, 0, N, 1, N, 1.0, h->dV, h->ldr, dr, N, 0.0, dvr, N, st, 1));  // V r
  HIPCHK(h, launch_gemm(1, 0, N, 1, N, 1.0, h->dV, h->ldr, dvr, N, 0.0, dz, N, st, 2));  // V^T (V r)
  const double one = 1.0, zero = 0.0;
  HIPCHK(h, launch_gemm(1, 0, d, 1, N, one, drdx, N, h->dgamma, N, zero, dout, d, st, 0, &h->gsplit));
  HIPCHK(h, launch_gemm(1, 0, d, 1, N, one, drdx, N, dz, N, zero, dout + d, d, st, 0, &h->gsplit));
  std::vector<double> out(3 * d, 0.0), tw;
  if (h->estimate_trend && pt > 1) {  // (Ft^T L^-1) [r | dr/dx] = W^T [r | dr/dx]   (gpr.py:570-571)
    HIPCHK(h, launch_gemm(1, 0, pt, d + 1, N, one, h->dWp, h->Np, dr, N, zero, dtw, pt, st, 0, &h->gsplit));
    tw.resize((size_t)pt * (d + 1));
    HIPCHK(h, hipMemcpyAsync(tw.data(), dtw, tw.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  HIPCHK(h, hipMemcpyAsync(out.data(), dout, 2 * d * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  {  // linear basis (the constant one returned above): f = [1, x], Jacobian rows 1..d = identity (trend.py:104-112)
    std::vector<double> su;  // S u with u = Ft^T rt - f and S = (Ft^T Ft)^-1 (:570-573)
    if (h->estimate_trend) {
      std::vector<double> u(pt);
      for (int c = 0; c < pt; ++c) u[c] = tw[c] - (c == 0 ? 1.0 : x[c - 1]);
      su.assign(pt, 0.0);
      for (int c = 0; c < pt; ++c)
        for (int r = 0; r < pt; ++r) su[r] += h->h_Sinv[(size_t)c * pt + r] * u[c];  // S symmetric, column-major
    }
    for (int k = 0; k < d; ++k) {
      dmu[k] = h->h_betav[1 + k] + out[k];  // beta^T f_dx + gamma^T r_dx (:561)
      double m = -1.0 * out[d + k];
      if (h->estimate_trend) {
        double acc = 0.0;
        for (int c = 0; c < pt; ++c) acc += su[c] * (tw[(size_t)(1 + k) * pt + c] - (c == 1 + k ? 1.0 : 0.0));  // u_dx = Ft^T rt_dx - f_dx
        m += acc;
      }
      dmse[k] = 2.0 * h->sigma2 * m;
    }
  }
  return BOGP_OK;
}

// Hessian of the posterior mean at x (GaussianProcess.Hessian, gpr.py:578-598): f_dx2 . beta + r_dx2 . gamma.  The trend
// part is zero for the constant and linear bases (trend.py:88-91, 113-116; the quadratic one raises); the correlation part
// exists for the squared exponential only (corr_Hessian, :663-734, leaves H undefined for every other kernel).
extern "C" int bogp_hessian(bogp_handle* h, const double* x, double* H) {
  if (!h) return BOGP_ERR_INVALID;
  if (!h->committed) FAIL(h, BOGP_ERR_INVALID, "bogp_hessian: no committed model");
  if (!x || !H) FAIL(h, BOGP_ERR_INVALID, "bogp_hessian: null pointer");
  if (h->kernel != BOGP_KERNEL_SE) FAIL(h, BOGP_ERR_UNSUPPORTED, "bogp_hessian: squared exponential only (the reference's corr_Hessian defines no other kernel)");
  if (h->trend == BOGP_TREND_QUADRATIC) FAIL(h, BOGP_ERR_UNSUPPORTED, "bogp_hessian: the quadratic trend has no Hessian in the reference (trend.py:141-142)");
  const int N = h->N, d = h->d;
  hipStream_t st = h->stream;
  HIPCHK(h, hipSetDevice(h->device));
  int e = ensure(h, &h->dgrad_partial, &h->grad_partial_cap, (size_t)N * (d + 1) + d + (size_t)d * d);
  if (e) return e;
  double* dr = h->dgrad_partial;
  double* drdx = dr + N;
  double* dx = drdx + (size_t)N * d;
  double* dH = dx + d;
  HIPCHK(h, hipMemcpyAsync(dx, x, d * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(h, launch_point_corr(h->kernel, h->dX, N, d, h->dtheta, dx, dr, drdx, st));
  HIPCHK(h, launch_point_hessian(h->dX, N, d, h->dtheta, dx, dr, drdx, h->dgamma, dH, st));
  HIPCHK(h, hipMemcpyAsync(H, dH, (size_t)d * d * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  return BOGP_OK;
}

// Correlation between the rows of X1 at the committed theta (GaussianProcess.prior_cov(X1, corr=True), gpr.py:318-353;
// its X2 argument cannot be used in the reference: `if X2` on an array raises).  R is n1 x n1, row-major.
extern "C" int bogp_prior_corr(bogp_handle* h, const double* X1, int n1, double* R) {
  if (!h) return BOGP_ERR_INVALID;
  if (!h->committed) FAIL(h, BOGP_ERR_INVALID, "bogp_prior_corr: no committed model");
  if (!X1 || !R || n1 <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_prior_corr: X1 / R must be non-null and n1 > 0");
  const int d = h->d;
  hipStream_t st = h->stream;
  HIPCHK(h, hipSetDevice(h->device));
  int e = ensure(h, &h->dbatch, &h->batch_cap, (size_t)n1 * d + 2 * (size_t)n1 * n1);
  if (e) return e;
  double* dX1 = h->dbatch;
  double* dr = dX1 + (size_t)n1 * d;
  double* ds2 = dr + (size_t)n1 * n1;
  HIPCHK(h, hipMemcpyAsync(dX1, X1, (size_t)n1 * d * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(h, launch_batch_corr(h->kernel, dX1, n1, d, h->dtheta, dX1, n1, dr, ds2, st));
  HIPCHK(h, hipMemcpyAsync(R, dr, (size_t)n1 * n1 * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  return BOGP_OK;
}

// ------------------------------------------------------------------------------------------------------
// self test of kernels_gemm.hip on host buffers (include/bogp.h)
// ------------------------------------------------------------------------------------------------------
#ifdef NS_PROFILE
// (profiling builds only, `make EXTRA=-DNS_PROFILE`: the 64 scalars of the last polled read-back / of the device block, incl. the
// clock words a profiled kernel leaves -- tools/prof_nll_small_phases.py)
extern "C" int bogp_debug_fit_scalars(bogp_handle* h, double* out) {
  if (!h || !out) return BOGP_ERR_INVALID;
  memcpy(out, h->hfit + 2048, 64 * sizeof(double));
  return BOGP_OK;
}

extern "C" int bogp_debug_dscal(bogp_handle* h, double* out) {
  if (!h || !out) return BOGP_ERR_INVALID;
  HIPCHK(h, hipMemcpy(out, h->dscal, 64 * sizeof(double), hipMemcpyDeviceToHost));
  return BOGP_OK;
}
#endif

#ifdef ELIM_PROFILE
// (profiling builds only, `make EXTRA=-DELIM_PROFILE`: the wall-clock stamps the fused elimination step leaves -- tools/probes/elim_stamps.py)
namespace bogp { hipError_t debug_elim_stamps(unsigned long long* out); }
extern "C" int bogp_debug_elim_stamps(unsigned long long* out) { return bogp::debug_elim_stamps(out) == hipSuccess ? BOGP_OK : BOGP_ERR_HIP; }
#endif

#ifdef CONTRACT_TRACE
// (profiling builds only, `make EXTRA=-DCONTRACT_TRACE`: the stamps the last k_contract16<4> launch left -- tools/contract_trace.py)
namespace bogp { hipError_t debug_contract_trace(unsigned long long* out, size_t cap_words, size_t* used_words, int* dims); }
extern "C" int bogp_debug_contract_trace(unsigned long long* out, size_t cap_words, size_t* used_words, int* dims) {
  return bogp::debug_contract_trace(out, cap_words, used_words, dims) == hipSuccess ? BOGP_OK : BOGP_ERR_HIP;
}
#endif

extern "C" int bogp_selftest_gemm(bogp_handle* h, int ta, int tb, int m, int n, int k, double alpha, const double* A, int lda,
                                  const double* B, int ldb, double beta, double* C, int ldc, int tri, int split) {
  if (!h) return BOGP_ERR_INVALID;
  if (!A || !B || !C || m <= 0 || n <= 0 || k <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_selftest_gemm: null pointer or empty shape");
  if (lda < (ta ? k : m) || ldb < (tb ? n : k) || ldc < m) FAIL(h, BOGP_ERR_INVALID, "bogp_selftest_gemm: leading dimension below the stored rows");
  if (tri && m != k) FAIL(h, BOGP_ERR_INVALID, "bogp_selftest_gemm: a triangular op(A) is square");
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const size_t na = (size_t)lda * (ta ? m : k), nb = (size_t)ldb * (tb ? k : n), nc = (size_t)ldc * n;
  double *dA = nullptr, *dB = nullptr, *dC = nullptr;
  int rc = BOGP_OK;
  if (split && (rc = ensure_gsplit(h))) return rc;
  if (hipMalloc((void**)&dA, na * sizeof(double)) != hipSuccess || hipMalloc((void**)&dB, nb * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&dC, nc * sizeof(double)) != hipSuccess) {
    dfree(dA); dfree(dB); dfree(dC);
    FAIL(h, BOGP_ERR_HIP, "bogp_selftest_gemm: hipMalloc failed");
  }
  hipError_t e = hipMemcpyAsync(dA, A, na * sizeof(double), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(dB, B, nb * sizeof(double), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(dC, C, nc * sizeof(double), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = launch_gemm(ta, tb, m, n, k, alpha, dA, lda, dB, ldb, beta, dC, ldc, st, tri, split ? &h->gsplit : nullptr);
  if (e == hipSuccess) e = hipMemcpyAsync(C, dC, nc * sizeof(double), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  dfree(dA); dfree(dB); dfree(dC);
  if (e != hipSuccess) FAIL(h, BOGP_ERR_HIP, "bogp_selftest_gemm: %s", hipGetErrorString(e));
  return BOGP_OK;
}
