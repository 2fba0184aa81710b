// bogp_api.hip -- the C ABI of libbogp.so (include/bogp.h), part 1 of 3: handles, the training set, trend bases (device state).
// Part 2 = bogp_api_fit.hip (likelihood, commit, committed state), part 3 = bogp_api_sweep.hip (candidates, the chunked posterior /
// acquisition sweep, one-point calls).  In-tree kernels only -- no rocSOLVER, no rocBLAS.  No host fallback exists: every numerical step
// runs on the gfx950 device, and every failure is reported as an error code + message.  (One file until r05; split in r06.)
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
  // (r04 experiment, removed in r06: the look-ahead update of the two-level factorisation on a CU-masked stream -- tools/ab/ab_big_chol_cumask.sh,
  // EXPERIMENTS.md; no gain.  h->stream_upd stays null: the update runs on stream2.)
  if (hipEventCreateWithFlags(&h->ev_chol[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_chol[1], hipEventDisableTiming) != hipSuccess) {
    g_create_error = "event creation failed";
    delete h;
    return BOGP_ERR_HIP;
  }
  *out = h;
  return BOGP_OK;
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
void bogp::select_target(bogp_handle* h, int t) {
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
  // leading dimension: N rounded up to 64; to 128 above N = 3072 (r06; 6080 before), where the inverse, R^-1 and the Cholesky's wide panels work on 128 x 128 tiles
  const int ld_need = N > 3072 ? ((N + 127) / 128) * 128 : ((N + 63) / 64) * 64;  // (above the elimination's limit the fit runs on 128 x 128 tiles: kernels_chol.hip, BIG_LD)
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
int bogp::ensure_gsplit(bogp_handle* h) {
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
int bogp::trend_solve(bogp_handle* h, int trend, int estimate_trend) {
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
