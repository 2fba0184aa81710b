// bogp_api_fit.hip -- the C ABI of libbogp.so (include/bogp.h), part 2 of 3: the fit path -- factorisation at a parameter vector, the
// concentrated and the restricted likelihood with their gradients, commit, and the committed state (orchestration of kernels_chol /
// kernels_fit / kernels_pairs / kernels_gemm / kernels_nllsmall).  Part 1 = bogp_api.hip, part 3 = bogp_api_sweep.hip.
#include <hip/hip_runtime.h>

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
// Bounded: after ~2 ms of polling the ordinary synchronisation takes over.
static int fit_readback(bogp_handle* h, const double* dS, int nS, double* blk /* 64 */, double* S_out) {
  hipStream_t st = h->stream;
  if (nS > 512) {
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
extern "C" int bogp_chol_wide_panels(int N, int* widths, int cap) {
  if (N <= 0) return 0;
  const int ld = N > 3072 ? ((N + 127) / 128) * 128 : ((N + 63) / 64) * 64;  // bogp_set_train's leading dimension
  return chol_wide_panels(ld, widths, cap < 0 ? 0 : cap);
}

extern "C" int bogp_nll_path(int N, int d, int trend, int n_targets) {
  if (N <= 0 || d <= 0 || trend != BOGP_TREND_CONSTANT || n_targets != 1) return BOGP_NLL_PATH_GENERAL;
  if (getenv("BOGP_NLL_FUSED") && atoi(getenv("BOGP_NLL_FUSED")) == 0) return BOGP_NLL_PATH_GENERAL;
  if (nll_small_fits(N, d)) return BOGP_NLL_PATH_ONE_LAUNCH;
  // 157 <= N <= 3072 (r05: 2048 -> 3072 after the step lost a third of its time -- llf + gradient 1.73 -> 0.95 ms at N = 2112,
  // 2.78 -> 2.28 at 3072, a slot of a batch of ten 1.61 -> 0.93 ms; it loses from ~3500 on: profiles/r05_elim_chain.txt): factor + inverse + solves as one
  // elimination at 64-block granularity (kernels_chol.hip: k_elim_step), one launch a block column; BOGP_NLL_ELIM=0 keeps the
  // Cholesky / recursive-doubling / U U^T kernels
  constexpr int elim_max = 3072;
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
  const bool deferred = grad != nullptr && trend == BOGP_TREND_CONSTANT;
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
  if (deferred && n_t == 1 && d + 3 <= 512) {
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
  // commit only, every trend basis (b = y - F beta with the committed coefficients).
  {
    constexpr int steps = 1;  // (0 / 2 steps: tools/refine_experiment.py, r04 -- one step is what the accuracy tests were fixed with)
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
