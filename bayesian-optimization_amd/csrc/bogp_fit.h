// bogp_fit.h -- host-side pieces of the likelihood path shared by bogp_api.hip (bogp_nll, bogp_commit) and bogp_batch.hip
// (bogp_nll_batch, bogp_mle_batch): what the device's scalars are turned into (gpr.py:931-1038).
#pragma once
#include "bogp_handle.h"

namespace bogp {

struct FitOut {
  double llf = 0, sigma2 = 0, noise_var = 0, s2t = 0, G = 0, beta = 0, ftyt = 0, ftft = 0, logdet = 0, rho_ss = 0;
  // per target (n_t > 1: llf above is the SUM over targets, gpr.py:1040; sigma2 / s2t / rho_ss above are target 0's)
  double sigma2_t[BOGP_MAX_TARGETS] = {0}, s2t_t[BOGP_MAX_TARGETS] = {0}, nv_t[BOGP_MAX_TARGETS] = {0};
};
struct FitPending {
  int mode = 0, estimate_trend = 0, ptrend = 1, n_t = 1, N = 0;
  double beta = 0, alpha = 0, sigma2_par = 0, noise_var = 0, s2t = 0;
};

int trend_size(int trend, int d);
// polls the sequence word at `flag_word` (host address of device-mapped pinned memory) for `seq`; bounded, then a stream synchronisation
int fit_wait_on(bogp_handle* h, const void* flag_word, unsigned long long seq);
// llf, sigma2, ... of one evaluation from its scalars `sc` ([0] sum(log diag L), [1] |Ft|, [2] Ft.Yt, [3] rho.rho, 4 per target) and
// the factorisation's info word; sets h->err and returns BOGP_ERR_NOT_POSDEF / BOGP_ERR_LLF_POSITIVE where the reference gives -inf
int factorize_finish(bogp_handle* h, const FitPending& fp, int info, const double* sc, const int* info2, bool reject_positive, FitOut* o);
// d llf / d par from the d + 1 contractions S[0 .. d], trace(R^-1) S[d + 1] and gamma.gamma S[d + 2] (gpr.py:1001-1038)
void nll_gradient_from_sums(int mode, bool iso, int d, const double* par, int n_par, int n_t, const double* S, double s2t, double* grad);

// helpers shared by the three parts of the C ABI (bogp_api.hip defines them)
void select_target(bogp_handle* h, int t);
int ensure_gsplit(bogp_handle* h);
int trend_solve(bogp_handle* h, int trend, int estimate_trend);
// polynomial bases with p > 32 columns take the trend-rows path (r05; against the r02-r04 tile products: profiles/r05_trend_timing.txt);
// p <= 32 stays fused into kernel A
inline bool trend_rows_enabled() { return true; }
inline int trend_rows_min() { return 33; }

}  // namespace bogp
