/* bogp.h -- C ABI of libbogp.so: the MI355X (gfx950) GP-surrogate + batch-acquisition engine.
 *
 * The reference (wangronin/Bayesian-Optimization, `bayes_optim` 0.3.0) is pure Python: it has no FFI for
 * this path.  The drop-in boundary is three Python duck-typed protocols; the entry
 * points below are what a binding for those protocols needs, and each one names the reference code it
 * replaces (paths relative to bayes_optim/).  `INTEGRATION.md` shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every host array is C-contiguous float64 / int32 / int64
 *   - return value: 0 = BOGP_OK, < 0 = error code (never throws across the ABI);
 *     `bogp_last_error(h)` gives a human-readable message for the last failing call on that handle
 *   - BOGP_ERR_NOT_POSDEF is the "Cholesky failed" outcome (LAPACK-style info > 0); the Python host maps it to
 *     llf = -inf exactly like the reference maps LinAlgError (surrogate/gaussian_process/gpr.py:946-947,
 *     960-961, 978-979)
 *   - the library owns all device memory; the caller owns all host buffers
 *   - one handle per device; calls on one handle must be serialised by the caller (ctypes releases the GIL)
 *   - all arithmetic is IEEE float64 on the device; there is NO host/CPU fallback path
 */
#ifndef BOGP_H
#define BOGP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bogp_handle bogp_handle;

/* error codes */
#define BOGP_ABI_VERSION 9 /* what bogp_abi_version() of a matching library returns */
#define BOGP_OK 0
#define BOGP_ERR_INVALID (-1)      /* bad argument / call order                                           */
#define BOGP_ERR_HIP (-2)          /* HIP runtime failure (or an in-kernel hand-over that timed out)       */
#define BOGP_ERR_NOT_POSDEF (-3)   /* correlation matrix not positive definite (potrf info > 0)            */
#define BOGP_ERR_UNSUPPORTED (-4)  /* valid in the reference but not built yet (see DESIGN.md "out of scope") */
#define BOGP_ERR_NO_DEVICE (-5)    /* no usable gfx950 device                                              */
#define BOGP_ERR_LLF_POSITIVE (-6) /* llf > 0: the reference rejects it as -inf (gpr.py:981-982)           */

/* correlation functions: surrogate/gaussian_process/kernel.py
 *   SE        squared_exponential :289-329   exp(-sum_k theta_k d_k^2)
 *   MATERN12  matern nu=0.5       :190       exp(-s),                     s = sqrt(sum_k theta_k d_k^2)
 *   MATERN32  matern nu=1.5       :193-196   (1+t) exp(-t),               t = sqrt(3) s   [corr="matern"]
 *   MATERN52  matern nu=2.5       :198-200   (1+t+t^2/3) exp(-t),         t = sqrt(5) s                    */
#define BOGP_KERNEL_SE 0
#define BOGP_KERNEL_MATERN12 1
#define BOGP_KERNEL_MATERN32 2
#define BOGP_KERNEL_MATERN52 3
#define BOGP_KERNEL_ABSEXP 4 /* absolute_exponential :247-286   exp(-sum_k theta_k |d_k|) */
#define BOGP_KERNEL_GENEXP 6 /* exp(-sum_k theta_k |d_k|^p), kernel.py:332-379; theta = [theta_1 .. theta_d, p] (d + 1 entries, or [theta, p]):
                                values only, like BOGP_KERNEL_CUBIC */
#define BOGP_KERNEL_MATERN_NU 7 /* matern with any nu > 0, kernel.py:201-207: (2^(1-nu) / Gamma(nu)) t^nu K_nu(t), t = sqrt(2 nu) s (scipy.special.kv there, a device
                                   K_nu here); theta = [theta_1 .. theta_d, nu] (or [theta, nu]); values only -- corr_grad_theta / corr_dx define nothing for it */
#define BOGP_KERNEL_CUBIC 5 /* prod_k max(0, 1 - 3 (theta_k d_k)^2 + 2 (theta_k d_k)^3), kernel.py:419-466: likelihood VALUE, commit,
                               predict, sweep -- no derivatives, like the reference (corr_grad_theta / corr_dx leave it undefined) */

/* estimation modes: gpr.py:252-263; parameter layouts gpr.py:1073-1086
 *   NOISELESS   par = [theta]          sigma2 = sum(rho^2)/(N-k)
 *   NOISY       par = [theta, sigma2]  R = (sigma2 R0 + noise_var I)/(sigma2 + noise_var)
 *   NOISE_ESTIM par = [theta, alpha]   R = alpha R0 + (1-alpha) I                                        */
#define BOGP_MODE_NOISELESS 0
#define BOGP_MODE_NOISY 1
#define BOGP_MODE_NOISE_ESTIM 2

/* acquisition functions: acquisition/acquisition_fun.py (EI :150-189, EpsilonPI/PI :192-235, UCB :107-147,
 * MGFI :238-310).  `acq_par` is unused for EI, epsilon for EPSILON_PI, alpha for UCB, t for MGFI.          */
#define BOGP_ACQ_EI 0
#define BOGP_ACQ_EPSILON_PI 1
#define BOGP_ACQ_UCB 2
#define BOGP_ACQ_MGFI 3

/* trend (prior mean) bases: surrogate/gaussian_process/trend.py; all three are built (fit, predict, sweep) */
#define BOGP_TREND_CONSTANT 0
#define BOGP_TREND_LINEAR 1    /* [1, x]                      trend.py:94-118  */
#define BOGP_TREND_QUADRATIC 2 /* [1, x, x_k x_j (j >= k)]    trend.py:121-142 */

#define BOGP_MAX_DIM 320   /* input dimensions d (64 x d doubles of LDS in the sweep's producer) */
#define BOGP_MAX_TARGETS 8 /* columns of y (n_targets, gpr.py:463) */
#define BOGP_MAX_Q 64    /* criteria evaluated in one sweep (ParallelBO batch size q) */
#define BOGP_MAX_TOPK 32 /* ranks returned per criterion by bogp_sweep_topk */
#define BOGP_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */

/* ---- lifetime ------------------------------------------------------------------------------------- */
int bogp_create(int device, bogp_handle** out);
void bogp_destroy(bogp_handle* h);
const char* bogp_last_error(const bogp_handle* h); /* h may be NULL: message of the last failed bogp_create */
int bogp_abi_version(void);                        /* bumps whenever a signature below changes            */

/* ---- training set --------------------------------------------------------------------------------
 * Replaces GaussianProcess._check_data (gpr.py:279-310): X (N x d, row-major), y (N x n_targets).
 * The pair-distance list D of the reference (gpr.py:48-61, 13.4 GB at N=8192,d=50) is never built.
 * 1 <= n_targets <= BOGP_MAX_TARGETS.  With several targets (gpr.py:463, 490, 502-505, 931-1040) the correlation
 * model and its factorisation are shared and Yt / rho / gamma / sigma2 exist once per target: bogp_nll returns the SUM
 * of the per-target likelihoods and of their gradients exactly as the reference forms them, bogp_commit commits all
 * targets, and every per-target consumer (bogp_get_state, bogp_predict, bogp_sweep*, bogp_gradient*) works on the
 * target chosen with bogp_select_target (0 after set_train / commit).  As in the reference, only a FIXED constant
 * trend works with n_targets > 1 (estimating it raises at gpr.py:787); REML is single-target.                  */
int bogp_set_train(bogp_handle* h, const double* X, const double* y, int N, int d, int n_targets);
int bogp_select_target(bogp_handle* h, int target);

/* ---- likelihood -----------------------------------------------------------------------------------
 * Replaces GaussianProcess.log_likelihood_concentrated(par, eval_grad) (gpr.py:920-1040):
 * correlation_matrix (:772-782) -> _compute_aux_var (:790-811: potrf, L^-1 y, trend QR, rho) -> llf ->
 * gradient (:994-1038: gamma, R^-1 = L^-T L^-1, corr_grad_theta contraction without the (N,N,d) tensor).
 *   par           [theta (n_theta = d or 1), then sigma2 | alpha per mode], NOT log10
 *   noise_var     nugget tau^2 (NOISY mode); ignored otherwise
 *   estimate_trend 1: beta is GLS-estimated (ordinary kriging);  0: fixed `beta` (simple kriging)
 *   llf           out, scalar
 *   grad          out, n_par doubles, d llf / d par (the un-scaled gradient the reference hands L-BFGS-B,
 *                 SURVEY 8a quirks); NULL to skip the gradient (saves R^-1 + contraction)
 * Returns BOGP_ERR_NOT_POSDEF / BOGP_ERR_LLF_POSITIVE where the reference returns -inf.                 */
int bogp_nll(bogp_handle* h, int kernel, int mode, const double* par, int n_par, double noise_var, int trend,
             int estimate_trend, double beta, double* llf, double* grad);

/* P likelihood (+ gradient) evaluations in ONE device round trip: the restarts of GaussianProcess._optimize_hyperparameter
 * (gpr.py:1127-1162) are independent L-BFGS-B runs whose objective calls (gpr.py:1113-1123 -> log_likelihood_concentrated) can be
 * evaluated together -- at the sizes of a BO run one evaluation occupies one workgroup of the 256 CUs.
 *   par   P x n_par row-major (row s: the parameter vector of slot s, layout as bogp_nll);  llf (P);  grad (P x n_par) or NULL
 *   info  P per-slot outcomes: BOGP_OK, BOGP_ERR_NOT_POSDEF, BOGP_ERR_LLF_POSITIVE (llf[s] then holds the finite value), or
 *         BOGP_ERR_INVALID (a non-finite / non-positive parameter: that slot is skipped, the others are evaluated); llf[s] is NaN
 *         and grad row s zero for a failed slot
 * Slot s returns exactly the bits the s-th of P sequential bogp_nll calls returns.  The return value is BOGP_OK when the batch ran
 * (whatever the slots' outcomes) and an error code for what stops a bogp_nll call before the device (bad ids, no training set, HIP).
 * N <= 156: one launch, one workgroup a slot; N <= 3072: the elimination kernels over P workspaces (2 ld^2 doubles a slot, groups
 * bounded by BOGP_BATCH_MAX_MB, default 8192); above, and for polynomial trends / several targets: the P sequential calls,
 * dealt over up to three handles of the library's own (the caller's + helpers with their own stream and factor buffers, one host
 * thread each; two above N = 4096; BOGP_NLL_WORKERS) so that independent evaluations interleave on the device -- C5: 15.4 -> 13.4 ms
 * per evaluation, a linear-trend model at N = 2048: 1.86 -> 0.98 ms.  The two batched paths leave the handle's factor buffers -- a
 * committed model -- untouched.                                                                                                */
int bogp_nll_batch(bogp_handle* h, int kernel, int mode, int P, const double* par, int n_par, double noise_var, int trend,
                   int estimate_trend, double beta, double* llf, double* grad, int* info);

/* ---- the restarts of the MLE in lock step --------------------------------------------------------------
 * Replaces the restart loop of GaussianProcess._optimize_hyperparameter (gpr.py:1127-1162: `random_start` runs of
 * scipy.optimize.fmin_l_bfgs_b over log10(parameters), one after the other) by R bound-constrained L-BFGS runs that advance
 * TOGETHER: every round evaluates the current trial point of each active run with one bogp_nll_batch call.  Kept from the reference:
 * the log10 search space and bounds (:1088-1090), the objective -llf(10^x) with the UN-scaled gradient -d llf / d par (:1113-1123),
 * +inf / zero gradient where the likelihood is rejected, fmin_l_bfgs_b's defaults (m = 10, factr = 1e7, pgtol = 1e-5) and its rule
 * that budgets are tested when an iterate is accepted, never inside a line search.  Relaxed (hence opt-in on the Python side): the
 * evaluation budget is shared by runs that all start, so `wait_iter` has nothing to count.
 *   restricted  0: concentrated likelihood (bogp_nll_batch);  1: REML (bogp_nll_restricted per slot; par layout gpr.py:826-834)
 *   x0          R x n_par starting points in log10 space (clipped into the bounds);  lo, hi: n_par log10 bounds
 *   eval_budget evaluations allowed in total over all runs (<= 0: 15000 per run)
 *   m, factr, pgtol  <= 0: the defaults above
 *   prune_reserve  0: every run keeps going until the shared budget is spent.  > 0: while fewer than this many evaluations per active
 *               run are left, the run with the worst value so far is stopped (status 2, its last iterate returned): the budget
 *               is concentrated on the leading runs as it runs out, where equal shares would leave all R runs unconverged
 *   flags       0: the reference's gradient.  BOGP_MLE_CHAIN_RULE: hand the optimiser d(-llf) / d log10(par) = ln(10) par d / d par,
 *               the gradient of the function it actually minimises -- NOT what the reference does (an extension: ~4 x fewer
 *               evaluations per run from a start inside a basin, an immediate stop on the flat plateau of huge theta)
 *   xopt (R x n_par, log10), fopt (R: -llf at xopt, +inf if a run never saw a finite value), n_evals (R), status (R: 0 projected
 *   gradient <= pgtol, 1 relative reduction <= factr eps, 2 budget, 3 iterations, 4 line search failed, 5 start not finite),
 *   n_rounds: batched device calls made; the last three may be NULL.                                                        */
#define BOGP_MLE_CHAIN_RULE 1 /* flags bit 0 */
int bogp_mle_batch(bogp_handle* h, int kernel, int mode, int restricted, int R, const double* x0, int n_par, const double* lo,
                   const double* hi, double noise_var, int trend, int estimate_trend, double beta, int eval_budget, int m, double factr,
                   double pgtol, int flags, int prune_reserve, double* xopt, double* fopt, int* n_evals, int* status, int* n_rounds);

/* ---- commit a fitted state ------------------------------------------------------------------------
 * Replaces the tail of GaussianProcess.fit (gpr.py:402-415) + compute_beta_gamma (:784-788): factorise at
 * the final parameters and keep L, V = L^-1 (packed for the MFMA sweep), gamma, beta, w = L^-T Ft on the
 * device.  Same arguments as bogp_nll.                                                                  */
int bogp_commit(bogp_handle* h, int kernel, int mode, const double* par, int n_par, double noise_var, int trend,
                int estimate_trend, double beta, double* llf);

/* Read the committed state back (so the Python attributes C, gamma, rho, Yt, Ft, G, Q, beta, sigma2 stay
 * populated and the object stays picklable, base.py:499-540).  Any pointer may be NULL.
 *   C (N x N row-major lower Cholesky factor, strict upper = 0), gamma/rho/Yt/Ft/Q (N), G, beta, sigma2,
 *   noise_var (scalars).                                                                                */
int bogp_get_state(bogp_handle* h, double* C, double* gamma, double* rho, double* Yt, double* Ft, double* Q,
                   double* G, double* beta, double* sigma2, double* noise_var);

/* ---- restricted (REML) likelihood ---------------------------------------------------------------------
 * Replaces GaussianProcess.log_likelihood_restricted(par, eval_grad) (gpr.py:813-918).  Parameter layout (:826-834):
 * NOISELESS [theta, sigma2]; NOISY [theta, sigma2] with the fixed `noise_var`; NOISE_ESTIM [theta, sigma2, noise_var].
 * All three trend bases (constant, linear, quadratic; estimated or fixed coefficients) with one target; with several targets (fixed
 * constant trend only, as everywhere) the VALUE the reference's arithmetic yields -- the scalar terms broadcast over the n_t x n_t matrix
 * rho^T rho and everything summed (:861-866) -- and BOGP_ERR_UNSUPPORTED for its gradient (ValueError there, :875, :896).  Quirks kept: the simple-kriging value subtracts the log-determinant term (:861-866);
 * exp(llf) > 1 is rejected (:868-871): BOGP_ERR_LLF_POSITIVE, with *llf = the finite value and grad (if requested)
 * filled as the reference returns it.  The state for prediction at REML parameters is the NOISY-mode one:
 * bogp_commit(mode = BOGP_MODE_NOISY, par = [theta, sigma2], noise_var).                                       */
int bogp_nll_restricted(bogp_handle* h, int kernel, int mode, const double* par, int n_par, double noise_var, int trend,
                        int estimate_trend, double beta, double* llf, double* grad);

/* ---- polynomial trend bases (trend.py:66-142) --------------------------------------------------------
 * `trend` in bogp_nll / bogp_commit selects the basis F: BOGP_TREND_CONSTANT (p = 1, the scalar `beta` argument),
 * BOGP_TREND_LINEAR (p = d + 1), BOGP_TREND_QUADRATIC (p = (d+1)(d+2)/2).  With estimate_trend = 1 the coefficients are
 * the GLS estimate (universal kriging, gpr.py:801-806: Ft = L^-1 F, Q G = Ft, rho = Yt - Q Q^T Yt, beta = G^-1 Q^T Yt);
 * with estimate_trend = 0 and p > 1 they are the p values last given to bogp_set_trend_beta (the scalar `beta` argument
 * is ignored).  The QR factor G has a positive diagonal (LAPACK's Householder G differs by row signs; beta, rho, the
 * predictor and the likelihood do not depend on them).
 *   bogp_trend_size       p for a basis id and d
 *   bogp_get_trend_state  Ft, Q (N x p row-major), G (p x p row-major), beta (p) of the committed model; any may be NULL */
int bogp_trend_size(int trend, int d);
int bogp_set_trend_beta(bogp_handle* h, const double* beta, int p);
int bogp_get_trend_state(bogp_handle* h, double* Ft, double* Q, double* G, double* beta);

/* ---- candidates -----------------------------------------------------------------------------------
 * M x d row-major float64.  `upload` copies from host (PCIe); `bind` adopts caller-owned DEVICE memory
 * (e.g. a torch tensor's data_ptr()) without copying -- it must stay alive until the next upload/bind.   */
int bogp_candidates_upload(bogp_handle* h, const double* Xs, int64_t M);
/* The same upload, overlapped with the sweep that follows: only the first 8 MB are copied here; the next bogp_predict / bogp_sweep /
 * bogp_sweep_topk copies the rows of candidate chunk c + 1 on a copy stream WHILE chunk c is contracted (a host-sampled ask() of
 * optimizer="sweep", acquisition/optim/__init__.py:55-153, no longer pays the H2D copy of its M x d candidates in front of the
 * sweep).  `Xs` must stay valid and unchanged until that next call returns -- it returns with every row resident, later calls
 * need the buffer no more.  One-launch sweeps (N <= 512) and single-chunk sweeps finish the upload first.                       */
int bogp_candidates_upload_lazy(bogp_handle* h, const double* Xs, int64_t M);
int bogp_candidates_bind(bogp_handle* h, const void* d_Xs, int64_t M);
/* `generate` draws M uniform points in the box [lo, hi] ON the device (replaces RealSpace._sample,
 * search_space/search_space.py:742-754, and the H2D copy): counter-based Philox4x32-10, element (row, k) is a pure
 * function of (seed, (first_row + row) * d + k), so shards of one stream are drawn independently per rank.
 * `read` copies chosen rows (e.g. the argmax) of the current candidates back: out is n x d.                  */
int bogp_candidates_generate(bogp_handle* h, const double* lo, const double* hi, int64_t M, uint64_t seed,
                             int64_t first_row);
/* `generate_lhs`: rows [first_row, first_row + M) of an n_strata-point Latin hypercube (method "LHS",
 * search_space.py:747-751 -> pyDOE.lhs): per dimension one point in each of n_strata equal strata, the strata
 * visited in a keyed pseudo-random permutation (Feistel network + cycle walking, evaluated per element, so no sort
 * and no exchange between ranks), jitter inside the stratum from the uniform stream.
 * `generate_lhs_maximin`: what the reference's DoE call actually asks pyDOE for (search_space.py:751: lhs(dim, samples=N,
 * criterion="maximin"), pyDOE's _lhsmaximin): `iterations` (pyDOE: 5) independent M-point hypercubes in the UNIT cube --
 * trial t is stream seed + 0x9E3779B97F4A7C15 * t -- are measured by the minimum pairwise Euclidean distance (the O(M^2 d)
 * pair sweep of `min_pdist2`; M <= 2^18) and the FIRST design with the largest minimum (pyDOE's `if maxdist < min(d)`) is
 * generated into the box [lo, hi] (+ the transform below); its minimum distance and trial number are returned.  Whole
 * designs only (no row shards: the criterion needs every pair).
 * `min_pdist2`: min over i < j of |x_i - x_j|^2 of the current candidates (sequential, un-fused sum over the dimensions:
 * bit-reproducible), +inf for fewer than two points.
 * `generate_sobol`: points [first_index, first_index + M) of the unscrambled Sobol' sequence (method "sobol",
 * search_space.py:752-753 -> sobol_seq.i4_sobol_generate, whose skip = 1 is first_index = 1) for caller-supplied
 * direction numbers sv (d x bits, row-major, sv[k][b] is XORed in when bit b of the Gray code of the index is set --
 * the layout of scipy.stats.qmc.Sobol._sv); value = integer * 2^-bits, then lo + (hi - lo) * value.            */
int bogp_candidates_generate_lhs(bogp_handle* h, const double* lo, const double* hi, int64_t M, uint64_t seed,
                                 int64_t first_row, int64_t n_strata);
int bogp_candidates_generate_sobol(bogp_handle* h, const double* lo, const double* hi, int64_t M, int64_t first_index,
                                   const uint64_t* sv, int bits);
int bogp_candidates_generate_lhs_maximin(bogp_handle* h, const double* lo, const double* hi, int64_t M, uint64_t seed,
                                         int iterations, double* best_min_dist, int* best_iteration);
int bogp_candidates_min_pdist2(bogp_handle* h, double* min_sq);
int bogp_candidates_read(bogp_handle* h, const int64_t* rows, int n, double* out);
/* Per-variable post-processing of the three generators above, as RealSpace._sample applies it (search_space.py:754:
 * `self.round(self.to_linear_scale(X))`): draw in the TRANSFORMED box (pass trans(lo), trans(hi) to the generator:
 * Real._bounds_transformed, variable.py:246), map every coordinate back with the inverse of its scale (variable.py:40-55)
 * and, where precision[k] >= 0, round to that many decimals and clip to [lo[k], hi[k]] (the variable's own bounds,
 * variable.py:250-257).  scale / precision / lo / hi hold d entries; scale == NULL && precision == NULL switches the
 * post-processing off again.  The setting persists on the handle until changed (it needs bogp_set_train for d).      */
#define BOGP_SCALE_LINEAR 0
#define BOGP_SCALE_LOG 1   /* x = exp(u)                      */
#define BOGP_SCALE_LOG10 2 /* x = 10^u                        */
#define BOGP_SCALE_LOGIT 3 /* x = 1 / (1 + exp(-u))           */
#define BOGP_SCALE_BILOG 4 /* x = sign(u) (exp(|u|) - 1)      */
int bogp_candidates_set_transform(bogp_handle* h, const int* scale, const int* precision, const double* lo, const double* hi);

/* ---- posterior ------------------------------------------------------------------------------------
 * Replaces GaussianProcess.predict(X, eval_MSE) (gpr.py:486-510) on the current candidates:
 * mu (M) and mse (M, may be NULL) are HOST buffers.  With mse == NULL the N^2-per-candidate variance contraction is
 * not run at all (the reference's eval_MSE=False path also returns before its triangular solve, :491).          */
int bogp_predict(bogp_handle* h, double* mu, double* mse);

/* ---- sweep: posterior + q acquisition criteria + argmax ---------------------------------------------
 * Replaces the inner maximiser behind acquisition/optim/__init__.py:55-153 (argmax_restart) for
 * optimizer="sweep", evaluating AcquisitionFunction.__call__ row by row (acquisition_fun.py:127-135,
 * 153-176, 208-217, 265-290; _predict :52-64; plugin :96-104) for q criteria that share (mu, MSE)
 * (ParallelBO._batch_arg_max_acquisition, bayes_opt.py:100-115).
 *   plugin    effective plugin as stored by ImprovementBased.plugin (already negated when maximising)
 *   minimize  0/1 as AcquisitionFunction.minimize
 *   best_val  out (q): acquisition value at the argmax;  best_idx out (q): np.argmax index (first
 *             maximum; a NaN, if present, wins at its first position) -- local to these M candidates
 *   acq_out   optional HOST buffer (q x M row-major) receiving every acquisition value; NULL to skip
 * best_val and best_idx may BOTH be NULL: the sweep is then only queued on the handle's stream (no host wait, no
 * read-back) and its winners stay on the device for the bogp_exchange_argmax call that follows (multi-GPU step).  */
int bogp_sweep(bogp_handle* h, int q, const int* acq_id, const double* acq_par, double plugin, int minimize,
               double* best_val, int64_t* best_idx, double* acq_out);

/* Same sweep, returning the k best candidates per criterion (best_val, best_idx: q x k row-major, rank 0 = the
 * argmax; ties -> lower index; slots beyond M are (-inf, -1)).  Gives BO.pre_eval_check (bayes_opt.py:27-55)
 * and ParallelBO's q criteria fall-backs instead of the reference's random padding (base.py:282-289) when
 * several criteria -- or a criterion and the history -- agree on the same candidate.                        */
int bogp_sweep_topk(bogp_handle* h, int q, const int* acq_id, const double* acq_par, double plugin, int minimize,
                    int k, double* best_val, int64_t* best_idx);

/* ---- input-gradient of the posterior at ONE point ---------------------------------------------------
 * Replaces GaussianProcess.gradient(x) (gpr.py:537-576, corr_dx :600-661): dmu (d), dmse (d).            */
int bogp_gradient(bogp_handle* h, const double* x, double* dmu, double* dmse);
/* The same for B points at once (Xb: B x d; dmu, dmse: B x d row-major); constant or (r05) linear trend basis, one target.  r03: k_point_rhs +
 * k_point_tri (csrc/kernels_point.hip) -- ONE pass over L^-1 per point serves all d + 1 right-hand sides.     */
int bogp_gradient_batch(bogp_handle* h, const double* Xb, int B, double* dmu, double* dmse);
/* Hessian of the posterior mean at x (GaussianProcess.Hessian, gpr.py:578-598), d x d row-major.  Squared exponential
 * only, like the reference's corr_Hessian (:663-734); constant / linear trend (their Hessians are zero, trend.py:88-116). */
int bogp_hessian(bogp_handle* h, const double* x, double* H);
/* Correlation matrix of the rows of X1 (n1 x d) at the committed theta: GaussianProcess.prior_cov(X1, corr=True)
 * (gpr.py:318-353).  R is n1 x n1 row-major; the covariance flavour is R scaled on the host (:350-351).               */
int bogp_prior_corr(bogp_handle* h, const double* X1, int n1, double* R);
/* One point, everything at once: what `criterion(x, return_dx=True)` needs (acquisition_fun.py:139-146, 181-188,
 * 220-227, 292-309 call predict, gradient and the closed form) -- mu, mse, dmu (d), dmse (d) and the q criterion values --
 * with a single host synchronisation.  This is the call the reference's DEFAULT inner optimiser (multi-restart
 * L-BFGS-B, base.py:201-243) makes thousands of times per ask().  Constant or (r05) linear trend basis -- the two the
 * reference's `gradient` differentiates (gpr.py:556-575; the quadratic basis has no Jacobian, trend.py:138-139) --, one target;
 * q may be 0.                                                                                                        */
int bogp_point_eval(bogp_handle* h, const double* x, int q, const int* acq_id, const double* acq_par, double plugin,
                    int minimize, double* mu, double* mse, double* dmu, double* dmse, double* acq);
/* The same for B points (Xb: B x d, host) in ONE device round trip, plus the criteria's own input-gradients -- the
 * `return_dx` chain rule of acquisition_fun.py:139-146 (UCB), 181-188 (EI), 220-227 (EpsilonPI), 292-309 (MGFI) evaluated
 * on the device, guards (zero gradient) included.  Outputs, row-major, any may be NULL: mu, mse (B), dmu, dmse (B x d),
 * acq (B x q), dacq (B x q x d).  The reference raises for more than one row (gpr.py:548-549); row b here is what its
 * one-row call returns for row b.  Constant trend basis.                                                      */
int bogp_point_eval_batch(bogp_handle* h, const double* Xb, int B, int q, const int* acq_id, const double* acq_par,
                          double plugin, int minimize, double* mu, double* mse, double* dmu, double* dmse, double* acq,
                          double* dacq);
/* Multi-start local maximisation of ONE criterion inside the box [lo, hi] (d each), all B starts in lock step on the
 * device: every iteration is one batched value + gradient evaluation of the B trial points (as bogp_point_eval_batch,
 * records stay on the device) and one optimiser step per start (projected L-BFGS, 8 curvature pairs, Armijo
 * backtracking; a start never moves to a worse point).  Replaces the restart loop of acquisition/optim/__init__.py:74-153
 * when it is started from the sweep's top-k -- the reference runs its restarts one after the other, one
 * point per call.  Stop rules per start as the reference configures scipy's L-BFGS-B (:94-101): projected gradient
 * below pgtol (1e-8), relative improvement below factr (1e6) x machine epsilon, or max_evals evaluations.
 * X0: B x d starting points (clipped into the box); Xout: B x d; fout: B (criterion value at Xout, >= the value at
 * X0); n_evals: B evaluations used, may be NULL.  any d up to BOGP_MAX_DIM (r05; r03-r04: d <= 64); constant or linear trend basis.                              */
int bogp_polish(bogp_handle* h, const double* X0, int B, const double* lo, const double* hi, int acq_id, double acq_par,
                double plugin, int minimize, int max_evals, double pgtol, double factr, double* Xout, double* fout,
                int* n_evals);

/* ---- multi-GPU: the one exchange step per sweep ------------------------------------------
 * Nothing like it exists in the reference (its only parallelism is joblib over the q criteria, bayes_opt.py:108-111);
 * this is the cross-rank half of the new inner maximiser behind acquisition/optim/__init__.py:55-153.  One process per
 * GPU, each with its own handle and its own contiguous block of the M candidates; the model is replicated by redundant,
 * deterministic factorisation (no broadcast).  After bogp_sweep / bogp_sweep_topk every rank calls bogp_exchange_*: its q
 * (or q x k) winners -- (value, LOCAL index + index_offset, point) -- are packed on the device from the sweep's own result
 * buffers, gathered with ONE ncclAllGather (RCCL over xGMI; q (2 + d) doubles per rank), read back once, and reduced by
 * the same deterministic rule on every rank: np.argmax over the concatenation of all shards (largest value; a NaN beats
 * every number; ties -> lowest global index).  All ranks return identical results.
 *   bogp_comm_unique_id   fills BOGP_COMM_ID_BYTES bytes (ncclGetUniqueId); call on ONE rank, hand the bytes to all
 *                         ranks by any means (MPI_Bcast, a file, torch.distributed's store ...)
 *   bogp_comm_init        ncclCommInitRank on the handle's device; collective over the `world` ranks; world = 1 is fine
 *   bogp_comm_attach      borrow an existing ncclComm_t instead (not destroyed with the handle)
 *   bogp_comm_info        rank / world of the handle's communicator (world = 0: none)
 *   bogp_exchange_argmax  after bogp_sweep:      best_val (q), best_gidx (q), best_x (q x d, may be NULL)
 *   bogp_exchange_topk    after bogp_sweep_topk: best_val, best_gidx (q x k), best_x (q x k x d, may be NULL); global
 *                         k best per criterion in the same order; empty slots (-inf, -1, NaN)
 *   bogp_reduce_pairs / bogp_merge_topk   the reduce itself on HOST records [R][q( x k)][2 + d] = (value, index bit
 *                         pattern, point) for callers that gather by their own transport (MPI, gloo); need no handle. */
int bogp_comm_unique_id(unsigned char* id_out);
int bogp_comm_init(bogp_handle* h, const unsigned char* id, int rank, int world);
int bogp_comm_attach(bogp_handle* h, void* nccl_comm);
int bogp_comm_info(const bogp_handle* h, int* rank, int* world);
int bogp_comm_destroy(bogp_handle* h);
int bogp_exchange_argmax(bogp_handle* h, int64_t index_offset, double* best_val, int64_t* best_gidx, double* best_x);
int bogp_exchange_topk(bogp_handle* h, int64_t index_offset, double* best_val, int64_t* best_gidx, double* best_x);
int bogp_reduce_pairs(int R, int q, int d, const double* gathered, double* val, int64_t* gidx, double* x);
int bogp_merge_topk(int R, int q, int k, int d, const double* gathered, double* val, int64_t* gidx, double* x);

/* ---- measurement ----------------------------------------------------------------------------------
 * HIP-event durations (ms, summed over candidate chunks) of the kernels of the LAST bogp_predict /
 * bogp_sweep call, recorded on the library's own stream: corr (k_corr_chunk, the K* producer), contract
 * (k_contract, the dominant L^-1 MFMA contraction) and acquisition/argmax.  n_chunks = launches of each.  */
int bogp_last_timing(bogp_handle* h, double* corr_ms, double* contract_ms, double* acquisition_ms, int* n_chunks);

/* Algorithmic FP64 flops per candidate of the posterior for the committed model:
 * N^2 + N (3d + 5 + 2p)  (SURVEY.md section 8d).                                                        */
double bogp_flops_per_candidate(const bogp_handle* h);

/* ---- self test -------------------------------------------------------------------------------------
 * The in-tree dense product behind the polynomial-trend, REML and small-batch paths (kernels_gemm.hip; the library links no
 * BLAS), on host buffers: C (m x n, ldc) = alpha op(A) (m x k) op(B) (k x n) + beta C, column-major; ta / tb != 0: the stored
 * matrix is the transpose; tri = 1 / 2: op(A) is square lower / upper triangular with explicit zeros in the other triangle
 * (the kernel then shortens its k range); split != 0: the deterministic split-K path may be taken (small C, long k).
 * What the reference gets from numpy.dot / scipy.linalg (gpr.py:799-808, 850-918); used by tests/test_gpu_gemm.py.        */
int bogp_selftest_gemm(bogp_handle* h, int ta, int tb, int m, int n, int k, double alpha, const double* A, int lda,
                       const double* B, int ldb, double beta, double* C, int ldc, int tri, int split);

/* The radial profile r(s2) of correlation function `kernel` (BOGP_SELFTEST_PROFILE; s2 = the weighted squared distance the producers
 * accumulate, pexp = the exponent p / the order nu where the kernel has one) or one of the special functions behind it, evaluated on the
 * device for n arguments: what the reference gets per pair from kernel.py:186-207, 289-329 (numpy sqrt / exp, scipy.special.kv and
 * scipy.special.gamma).  BESSEL_K: K_pexp(arg); RGAMMA: 1 / Gamma(arg); BESSEL_K_PAIRS: arg holds n pairs (nu, x) -> K_nu(x);
 * MATERN_NU_PAIRS: n pairs (nu, s2) -> the general-nu Matern profile at squared distance s2.
 * Used by tests/test_gpu_special.py to hold them to committed tables argument by argument.                                            */
#define BOGP_SELFTEST_PROFILE 0
#define BOGP_SELFTEST_BESSEL_K 1
#define BOGP_SELFTEST_RGAMMA 2
#define BOGP_SELFTEST_BESSEL_K_PAIRS 3
#define BOGP_SELFTEST_MATERN_NU_PAIRS 4
int bogp_selftest_profile(bogp_handle* h, int what, int kernel, double pexp, const double* arg, int64_t n, double* out);

/* The optimiser behind bogp_mle_batch (csrc/bogp_lbfgsb.h: L-BFGS-B after Byrd, Lu, Nocedal & Zhu 1995 with the More'-Thuente
 * line search) on a HOST objective: what the reference gets from scipy.optimize.fmin_l_bfgs_b (gpr.py:1136).  No device, no handle;
 * used by tests/test_lbfgsb.py to compare it with scipy on the CPU.  x: in the start, out the minimiser (n); fn fills *f and g (n).
 * m / maxfun / maxiter <= 0: 10 / 15000 / 15000.  status as in bogp_mle_batch.                                                  */
typedef void (*bogp_objective_fn)(const double* x, int n, double* f, double* g, void* user);
int bogp_lbfgsb_minimize(int n, int m, double* x, const double* lo, const double* hi, double factr, double pgtol, int maxfun,
                         int maxiter, bogp_objective_fn fn, void* user, double* f, int* nfev, int* nit, int* status);

/* Which device path a concentrated-likelihood evaluation (bogp_nll) of N points in d dimensions would take with the constant
 * basis (trend) and n_targets columns of y -- no handle, no device call (the decision is made from sizes and the environment
 * switches alone; gpr.py:920-1040 is the function all three evaluate):
 *   BOGP_NLL_PATH_GENERAL   the multi-kernel path (blocked Cholesky, recursive-doubling inverse, U U^T, ...)
 *   BOGP_NLL_PATH_ONE_LAUNCH  the whole evaluation in one launch of one workgroup (k_nll_small; N <= 156 if it fits one CU's LDS)
 *   BOGP_NLL_PATH_ELIM      one launch per 64 columns (k_elim_step; up to N = 3072)
 * DESIGN.md section 5.12; used by tests/test_abi.py.                                                                         */
#define BOGP_NLL_PATH_GENERAL 0
#define BOGP_NLL_PATH_ONE_LAUNCH 1
#define BOGP_NLL_PATH_ELIM 2
int bogp_nll_path(int N, int d, int trend, int n_targets);

/* The schedule of WIDE FIRST PANELS the blocked Cholesky of the general path (gpr.py:795, `cholesky(R, lower=True)`) takes for N training
 * points -- no handle, no device call.  Returns the number of wide panels (0 below N = 6017 and with BOGP_BIG_CHOL=0) and writes up to `cap`
 * widths (64-column blocks per panel) to `widths` (may be NULL).  Inside a wide panel the rank-64 updates touch the panel's own columns
 * only and ONE rank-64w product on 128 x 128 tiles updates the rest of the trailing matrix; every schedule yields the SAME BITS as the
 * one-level chain (the products run over k in the same order from the same starting value).  DESIGN.md section 5.9; tests/test_abi.py,
 * tests/test_gpu_driver.py.                                                                                                         */
int bogp_chol_wide_panels(int N, int* widths, int cap);

#ifdef __cplusplus
}
#endif
#endif /* BOGP_H */
