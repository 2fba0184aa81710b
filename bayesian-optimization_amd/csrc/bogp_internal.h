// bogp_internal.h -- host-side declarations shared by the translation units of libbogp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace bogp {

// ---- kernel argument blocks (passed by value) -------------------------------------------------------
struct CorrArgs {
  const double* Xs;          // candidates, M x d row-major (device)
  int64_t M;                 // total candidates
  int64_t m0;                // first candidate of this chunk
  int64_t Mc;                // chunk size padded to a multiple of 64
  int d;
  int Np;                    // training points padded to a multiple of 32
  int nblk_per_split;        // 32-blocks of n per workgroup (grid.y slices the training set)
  const double* sqrt_theta;  // d
  const double* XthT;        // [d][Np] theta-scaled training points, transposed
  const double* xnorm = nullptr;  // [Np] squared norms of the columns of XthT (k_corr_mfma); null: kernel A
  const double* gamma;       // Np (zero padded)
  const double* wvec;        // Np (zero padded): L^-T Ft, or zeros for simple kriging
  double* rT;                // [Np][Mc] correlation chunk, n-major
  double* mu_part;           // [S][Mc]
  double* w_part;            // [S][Mc]
  // polynomial trend with few columns, fused into the producer (k_corr_chunk<K, PV>): pv = 0 (off) / 16 / 24 / 32 >= p
  int pv = 0;
  const double* Wrow = nullptr;  // [Np][wld] row-major W = L^-T Ft, zero padded columns
  int wld = 0;
  double* t_part = nullptr;      // [S][pv][Mc] slice sums of W^T r
};
int corr_trend_columns(int p);  // the PV instantiation serving p trend columns, 0 if none does

struct ContractArgs {
  const double* rT;   // [Np][Mc]
  const double2* Vp;  // packed L^-1: [NJ16][NKP][64] double2
  double* ss_part;    // [nJ][Mc]
  int64_t Mc;
  int nMt;   // Mc / 64
  int nJ;    // column groups
  int NJ16;  // Np / 16
  int NKP;   // Np / 8
  int64_t cross_B;  // launch_contract_cross: points (ss_part is then their record array, see k_contract16<NR, NCP>)
};

struct AcqArgs {
  const double* mu_part;  // [S][Mc]
  const double* w_part;   // [S][Mc]
  const double* ss_part;  // [nJ][Mc]
  int S, nJ;
  int nJ_plus = 0;        // trend-rows path (k_pack_Vx): column groups nJ .. nJ + nJ_plus - 1 of ss_part hold |u|^2 and are ADDED
  int64_t Mc;      // chunk stride of the partial arrays
  int64_t mcount;  // valid candidates in this chunk
  int64_t m0;      // global index of the chunk's first candidate
  double beta;     // trend coefficient (constant basis)
  const double* mtrend;  // [Mc] f(x*) . beta for polynomial bases (replaces beta), or null
  const double* uu;      // [Mc] u^T u for polynomial bases under universal kriging (replaces the w_part term), or null
  double G;        // QR factor of Ft (ordinary kriging), unused otherwise
  int estimate_trend;
  double sigma2;
  double* mu_out;   // [M] or null (global arrays, indexed by m0 + i)
  double* mse_out;  // [M] or null
  int q;
  int acq_id[64];
  double acq_par[64];
  double plugin;
  int minimize;
  double* acq_out;     // [q][M] or null
  int64_t M;           // row stride of acq_out
  double* blk_val;     // [q][nblk_total] per-block partial argmax
  int64_t* blk_idx;    // [q][nblk_total]
  int64_t blk_offset;  // first block slot of this chunk
  int64_t nblk_total;
};

// the fused small-N sweep (kernels_small.hip): one launch = producer + contraction + posterior + q criteria + argmax
struct SmallArgs {
  const double* Xs;          // candidates, M x d row-major
  const double* sqrt_theta;  // d
  const double* XthT;        // [d][Np]
  const double* gamma;       // Np
  const double* wvec;        // Np
  const double2* Vp;         // packed L^-1
  int64_t M;
  int d, Np, NJ16, NKP;
  int need_var;              // 0: predict without MSE (no contraction)
  double beta, G, sigma2, plugin;
  int estimate_trend, minimize, q;
  int acq_id[64];
  double acq_par[64];
  double* mu_out;            // [M] or null
  double* mse_out;           // [M] or null
  double* acq_out;           // [q][M] or null
  double* blk_val;           // [q][nblk]
  int64_t* blk_idx;
  int64_t nblk;
  unsigned int* counter;     // device word, zero between launches: ticket of the last workgroup
  double* best_val;          // [q]
  int64_t* best_idx;
  long long* stamps;         // null, or 5 device words: wave-cycles producing / contracting / waiting / epilogue, waves
  int64_t m_begin;           // first candidate of THIS launch (set by launch_sweep_small: bulk launch 0, tail launch after it)
  int64_t blk_begin;         // first partial-argmax slot of this launch
  int final_launch;          // the launch whose last workgroup reduces the partial winners
};
bool sweep_small_supported(int Np, int d, int kernel);
int64_t sweep_small_blocks(int64_t M, int n_cu);
hipError_t launch_sweep_small(int kernel, const SmallArgs& a, int n_cu, hipStream_t st);

hipError_t launch_corr_chunk(int kernel, const CorrArgs& a, int nMt, int S, hipStream_t st);
hipError_t launch_contract(const ContractArgs& a, hipStream_t st);
hipError_t launch_contract_cross(const ContractArgs& a, int ncp, hipStream_t st);
int contract_cols_per_group();
hipError_t launch_acquisition(const AcqArgs& a, hipStream_t st);
hipError_t launch_argmax_final(const double* blk_val, const int64_t* blk_idx, int64_t nblk, int64_t stride, int q,
                               double* out_val, int64_t* out_idx, hipStream_t st);

hipError_t launch_topk(const double* vals, int64_t M, int q, int k, double* blk_val, int64_t* blk_idx, double* out_val,
                       int64_t* out_idx, hipStream_t st);

hipError_t launch_generate_uniform(double* Xs, int64_t n_elem, int d, const double* lo, const double* hi, uint64_t seed,
                                   uint64_t first_elem, hipStream_t st);
hipError_t launch_generate_lhs(double* Xs, int64_t n_elem, int d, const double* lo, const double* hi, uint64_t seed,
                               uint64_t first_elem, uint64_t n_strata, hipStream_t st);
hipError_t launch_candidates_transform(double* Xs, int64_t n_elem, int d, const double* spec, hipStream_t st);
hipError_t launch_generate_sobol(double* Xs, int64_t n_elem, int d, const double* lo, const double* hi, const uint64_t* sv,
                                 int bits, uint64_t first_elem, hipStream_t st);

// fit-path kernels (kernels_fit.hip)
hipError_t launch_min_pdist2(const double* X, int M, int d, unsigned long long* out, hipStream_t st);
hipError_t launch_build_R(int kernel, const double* X, int N, int d, const double* theta, double off_scale, double diag,
                          double* R, int ld, hipStream_t st);
hipError_t launch_build_R_div(int kernel, const double* X, int N, int d, const double* theta, double mul, double div,
                              double diag, double* R, int ld, hipStream_t st);
hipError_t launch_resid_gamma(int kernel, bool div, const double* X, int N, int d, const double* theta, double a, double b,
                              double diag, const double* bvec, const double* gamma, double* res, hipStream_t st);
hipError_t launch_sub_const(const double* y, double c, double* out, int N, hipStream_t st);
hipError_t launch_add_vec(double* y, const double* x, int N, hipStream_t st);
hipError_t launch_scale_transpose(const double* X, int N, int d, int Np, const double* sqrt_theta, double* XthT, double* xnorm,
                                  hipStream_t st);
hipError_t launch_pack_V(const double* Vcm, int N, int ld, int Np, double2* Vp, hipStream_t st);
hipError_t launch_logdet(const double* L, int N, int ld, double* out, hipStream_t st);
// kernels_chol.hip: in-place lower Cholesky of a column-major matrix whose ld is a multiple of 64 (identity padded)
// ---- kernels_gemm.hip: the small dense products that used to be library calls ------------------------------------
struct GemmArgs {
  const double* A;
  const double* B;
  double* C;
  int m, n, k, ldc;
  long sai, sak;  // op(A)(i, kk) at A[i * sai + kk * sak]
  long sbj, sbk;  // op(B)(kk, j) at B[j * sbj + kk * sbk]
  double alpha, beta;
  int tri;  // 1 / 2: op(A) square lower / upper triangular (zeros stored): the k range of a row tile shrinks
  double* scratch;        // split K (gridDim.z > 1): [tile][slice][64 x 64] partial tiles
  unsigned int* tickets;  // one zeroed word per tile
};
struct GemmSplit {  // what launch_gemm needs to split K: scratch of `cap` doubles, `max_tiles` zeroed ticket words
  double* scratch;
  size_t cap;
  unsigned int* tickets;
  int max_tiles;
};
// C (m x n) = alpha op(A) op(B) + beta C, column-major; ta / tb != 0: the stored matrix is the transpose (any shape)
hipError_t launch_gemm(int ta, int tb, int m, int n, int k, double alpha, const double* A, int lda, const double* B, int ldb,
                       double beta, double* C, int ldc, hipStream_t st, int tri = 0, const GemmSplit* sp = nullptr);
// out[c + r * ldo] = in[r + c * ldi], r < rows, c < cols, zero for cols <= c < ldo
hipError_t launch_transpose_pad(const double* in, int ldi, int rows, int cols, double* out, int ldo, hipStream_t st);
// kernels_chol.hip, k_mm128 on a plain product: out (128 TI x 128 TJ, ldo) = sum_{k < K} Rs(row, k) Cs(col, k), element (row, k)
// at Rs[row + k * ldr], (col, k) at Cs[col + k * ldc]; K a multiple of 16, every tile complete
hipError_t launch_mm128_gen(const double* Rs, int ldr, const double* Cs, int ldc, double* out, int ldo, int TI, int TJ, int K,
                            hipStream_t st);
hipError_t launch_pad_identity(double* A, int N, int ld, hipStream_t st);
// st2 + ev[2] (optional): a second stream and two events for the look-ahead of the large-matrix path; N (optional): the
// data columns -- [N, ld) is identity padding whose diagonal-block steps are skipped; chain_flags (optional, with st2 and ev):
// 2 * (ld / 64) words for the resident diagonal chain of the fused block columns (k_chol_chain)
int chol_wide_panels(int ld, int* widths, int cap);  // wide first panels launch_chol_lower takes at this leading dimension (host only)
hipError_t launch_chol_lower(double* A, int ld, double* Winv, int* info, hipStream_t st, hipStream_t st2 = nullptr,
                             hipEvent_t* ev = nullptr, double* scratch = nullptr, int N = 0, unsigned int* chain_flags = nullptr);
hipError_t launch_tri_inverse(const double* L, const double* Winv, double* V, double* U, double* T, int ld, hipStream_t st);
constexpr int UUT_PARTS = 4;  // R^-1 = U U^T is produced as this many K-slices (ld*ld doubles apart) that the consumers add
hipError_t launch_uut(const double* U, double* Rinv, int ld, hipStream_t st, int* nparts);
hipError_t launch_copy_lower(const double* L, int N, int ld, double* dst, hipStream_t st);
// the rank-1 terms of the likelihood gradient: vectors v + t * stride (t < n) with weights cA (theta contractions) and
// cB (R0 contraction); c0 multiplies R^-1
struct GradVecs {
  const double* v;
  size_t stride;
  int n;
  double c0;
  double cA[BOGP_MAX_TARGETS], cB[BOGP_MAX_TARGETS];
  const double* dcoef = nullptr;  // if set: cA = dcoef[0..7], cB = dcoef[8..15] read on the device (k_grad_coef) instead
};
hipError_t launch_grad_contract(int kernel, const double* X, int N, int d, const double* theta, const GradVecs& gv,
                                const double* qv, double c2, const double* Rinv, int ld, int nparts,
                                size_t part_stride, double* partial, int nblk, hipStream_t st);
hipError_t launch_grad_reduce(const double* partial, int nblk, int nout, double* out, hipStream_t st);
int grad_contract_blocks(int N);
size_t gemv2_scratch_doubles(int N);
// polynomial trends (p > 1)
hipError_t launch_trend_train(int trend, const double* X, int N, int d, double* F, hipStream_t st);
hipError_t launch_pack_Vx(const double* Vcm, int N, int ld, const double* At, int ldA, const double* Ginv, int p, int Ne, int Nt,
                          double2* Vp, hipStream_t st);
hipError_t launch_trend_rows(int trend, const double* Xs, int64_t m0, int64_t mcount, int64_t mrows, int d, int64_t Mc, const double* beta,
                             double* Rext, int p, int prows, double* mtrend, hipStream_t st);
hipError_t launch_trend_terms(int trend, const double* Xs, int64_t m0, int64_t mcount, int d, int64_t Mc, const double* beta,
                              double* T, double* mtrend, hipStream_t st);
// small polynomial bases after the fused producer: T = sum of the slice sums, c = T - f(x*), mtrend = f(x*) . beta, uu = c^T Sinv c in ONE launch
hipError_t launch_trend_small(int trend, const double* Xs, int64_t m0, int64_t mcount, int d, int64_t Mc, const double* beta, const double* t_part,
                              int S, int pv, int p, const double* Sinv, double* mtrend, double* uu, hipStream_t st);
hipError_t launch_rowdot(const double* Cm, const double* CS, int64_t Mc, int64_t mcount, int p, double* uu, hipStream_t st);
hipError_t launch_sumsq(const double* v, int N, double* out, hipStream_t st);
// the one-launch likelihood of a small training set (kernels_nllsmall.hip)
struct NllSmallArgs {
  const double* X;  // N x d, row-major
  const double* y;  // N
  int N, d;
  double theta[64];  // d entries
  double pexp;       // generalized_exponential's exponent
  double a, b, diag;  // off-diagonal = a corr (/ b if div), diagonal = diag: k_build_R's arguments
  int div, estimate_trend, mode;
  double beta, s2t_host;
  double* out_scal;  // 64 doubles (device address of pinned host memory)
  double* out_S;     // d + 3 doubles
  unsigned long long* flag;
  unsigned long long seq;
  // bogp_nll_batch: P parameter vectors, one workgroup each.  bpar[s][NS_BPAR] = theta (64) | pexp | a | b | diag | s2t_host | div of
  // slot s (what `theta` .. `s2t_host` above are for one evaluation), bout[s][NS_BOUT] = its 64 scalars | d + 3 sums, bticket = one
  // zeroed device word (the last workgroup publishes `seq`).  bpar == nullptr: the one-evaluation launch.
  const double* bpar = nullptr;
  double* bout = nullptr;
  unsigned int* bticket = nullptr;
  int P = 1;
  int bout_stride = 0;  // doubles per slot of bout: 64 scalars + the d + 3 sums
};
constexpr int NS_BPAR = 72;   // doubles per slot of NllSmallArgs::bpar
int nll_small_max_n();
bool nll_small_fits(int N, int d);
// the elimination at 64-block granularity (kernels_chol.hip: k_elim_*): 157 <= N <= 1024, constant basis, one target
struct ElimArgs {
  double* E;    // ld x ld, column-major: the state blocks (k_build_R's output to start with)
  double* Eb;   // 64 x ld (leading dimension 64): block row nb = the right-hand sides [y; 1; 0 ...]
  int ld, nb, N;
  double* yt;   // ld: L^-1 y
  double* ft;   // ld: L^-1 1
  double* logpart;  // nb: sum(log diag L_kk)
  int* info;
};
// one parameter vector of bogp_nll_batch on the elimination path: its workspace (fixed while the batch buffers live) and where this
// evaluation's parameters are (rewritten by every call) -- an array of these in device memory is what the *_b kernels index by slot
struct BatchSlot {
  const double* theta;  // d + 1 doubles: theta, then the exponent of generalized_exponential
  const double* par;    // [0] a, [1] b, [2] diag (k_build_R's arguments), [3] sigma2 + noise_var (noisy mode)
  ElimArgs ea;          // state blocks, block row nb, Yt, Ft, log-determinant parts; ea.info = (int*)(scal + 62)
  double* Winv;         // (nb + 1) x 64 x 64: inverses of the diagonal blocks
  double* panels;       // 2 x (ld + 64) x 64: the two raw panels
  double* xpanel;       // 4 x (nb + 1) x 64 x 64: the solved panels of split steps, by k mod 4 (k_elim_panel_b -> k_elim_update_b / updateG_b / updateS_b)
  double* Rinv;         // ld x ld
  double* gamma;        // Np
  double* scal;         // 64: [0..3] the likelihood's scalars, [32..48) the gradient's weights, [62] info
  double* partial;      // grad_contract_blocks(N) x (d + 1) tile sums
  double* S;            // d + 3 gradient sums
  unsigned int* ticket; // zeroed word of k_grad_finish_b
};
hipError_t launch_build_R_batch(int kernel, bool div, const double* X, int N, int d, const BatchSlot* slots, int P, int ld, hipStream_t st);
hipError_t launch_elim_batch(const BatchSlot* slots, int P, int ld, const double* y, int estimate_trend, int mode, double beta, hipStream_t st);
hipError_t launch_grad_contract_batch(int kernel, const double* X, int N, int d, const BatchSlot* slots, int P, int Np, int ld, hipStream_t st);
hipError_t launch_grad_finish_batch(const BatchSlot* slots, int P, int nblk, int nout, int ld, int N, int with_trace, double* bout,
                                    int bout_stride, unsigned long long* flag, unsigned long long seq, unsigned int* gticket, hipStream_t st);
hipError_t launch_fit_gather_batch(const BatchSlot* slots, int P, double* bout, int bout_stride, unsigned long long* flag,
                                   unsigned long long seq, unsigned int* gticket, hipStream_t st);
hipError_t launch_elim(const ElimArgs& a, const double* y, double* Winv, double* panels, double* Rinv, int ldr, double* gamma, double* scal,
                       double* coefw, int estimate_trend, int mode, double beta, double s2t_host, hipStream_t st);
hipError_t launch_nll_small(int kernel, bool grad, const NllSmallArgs& a, hipStream_t st);
hipError_t launch_fit_gather(const double* scal, const double* S, int nS, double* out_scal, double* out_S, unsigned long long* flag,
                             unsigned long long seq, hipStream_t st);
hipError_t launch_grad_coef(const double* scal, int n_t, int mode, int N, int krank, double s2t_host, double* coef, hipStream_t st);
hipError_t launch_gemv2(const double* M, int ld, int N, int tri, const double* x0, const double* x1, double* y0, double* y1,
                        double* scratch, hipStream_t st);
// Lfac != null: also scal[0] = sum(log(diag(Lfac))) (what launch_logdet writes, in the same launch)
hipError_t launch_fit_rho(const double* Yt, const double* Ft, int N, int estimate_trend, double beta, double* rho, double* scal,
                          hipStream_t st, const double* Lfac = nullptr, int ldL = 0, double* coef = nullptr, int mode = 0,
                          double s2t_host = 0.0);
hipError_t launch_grad_finish(const double* partial, int nblk, int nout, double* out, const double* Rinv, int ld, int nparts,
                              size_t part_stride, int N, const double* gamma, int with_trace, const double* scal, double* out_scal,
                              double* out_S, unsigned long long* flag, unsigned long long seq, unsigned int* ticket, hipStream_t st);
hipError_t launch_trace_gg(const double* Rinv, int ld, int nparts, size_t part_stride, int N, const double* gamma,
                           const double* qv, double* out, hipStream_t st);
hipError_t launch_point_hessian(const double* X, int N, int d, const double* theta, const double* x, const double* r,
                                const double* rdx, const double* gamma, double* H, hipStream_t st);
hipError_t launch_point_corr(int kernel, const double* X, int N, int d, const double* theta, const double* x,
                             double* r, double* rdx, hipStream_t st);

hipError_t launch_col_reduce(const double* r, const double* rt, int N, int B, const double* gamma, const double* wvec,
                             double* mu, double* wd, double* ss, hipStream_t st);
hipError_t launch_batch_corr(int kernel, const double* X, int N, int d, const double* theta, const double* Xb, int B,
                             double* r, double* s2, hipStream_t st);
hipError_t launch_batch_grad(int kernel, const double* X, int N, int d, const double* theta, const double* Xb, int B,
                             const double* r, const double* s2, const double* Z, const double* gamma,
                             const double* wvec, double* out, hipStream_t st);


// ---- one-point / B-point evaluation and the lock-step polish (kernels_point.hip) -------------------------------------
constexpr int BOGP_POINT_MAX_D = 320;   // = BOGP_MAX_DIM: input dimensions the finishing workgroup stages in LDS
constexpr int BOGP_POINT_ARG_D = 64;    // a single point of at most this many dimensions travels as a kernel argument
constexpr int POLISH_M = 8;             // curvature pairs of the lock-step L-BFGS
struct PointRhsArgs {
  const double* X;      // N x d training points, row-major
  const double* theta;  // d
  const double* Xb;     // B x d points (device), or null: ONE point in `x`
  double x[BOGP_POINT_ARG_D];
  double* rhs;          // [B][npass][Npp][NC]
  int N, d, Npp, npass;
};
struct PointTriArgs {
  const double* V;      // L^-1, column-major, leading dimension ld, exact zeros above the diagonal
  const double* gamma;  // Np
  const double* wvec;   // Np (zeros under simple kriging)
  const double* rhs;
  double* part;         // [B][npass][nRB + 1][2 NC] block records
  double* split_scratch;        // [B][npass][nRB + 1][nsplit][rb x NC] partial tiles of a split row block (nsplit > 1)
  unsigned int* split_counter;  // [B][npass][nRB + 1] arrival tickets of the splits, zero between launches
  int rb, nsplit;               // rows of V per workgroup (16 / 32 / 64); workgroups per row block
  unsigned int* counter;  // [B] arrival tickets, zero between launches
  double* out;          // [B][rec_stride]: mu, mse, acq (q), dmu (d), dmse (d), dacq (q x d)
  int ld, Npp, Nr32, nRB, npass, d;
  int rec_stride, q, want_dacq, estimate_trend, minimize;
  int acq_id[64];
  double acq_par[64];
  double plugin, beta, G, ftft, sigma2;
  const double* trend_rec;  // [B][2 + 2 d] = mu_t, uu, dmu_t, duu of a linear trend basis (k_point_trend_fin), or null: constant basis
  unsigned long long* done_flag;  // device-mapped pinned word k_point_finish stores `done_seq` into last (one-point calls), or null
  unsigned long long done_seq;
};
struct PolishArgs {
  double* state;        // [B][state_stride]
  const double* rec;    // the records k_point_tri wrote for the current trial points
  double* Xt;           // [B][d] trial points (read: the evaluated ones; written: the next ones)
  const double *lo, *hi;  // d each
  unsigned int* n_done;
  int d, q, rec_stride, state_stride, max_evals;
  double pgtol, factr_eps, first_step;
};
int point_columns_per_pass(int d);
int point_passes(int d);
void point_tri_geometry(int N, int d, int B, int* rb, int* nsplit);
hipError_t launch_point_rhs(int kernel, const PointRhsArgs& a, int B, hipStream_t st);
hipError_t launch_point_tri(const PointTriArgs& a, int B, hipStream_t st);
hipError_t launch_point_trend(const PointRhsArgs& ra, const double* W, int ldW, int p, const double* betav, const double* Sinv,
                              int estimate_trend, double* Tw, double* trec, int B, hipStream_t st);
// MFMA flavour of the B-point path (kernels_point.hip): ncp = point_mfma_columns(d) right-hand-side columns per point (0: d too large)
int point_mfma_columns(int d);
hipError_t launch_point_rhs_T(int kernel, const PointRhsArgs& a, int ncp, double* rT, long long Mc, int Np, int B, hipStream_t st);
hipError_t launch_point_gw(int ncp, const double* rT, long long Mc, const double* gamma, const double* wvec, int Nr32, int B, int nJ,
                           double* part, hipStream_t st);
hipError_t launch_point_finish_mfma(const PointTriArgs& a, int ncp, int B, hipStream_t st);
size_t polish_state_doubles(int d);
hipError_t launch_polish_step(const PolishArgs& a, int B, hipStream_t st);

}  // namespace bogp
