// bogp_handle.h -- the device state behind an opaque bogp_handle (include/bogp.h) and the error plumbing shared by the
// translation units that implement the C ABI (bogp_api.hip, bogp_comm.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <vector>

#include "../../include/bogp.h"
#include "bogp_internal.h"

struct bogp_handle;
namespace bogp {
void comm_release(bogp_handle* h);   // bogp_comm.hip: destroys an owned communicator, frees the exchange buffers
void point_release(bogp_handle* h);  // bogp_point.hip: frees the point-evaluation buffers
void batch_release(bogp_handle* h);  // bogp_batch.hip: frees the batched-likelihood staging and workspaces
std::vector<bogp_handle*> nll_team(bogp_handle* h, int P);  // bogp_batch.hip: the handles a batch of one-evaluation calls is dealt over
// bogp_point.hip: posterior, input-gradients and q criteria of B points through k_point_rhs + k_point_tri.  `Xb` is a HOST
// array (B x d).  Outputs (host, any may be null): mu, mse (B), dmu, dmse (B x d), acq (B x q), dacq (B x q x d).
int point_eval_host(bogp_handle* h, const char* who, const double* Xb, int B, int q, const int* acq_id, const double* acq_par,
                    double plugin, int minimize, double* mu, double* mse, double* dmu, double* dmse, double* acq, double* dacq);
}

struct bogp_handle {
  int device = 0;
  int n_cu = 256;  // compute units of the device (launch planning of the fused small-N sweep)
  hipStream_t stream = nullptr;
  unsigned int* dchain_flags = nullptr;  // 2 * (cap_ld / 64) words: hand-over flags of the resident diagonal chain (k_chol_chain)
  hipEvent_t ev_chol[2] = {nullptr, nullptr};  // look-ahead of the large-matrix Cholesky (second stream)
  hipStream_t stream_upd = nullptr;  // (r04 experiment, always null since r06: a CU-masked stream for the look-ahead update of the two-level Cholesky)
  hipStream_t stream2 = nullptr;  // producer stream: k_corr_chunk of chunk c+1 runs beside k_contract of chunk c
  std::string err;

  // training set.  The buffers are sized for cap_ld rows / cap_d columns / cap_nt targets and re-used by every
  // bogp_set_train that fits (a BO loop grows N by one point per tell(): no free / malloc of the N x N buffers per iteration)
  int cap_ld = 0, cap_d = 0, cap_nt = 0;
  int N = 0, d = 0, Np = 0;
  int ldr = 0;  // leading dimension of dR / dV / dRinv: N rounded up to 64 (identity padding, kernels_chol.hip)
  double *dX = nullptr, *dy = nullptr;
  // multi-target y (gpr.py:463,490,502-505): the factorisation is shared, the vectors exist once per target.  dy / dyt /
  // drho / dgamma above and `sigma2` below always point at the ACTIVE target (bogp_select_target) inside these slabs.
  int n_t = 1, target = 0;
  double *dy_base = nullptr, *dyt_base = nullptr, *drho_base = nullptr, *dgamma_base = nullptr;
  std::vector<double> sigma2_t, nv_t;  // committed, per target

  // factorisation workspace (column-major, ld = ldr)
  double *dR = nullptr, *dV = nullptr, *dU = nullptr, *dT = nullptr, *dRinv = nullptr;  // L, L^-1, L^-T, scratch, R^-1
  double* dones = nullptr;  // N ones (the constant trend basis)
  double* dgemv_scratch = nullptr;  // segment partials of launch_gemv2
  std::vector<double> h_theta;  // [theta (d + 1) | sqrt_theta (d + 1)]: the block uploaded by factorize
  double* ddinv = nullptr;  // ldr x 64: inverses of the diagonal blocks of the running factorisation (kernels_chol.hip)
  double *dyt = nullptr, *dft = nullptr, *drho = nullptr, *dtmp = nullptr;  // N each
  double *dgamma = nullptr, *dw = nullptr;                                  // Np each (zero padded)
  double *dtheta = nullptr, *dsqrt_theta = nullptr;                         // d each
  double* dscal = nullptr;                                                  // small scalar scratch
  int* dinfo = nullptr;
  double* dgrad_partial = nullptr;
  size_t grad_partial_cap = 0;
  double* dbatch = nullptr;
  size_t batch_cap = 0;

  // how the last factorisation formed R from the correlations (k_build_R's arguments): what bogp_commit's refinement of
  // gamma recomputes R with
  bool R_div = false;
  double R_a = 1.0, R_b = 1.0, R_diag = 1.0;

  // committed state
  bool committed = false;
  int kernel = 0, mode = 0, estimate_trend = 0;
  double beta = 0, G = 0, sigma2 = 0, noise_var = 0, llf = 0, ftft = 0;
  double* dXthT = nullptr;  // [d][Np]
  double* dXnorm = nullptr;  // [Np] squared norms of the columns of XthT (k_corr_mfma)
  // trend-rows path (p > 32 columns under universal kriging, kernels_fit.hip: k_pack_Vx): the packed extended factor, its row offsets
  double2* dVpx = nullptr;
  size_t vpx_cap = 0;
  double* dAtx = nullptr;  // W G^-1 (N x p)
  size_t atx_cap = 0;
  int vx_Ne = 0, vx_Nt = 0;  // 0: the committed model does not use the path
  double2* dVp = nullptr;   // [Np/16][Np/8][64]

  // candidates
  const double* dXs = nullptr;
  double* dXs_owned = nullptr;
  size_t xs_cap = 0;
  double* dbounds = nullptr;
  size_t bounds_cap = 0;
  double* dxform = nullptr;  // per dimension [scale id, precision, lo, hi] of bogp_candidates_set_transform, or null
  std::vector<double> h_xform;
  double* dsobol = nullptr;  // d x bits direction numbers (uint64 bit patterns)
  size_t sobol_cap = 0;
  int64_t M = 0;
  // bogp_candidates_upload_lazy: host rows that are copied chunk by chunk on `stream_copy` WHILE the sweep contracts the chunk before
  // (run_sweep); lazy_done = rows whose copy has been enqueued, ev_copy = recorded behind the last enqueued copy
  const double* hXs_lazy = nullptr;
  int64_t lazy_done = 0;
  hipStream_t stream_copy = nullptr;
  hipEvent_t ev_copy = nullptr;

  // sweep scratch
  double *drT[2] = {nullptr, nullptr}, *dmu_part[2] = {nullptr, nullptr}, *dw_part[2] = {nullptr, nullptr};
  double* dss_part = nullptr;
  size_t rT_cap[2] = {0, 0}, mu_part_cap[2] = {0, 0}, w_part_cap[2] = {0, 0}, ss_part_cap = 0;
  double *dblk_val = nullptr, *dmu_out = nullptr, *dmse_out = nullptr, *dacq_out = nullptr, *dbest_val = nullptr;
  int64_t *dblk_idx = nullptr, *dbest_idx = nullptr;
  unsigned int* dcounter = nullptr;  // arrival ticket of k_sweep_small's last workgroup (zero between launches)
  double* dtopk_val = nullptr;   // [q][k] winners of bogp_sweep_topk (device-resident between its passes)
  int64_t* dtopk_idx = nullptr;
  size_t topk_val_cap = 0, topk_idx_cap = 0;
  size_t blk_val_cap = 0, blk_idx_cap = 0, mu_out_cap = 0, mse_out_cap = 0, acq_out_cap = 0;

  // polynomial trend bases with p > 1 columns (linear / quadratic; the constant basis keeps its scalar fast path)
  int trend = BOGP_TREND_CONSTANT, p = 1;  // committed
  double reml_logdet_ftf = 0.0;  // log det(F^T F) of the basis `reml_ftf_basis` (REML value, p > 1); -1: none cached
  int reml_ftf_basis = -1;
  int tr_built = -1, tr_p = 0, ldp = 0;    // basis currently held in dF / sizes of the buffers below
  std::vector<double> h_beta_fixed;        // bogp_set_trend_beta: simple-kriging coefficients
  // bogp_nll_batch above N = 2048 (r05): a second handle -- own stream, own factor buffers -- evaluates every other slot on a second host
  // thread, so that one evaluation's chain of small launches runs beside the other's rank-128 updates.  The host copy of the training set
  // (as bogp_set_train received it) is what the second handle is fed from; `train_gen` tells it when to take it again.
  std::vector<double> h_X, h_y;
  unsigned long train_gen = 0;
  std::vector<bogp_handle*> aux;       // helper handles (up to 2)
  std::vector<unsigned long> aux_gen;  // the train_gen each was last fed at
  bool aux_fail_valid = false;         // a helper could not be loaded (device memory) at train_gen == aux_fail_gen: not retried until the training set changes
  unsigned long aux_fail_gen = 0;
  std::vector<double> h_betav, h_Sinv;     // committed beta (p) and (Ft^T Ft)^-1 (p x p, column-major) for bogp_gradient
  double *dF = nullptr, *dFt = nullptr, *dQ1 = nullptr, *dQ = nullptr;  // N x p, column-major, ld = N
  double* dWp = nullptr;                                                // Np x p: L^-T Ft, zero-padded rows
  bogp::GemmSplit gsplit = {nullptr, 0, nullptr, 0};  // split-K scratch of the small products (kernels_gemm.hip), allocated with the trend buffers
  double* dWpT = nullptr;    // pp x Np (pp = p rounded up to 128): W^T, zero rows in the padding -- column side of the k_mm128 trend product
  double* dSinvP = nullptr;  // pp x pp: (Ft^T Ft)^-1, zero padded
  double *dA[2] = {nullptr, nullptr}, *dAV[2] = {nullptr, nullptr}, *dAU[2] = {nullptr, nullptr};  // ldp x ldp (CholeskyQR2)
  double *dAw = nullptr, *dAT = nullptr;                                // ldp x 64, ldp x ldp scratch
  double *dGinv = nullptr, *dSinv = nullptr, *dbetav = nullptr, *dqty = nullptr;  // p x p, p x p, p, p
  int* dinfo2 = nullptr;
  double *dTt = nullptr, *dCS = nullptr, *duu = nullptr, *dmtrend = nullptr;  // per sweep chunk: Mc x p, Mc x p, Mc, Mc
  size_t Tt_cap = 0, CS_cap = 0, uu_cap = 0, mtrend_cap = 0;
  double* dtpart[2] = {nullptr, nullptr};  // [S][pv][Mc] slice sums of W^T r from the fused producer (p <= 32)
  size_t tpart_cap[2] = {0, 0};

  // cross-rank exchange (bogp_comm.hip): RCCL communicator (owned or borrowed), send / receive records on the device, and
  // what the last sweep left in dbest_* / dtopk_* for bogp_exchange_* to pack
  void* comm = nullptr;
  bool comm_owned = false;
  int comm_rank = 0, comm_world = 0;
  double *dxchg_send = nullptr, *dxchg_recv = nullptr;
  size_t xchg_send_cap = 0, xchg_recv_cap = 0;
  int last_q = 0, last_topk_q = 0, last_topk_k = 0;

  // one-point / B-point evaluation and the lock-step polish (kernels_point.hip, bogp_point.hip)
  double *dpt_rhs = nullptr, *dpt_part = nullptr, *dpt_out = nullptr, *dpt_Xb = nullptr, *dpt_state = nullptr, *dpt_box = nullptr;
  size_t pt_rhs_cap = 0, pt_part_cap = 0, pt_out_cap = 0, pt_Xb_cap = 0, pt_state_cap = 0, pt_box_cap = 0;
  unsigned int* dpt_counter = nullptr;  // [cap] arrival tickets (zero between launches) + 1 word: finished starts of the polish
  size_t pt_counter_cap = 0;
  double* dpt_split = nullptr;  // partial tiles of split row blocks (one-point latency mode of k_point_tri)
  double *dpt_tw = nullptr, *dpt_trec = nullptr;  // linear trend in the one-point path: W^T [r | dr/dx] per point, the trend records
  size_t pt_tw_cap = 0, pt_trec_cap = 0;
  unsigned int* dpt_splitc = nullptr;
  size_t pt_split_cap = 0, pt_splitc_cap = 0;
  // the likelihood's host traffic (r03): theta travels through a pinned staging block, the scalars / gradient sums come back through
  // a device-mapped pinned block that one gather kernel fills, completion read off a sequence word (bogp_api.hip: fit_readback)
  double* hfit = nullptr;      // pinned: [0, 2048) theta staging | [2048, 2112) the 64 scalars | [2112, 2112 + 512) gradient sums | [3000] sequence word
  double* hfit_dev = nullptr;  // its device address
  unsigned long long fit_seq = 0;
  unsigned int* dfin_ticket = nullptr;  // k_grad_finish: which workgroup finished last
  double* hpin = nullptr;  // pinned host buffer the finishing workgroup writes its records into (device-mapped)
  double* hpin_dev = nullptr;
  size_t hpin_cap = 0;
  unsigned long long pt_seq = 0;  // sequence number of the last one-point call (completion word behind its record)
  // bogp_nll_batch (bogp_batch.hip): pinned device-mapped staging (parameter rows in, records out; the sequence word sits behind
  // hbatch_cap doubles), the slot-completion ticket, and the P workspaces of the elimination path with their BatchSlot table
  double* hbatch = nullptr;
  double* hbatch_dev = nullptr;
  size_t hbatch_cap = 0;
  unsigned int* dbatch_ticket = nullptr;
  unsigned long long batch_seq = 0;
  double* dbws = nullptr;
  size_t bws_cap = 0;
  bogp::BatchSlot* dbslots = nullptr;
  size_t bslots_cap = 0;
  int bws_P = 0, bws_ld = 0, bws_d = 0, bws_N = 0;
  double* bws_rows = nullptr;  // [bws_P][bws_row] parameter rows on the device
  size_t bws_row = 0;

  // timing of the last sweep/predict
  std::vector<hipEvent_t> ev;
  double t_corr_ms = 0, t_contract_ms = 0, t_acq_ms = 0;
  int n_chunks = 0;
  bool timing_pending = false, timing_fused = false;  // event times not read back yet / of the one-launch small-N sweep
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
template <typename T>
static inline int ensure(bogp_handle* h, T** p, size_t* cap, size_t n) {
  if (*cap >= n && *p) return BOGP_OK;
  if (*p) HIPCHK(h, hipFree(*p));
  *p = nullptr;
  *cap = 0;
  HIPCHK(h, hipMalloc((void**)p, n * sizeof(T)));
  *cap = n;
  return BOGP_OK;
}
template <typename T>
static inline void dfree(T*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

