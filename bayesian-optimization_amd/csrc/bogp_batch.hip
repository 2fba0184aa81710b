// bogp_batch.hip -- P likelihood evaluations per device round trip (bogp_nll_batch) for the restart-parallel MLE (bogp_mle.cpp).
//
// Why: GaussianProcess._optimize_hyperparameter (gpr.py:1058-1197) runs `random_start` L-BFGS-B restarts of a few hundred
// likelihood + gradient evaluations each, one after the other, and at the sizes of an ordinary BO run ONE evaluation occupies one
// workgroup (k_nll_small, N <= 156) or a thin chain of launches (k_elim_*, N <= 2048): 255 of the 256 CUs idle, and `tell()` is
// 95 % of a BO loop (profiles/r03_bo_loop.txt).  The restarts are independent, so their evaluations can share a launch:
//   N <= 156   k_nll_small<.., BATCH>   grid = P workgroups, slot s = parameter vector s
//   N <= 2048  k_build_R_b / k_elim_*_b / k_grad_contract_b / k_grad_finish_b   the one-evaluation kernels with a grid dimension
//              over P workspaces (BatchSlot)
//   otherwise  (N > 2048, polynomial trends, several targets: evaluations that fill the GPU on their own, or rare) the P calls
//              of bogp_nll one after the other.
// Each slot runs the one-evaluation kernels' own block routines on its own parameters, so slot s returns the bits of the s-th
// sequential bogp_nll call (tests/test_gpu_nll_batch.py).  The batch has its own workspaces: the handle's factor buffers -- and with
// them a committed model -- are not touched on the two batched paths.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../../include/bogp.h"
#include "bogp_fit.h"
#include "bogp_handle.h"
#include "bogp_internal.h"

using namespace bogp;

namespace {

inline size_t up8(size_t n) { return (n + 7) & ~(size_t)7; }

// what the host decides per slot before anything is launched: theta (d + 1), k_build_R's arguments, the mode's scalars
struct SlotPrep {
  bool valid = false;
  FitPending fp;
  double a = 1, b = 1, diag = 1, div = 0, pexp = 0;
};

// the checks of `factorize` (bogp_api.hip) on one parameter vector; th receives theta (d) + the exponent (entry d)
bool prep_slot(int kernel, int mode, int d, const double* par, int n_par, double noise_var, int estimate_trend, double beta, int N,
               double* th, SlotPrep* sp) {
  int n_theta = n_par - (mode == BOGP_MODE_NOISELESS ? 0 : 1);
  double pexp = 0.0;
  // ONE predicate for the three paths of bogp_nll_batch (include/bogp.h: a non-finite or non-positive parameter -> BOGP_ERR_INVALID for that slot):
  // the sequential path below applies the same test before it calls bogp_nll
  for (int k = 0; k < n_par; ++k)
    if (!std::isfinite(par[k]) || !(par[k] > 0)) return false;
  if (kernel == BOGP_KERNEL_GENEXP || kernel == BOGP_KERNEL_MATERN_NU) {
    pexp = par[n_theta - 1];
    if (!(pexp > 0) || (kernel == BOGP_KERNEL_MATERN_NU && pexp > 60.0)) return false;
    n_theta -= 1;
  }
  for (int k = 0; k < d; ++k) {
    th[k] = par[n_theta == 1 ? 0 : k];
    if (!(th[k] > 0)) return false;
  }
  th[d] = pexp;
  FitPending& fp = sp->fp;
  fp.mode = mode; fp.estimate_trend = estimate_trend; fp.ptrend = 1; fp.n_t = 1; fp.N = N;
  fp.beta = beta; fp.alpha = 0; fp.sigma2_par = 0; fp.noise_var = noise_var; fp.s2t = 0;
  if (mode == BOGP_MODE_NOISELESS) {
    sp->div = 0; sp->a = 1.0; sp->b = 1.0; sp->diag = 1.0;
  } else if (mode == BOGP_MODE_NOISE_ESTIM) {
    fp.alpha = par[n_par - 1];
    sp->div = 0; sp->a = fp.alpha; sp->b = 1.0; sp->diag = fp.alpha * 1.0 + (1 - fp.alpha) * 1.0;
  } else {
    fp.sigma2_par = par[n_par - 1];
    fp.s2t = fp.sigma2_par + noise_var;
    sp->div = 1; sp->a = fp.sigma2_par; sp->b = fp.s2t; sp->diag = (fp.sigma2_par * 1.0 + noise_var * 1.0) / fp.s2t;
  }
  sp->pexp = pexp;
  sp->valid = true;
  return true;
}

// pinned, device-mapped staging of the batch: [cap doubles] (parameter rows, then result records) + the sequence word behind them
int ensure_pinned(bogp_handle* h, size_t doubles) {
  if (h->hbatch && h->hbatch_cap >= doubles) return BOGP_OK;
  if (h->hbatch) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipHostFree(h->hbatch));
    h->hbatch = h->hbatch_dev = nullptr;
    h->hbatch_cap = 0;
  }
  const size_t cap = std::max<size_t>(doubles, 16384);
  HIPCHK(h, hipHostMalloc((void**)&h->hbatch, (cap + 8) * sizeof(double), hipHostMallocMapped));
  HIPCHK(h, hipHostGetDevicePointer((void**)&h->hbatch_dev, h->hbatch, 0));
  memset(h->hbatch, 0, (cap + 8) * sizeof(double));
  h->hbatch_cap = cap;
  if (!h->dbatch_ticket) {
    HIPCHK(h, hipMalloc((void**)&h->dbatch_ticket, 2 * sizeof(unsigned int)));
    HIPCHK(h, hipMemset(h->dbatch_ticket, 0, 2 * sizeof(unsigned int)));
  }
  return BOGP_OK;
}

// doubles of one slot's workspace on the elimination path, and the offsets of its parts
struct SlotLayout {
  size_t E, Eb, yt, ft, logpart, Winv, panels, xpanel, Rinv, gamma, scal, partial, S, total;
};
SlotLayout slot_layout(int ld, int d, int N) {
  const size_t nb = (size_t)ld / 64, lde = (size_t)ld + 64;
  const size_t Np = up8(((size_t)N + 31) / 32 * 32);
  SlotLayout L;
  size_t o = 0;
  L.E = o; o += (size_t)ld * ld;
  L.Eb = o; o += (size_t)64 * ld;
  L.yt = o; o += ld;
  L.ft = o; o += ld;
  L.logpart = o; o += up8(nb + 1);
  L.Winv = o; o += (nb + 1) * 64 * 64;
  L.panels = o; o += 2 * lde * 64;
  L.xpanel = o; o += 4 * (nb + 1) * 64 * 64;  // four solved panels (grouped steps)
  L.Rinv = o; o += (size_t)ld * ld;
  L.gamma = o; o += Np;
  L.scal = o; o += 64;
  L.partial = o; o += up8((size_t)grad_contract_blocks(N) * (d + 1));
  L.S = o; o += up8((size_t)d + 4);
  L.total = o;
  return L;
}

size_t batch_max_bytes() {
  static const size_t mb = [] { const char* e = getenv("BOGP_BATCH_MAX_MB"); return e ? (size_t)atol(e) : (size_t)8192; }();
  return mb << 20;
}

// the workspaces of (at least) `P` slots for an (N, d) training set + their BatchSlot table on the device
int ensure_slots(bogp_handle* h, int P, int ld, int d, int N) {
  if (P <= h->bws_P && h->bws_ld == ld && h->bws_d == d && h->bws_N == N) return BOGP_OK;
  // (a table is built for the largest P seen at this shape, so that a batch that loses a slot to an invalid parameter vector --
  // or a caller that alternates batch sizes -- does not rebuild it)
  if (h->bws_ld == ld && h->bws_d == d && h->bws_N == N) P = std::max(P, h->bws_P);
  const SlotLayout L = slot_layout(ld, d, N);
  const size_t row = up8((size_t)d + 1) + 8;  // a parameter row: theta, exponent | a, b, diag, s2t
  const size_t need = (size_t)P * (L.total + row + 8);
  if (h->bws_cap < need) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    dfree(h->dbws);
    h->bws_cap = 0;
    h->bws_P = 0;
    // (a BO loop grows N by one point per tell(): a quarter of head room keeps the slab for ~60 iterations instead of re-allocating
    // every time a tile count ticks up)
    const size_t cap = need + need / 4;
    HIPCHK(h, hipMalloc((void**)&h->dbws, cap * sizeof(double)));
    h->bws_cap = cap;
  }
  if (h->bslots_cap < (size_t)P) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    dfree(h->dbslots);
    h->bslots_cap = 0;
    h->bws_P = 0;
    HIPCHK(h, hipMalloc((void**)&h->dbslots, (size_t)P * sizeof(BatchSlot)));
    h->bslots_cap = (size_t)P;
  }
  std::vector<BatchSlot> tab((size_t)P);
  double* rows = h->dbws + (size_t)P * L.total;
  double* tickets = rows + (size_t)P * row;  // one zeroed 8-byte word a slot (k_grad_finish_b's arrival counter)
  HIPCHK(h, hipMemsetAsync(tickets, 0, (size_t)P * 8 * sizeof(double), h->stream));
  for (int s = 0; s < P; ++s) {
    double* w = h->dbws + (size_t)s * L.total;
    BatchSlot& b = tab[(size_t)s];
    b.theta = rows + (size_t)s * row;
    b.par = b.theta + up8((size_t)d + 1);
    b.ea.E = w + L.E; b.ea.Eb = w + L.Eb; b.ea.ld = ld; b.ea.nb = ld / 64; b.ea.N = N;
    b.ea.yt = w + L.yt; b.ea.ft = w + L.ft; b.ea.logpart = w + L.logpart;
    b.Winv = w + L.Winv; b.panels = w + L.panels; b.xpanel = w + L.xpanel; b.Rinv = w + L.Rinv; b.gamma = w + L.gamma; b.scal = w + L.scal;
    b.ea.info = reinterpret_cast<int*>(b.scal + 62);
    b.partial = w + L.partial; b.S = w + L.S;
    b.ticket = reinterpret_cast<unsigned int*>(tickets + (size_t)s * 8);
  }
  HIPCHK(h, hipMemcpyAsync(h->dbslots, tab.data(), (size_t)P * sizeof(BatchSlot), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));  // (`tab` is pageable and leaves scope)
  h->bws_P = P; h->bws_ld = ld; h->bws_d = d; h->bws_N = N;
  h->bws_rows = rows;
  h->bws_row = row;
  return BOGP_OK;
}

}  // namespace

namespace bogp {
void batch_release(bogp_handle* h) {
  if (h->hbatch) (void)hipHostFree(h->hbatch);
  h->hbatch = h->hbatch_dev = nullptr;
  h->hbatch_cap = 0;
  dfree(h->dbatch_ticket);
  dfree(h->dbws);
  dfree(h->dbslots);
  h->bws_cap = h->bslots_cap = 0;
  h->bws_P = 0;
}
}  // namespace bogp

// One group of at most `Pg` slots that all take the same device path.  slot_of[i] = index into the caller's arrays.
static int run_group(bogp_handle* h, int path, int kernel, int mode, const std::vector<int>& slot_of, const double* par, int n_par,
                     double noise_var, int estimate_trend, double beta, const std::vector<SlotPrep>& prep, const std::vector<double>& theta,
                     double* llf, double* grad, int* info) {
  const int N = h->N, d = h->d, Pg = (int)slot_of.size();
  const bool want_grad = grad != nullptr;
  hipStream_t st = h->stream;
  const int n_theta = n_par - (mode == BOGP_MODE_NOISELESS ? 0 : 1) - ((kernel == BOGP_KERNEL_GENEXP || kernel == BOGP_KERNEL_MATERN_NU) ? 1 : 0);
  const bool iso = n_theta != d;
  const size_t out_stride = 64 + up8((size_t)d + 3);
  const size_t in_stride = path == BOGP_NLL_PATH_ONE_LAUNCH ? (size_t)NS_BPAR : up8((size_t)d + 1) + 8;
  {
    const int e = ensure_pinned(h, (size_t)Pg * (in_stride + out_stride));
    if (e) return e;
  }
  double* hin = h->hbatch;
  double* hout = h->hbatch + (size_t)Pg * in_stride;
  double* dout = h->hbatch_dev + (size_t)Pg * in_stride;
  unsigned long long* dflag = reinterpret_cast<unsigned long long*>(h->hbatch_dev + h->hbatch_cap);
  const void* hflag = h->hbatch + h->hbatch_cap;
  const unsigned long long seq = ++h->batch_seq;

  if (path == BOGP_NLL_PATH_ONE_LAUNCH) {
    for (int i = 0; i < Pg; ++i) {
      const int s = slot_of[(size_t)i];
      double* row = hin + (size_t)i * NS_BPAR;
      const double* th = theta.data() + (size_t)s * (d + 1);
      for (int k = 0; k < d; ++k) row[k] = th[k];
      for (int k = d; k < 64; ++k) row[k] = 0.0;
      const SlotPrep& sp = prep[(size_t)s];
      row[64] = sp.pexp; row[65] = sp.a; row[66] = sp.b; row[67] = sp.diag; row[68] = sp.fp.s2t; row[69] = sp.div;
      row[70] = row[71] = 0.0;
    }
    NllSmallArgs na;
    na.X = h->dX; na.y = h->dy_base; na.N = N; na.d = d;
    for (int k = 0; k < 64; ++k) na.theta[k] = 0.0;
    na.pexp = 0; na.a = na.b = na.diag = 1.0; na.div = 0; na.s2t_host = 0;
    na.estimate_trend = estimate_trend; na.mode = mode; na.beta = beta;
    na.out_scal = nullptr; na.out_S = nullptr;
    na.flag = dflag; na.seq = seq;
    na.bpar = h->hbatch_dev; na.bout = dout; na.bticket = h->dbatch_ticket; na.P = Pg; na.bout_stride = (int)out_stride;
    HIPCHK(h, launch_nll_small(kernel, want_grad, na, st));
  } else {  // BOGP_NLL_PATH_ELIM
    const int ld = h->ldr;
    {
      const int e = ensure_slots(h, Pg, ld, d, N);
      if (e) return e;
    }
    const size_t row = h->bws_row;
    for (int i = 0; i < Pg; ++i) {
      const int s = slot_of[(size_t)i];
      double* r = hin + (size_t)i * row;
      const double* th = theta.data() + (size_t)s * (d + 1);
      for (int k = 0; k <= d; ++k) r[k] = th[k];
      double* q = r + up8((size_t)d + 1);
      const SlotPrep& sp = prep[(size_t)s];
      q[0] = sp.a; q[1] = sp.b; q[2] = sp.diag; q[3] = sp.fp.s2t;
    }
    HIPCHK(h, hipMemcpyAsync(h->bws_rows, hin, (size_t)Pg * row * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(h, launch_build_R_batch(kernel, mode == BOGP_MODE_NOISY, h->dX, N, d, h->dbslots, Pg, ld, st));
    HIPCHK(h, launch_elim_batch(h->dbslots, Pg, ld, h->dy_base, estimate_trend, mode, beta, st));
    if (want_grad) {
      HIPCHK(h, launch_grad_contract_batch(kernel, h->dX, N, d, h->dbslots, Pg, h->Np, ld, st));
      HIPCHK(h, launch_grad_finish_batch(h->dbslots, Pg, grad_contract_blocks(N), d + 1, ld, N, mode == BOGP_MODE_NOISY ? 1 : 0, dout,
                                         (int)out_stride, dflag, seq, h->dbatch_ticket, st));
    } else {
      HIPCHK(h, launch_fit_gather_batch(h->dbslots, Pg, dout, (int)out_stride, dflag, seq, h->dbatch_ticket, st));
    }
  }
  {
    const int ew = fit_wait_on(h, hflag, seq);
    if (ew) return ew;
  }
  const int info2[2] = {0, 0};
  for (int i = 0; i < Pg; ++i) {
    const int s = slot_of[(size_t)i];
    const double* rec = hout + (size_t)i * out_stride;
    int iw = 0;
    memcpy(&iw, rec + 62, sizeof(iw));
    FitOut o;
    const int rc = factorize_finish(h, prep[(size_t)s].fp, iw, rec, info2, true, &o);
    info[s] = rc;
    llf[s] = (rc == BOGP_OK || rc == BOGP_ERR_LLF_POSITIVE) ? o.llf : std::numeric_limits<double>::quiet_NaN();
    if (rc == BOGP_ERR_HIP) return rc;
    if (want_grad) {
      double* g = grad + (size_t)s * n_par;
      if (rc == BOGP_OK) nll_gradient_from_sums(mode, iso, d, par + (size_t)s * n_par, n_par, 1, rec + 64, o.s2t, g);
      else for (int k = 0; k < n_par; ++k) g[k] = 0.0;
    }
  }
  return BOGP_OK;
}

namespace bogp {

// The handles a batch of P independent one-evaluation calls (bogp_nll, bogp_nll_restricted) is dealt over: the caller's + helpers of the library's
// own (own stream, own factor buffers, fed from the host copy of the training set), one host thread each.  An evaluation is a chain of small
// launches between larger ones; independent chains interleave on the device (profiles/r05_nll_two_handles.txt).
std::vector<bogp_handle*> nll_team(bogp_handle* h, int P) {
  const int N = h->N, d = h->d;
  // workers: 3 handles up to N = 4096 (a linear-trend evaluation at N = 1024: 0.97 -> 0.47 ms), 2 above (C5: three gain nothing over two and
  // cost another set of N^2 buffers); none below N = 192, where an evaluation is a handful of launches.  BOGP_NLL_WORKERS = 1: off.
  static const int workers_env = [] { const char* e = getenv("BOGP_NLL_WORKERS"); return e ? std::max(1, std::min(3, atoi(e))) : 3; }();
  const int workers = std::min(std::min(workers_env, N > 4096 ? 2 : 3), P);
  std::vector<bogp_handle*> team{h};
  if (workers > 1 && N >= 192 && !h->h_X.empty()) {
    for (int w = 1; w < workers; ++w) {
      if ((int)h->aux.size() < w) {
        if (h->aux_fail_valid && h->aux_fail_gen == h->train_gen) break;  // a helper could not be loaded with THIS training set: not tried again
        bogp_handle* a = nullptr;
        if (bogp_create(h->device, &a) != BOGP_OK) break;  // (no helper: fewer handles do it all)
        h->aux.push_back(a);
        h->aux_gen.push_back(0);
      }
      bogp_handle* a = h->aux[(size_t)w - 1];
      if (h->aux_gen[(size_t)w - 1] != h->train_gen) {
        if (bogp_set_train(a, h->h_X.data(), h->h_y.data(), N, d, h->n_t) != BOGP_OK) {
          // (device memory, most likely: a helper that stopped halfway through its N^2 buffers keeps none of them -- the primary handle may
          // need the room -- and the same allocation is not retried on every later batch call)
          for (size_t k = (size_t)w - 1; k < h->aux.size(); ++k) bogp_destroy(h->aux[k]);
          h->aux.resize((size_t)w - 1);
          h->aux_gen.resize((size_t)w - 1);
          h->aux_fail_valid = true;
          h->aux_fail_gen = h->train_gen;
          break;
        }
        h->aux_gen[(size_t)w - 1] = h->train_gen;
      }
      a->h_beta_fixed = h->h_beta_fixed;  // fixed coefficients of a polynomial basis, if any
      team.push_back(a);
    }
  }
  return team;
}

}  // namespace bogp

extern "C" int bogp_nll_batch(bogp_handle* h, int kernel, int mode, int P, const double* par, int n_par, double noise_var, int trend,
                              int estimate_trend, double beta, double* llf, double* grad, int* info) {
  if (!h) return BOGP_ERR_INVALID;
  if (!par || !llf || !info || n_par <= 0 || P <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_nll_batch: par / llf / info must be non-null, P and n_par > 0");
  if (!h->dX) FAIL(h, BOGP_ERR_INVALID, "no training set: call bogp_set_train first");
  if (kernel < 0 || kernel > BOGP_KERNEL_MATERN_NU) FAIL(h, BOGP_ERR_INVALID, "unknown kernel id %d", kernel);
  if (mode < 0 || mode > 2) FAIL(h, BOGP_ERR_INVALID, "unknown estimation mode %d", mode);
  if (trend < BOGP_TREND_CONSTANT || trend > BOGP_TREND_QUADRATIC) FAIL(h, BOGP_ERR_INVALID, "unknown trend id %d", trend);
  if (grad && (kernel == BOGP_KERNEL_CUBIC || kernel == BOGP_KERNEL_GENEXP || kernel == BOGP_KERNEL_MATERN_NU))
    FAIL(h, BOGP_ERR_UNSUPPORTED, "bogp_nll_batch: the cubic / generalized_exponential correlation has no theta-derivative (gpr.py:763-766)");
  const int N = h->N, d = h->d;
  {
    int n_theta = n_par - (mode == BOGP_MODE_NOISELESS ? 0 : 1);
    if (kernel == BOGP_KERNEL_GENEXP || kernel == BOGP_KERNEL_MATERN_NU) {
      if (n_theta != d + 1 && n_theta != 2) FAIL(h, BOGP_ERR_INVALID, "generalized_exponential: len(theta) = %d must be 2 or d + 1 = %d", n_theta, d + 1);
      n_theta -= 1;
    }
    if (n_theta != d && n_theta != 1) FAIL(h, BOGP_ERR_INVALID, "len(theta) = %d must be 1 or d = %d", n_theta, d);
  }
  HIPCHK(h, hipSetDevice(h->device));
  const int path = bogp_nll_path(N, d, trend, h->n_t);
  if (path == BOGP_NLL_PATH_GENERAL) {
    // evaluations above N = 2048, with a polynomial trend or with several targets: the sequential call, slot by slot -- on SEVERAL handles at
    // once when there are at least two slots (r05): an evaluation is a chain of small launches between larger ones, and independent chains
    // interleave on the device (C5: 15.4 -> 13.5 ms per evaluation, N = 4096: 4.4 -> 3.2, N = 3000: 2.7 -> 1.8; linear trend at N = 2048:
    // 1.88 -> 1.22; profiles/r05_nll_two_handles.txt).  Every slot is the sequential call's bits either way: same kernels, same launch
    // geometry, no cross-stream reduction.
    auto run_slot = [&](bogp_handle* hh, int s) -> int {
      double* g = grad ? grad + (size_t)s * n_par : nullptr;
      const double* p = par + (size_t)s * n_par;
      bool ok = true;
      for (int k = 0; k < n_par; ++k) ok = ok && std::isfinite(p[k]) && p[k] > 0;
      int rc = BOGP_ERR_INVALID;
      llf[s] = std::numeric_limits<double>::quiet_NaN();
      if (ok) rc = bogp_nll(hh, kernel, mode, p, n_par, noise_var, trend, estimate_trend, beta, &llf[s], g);
      info[s] = rc;
      if (rc != BOGP_OK && rc != BOGP_ERR_LLF_POSITIVE) llf[s] = std::numeric_limits<double>::quiet_NaN();
      if (rc != BOGP_OK && g) for (int k = 0; k < n_par; ++k) g[k] = 0.0;
      return rc;
    };
    auto fatal = [](int rc) { return rc == BOGP_ERR_HIP || rc == BOGP_ERR_UNSUPPORTED || rc == BOGP_ERR_NO_DEVICE; };
    std::vector<bogp_handle*> team = nll_team(h, P);
    const int W = (int)team.size();
    if (W == 1) {
      for (int s = 0; s < P; ++s) {
        const int rc = run_slot(h, s);
        if (fatal(rc)) return rc;
      }
      return BOGP_OK;
    }
    std::vector<int> rcs((size_t)W, BOGP_OK);
    std::vector<std::thread> threads;
    int started = 1;  // workers that have a thread (this thread is worker 0)
    try {  // (nothing may leave an extern "C" function as an exception, least of all with joinable threads behind it)
      threads.reserve((size_t)W - 1);
      for (int w = 1; w < W; ++w) {
        threads.emplace_back([&, w] {
          for (int s = w; s < P && !fatal(rcs[(size_t)w]); s += W) rcs[(size_t)w] = run_slot(team[(size_t)w], s);
        });
        ++started;
      }
    } catch (...) {
    }
    for (int s = 0; s < P && !fatal(rcs[0]); s += W) rcs[0] = run_slot(h, s);
    for (int w = started; w < W; ++w)  // a worker without a thread: its slots on the caller's handle, here
      for (int s = w; s < P && !fatal(rcs[(size_t)w]); s += W) rcs[(size_t)w] = run_slot(h, s);
    for (auto& t : threads) t.join();
    if (fatal(rcs[0])) return rcs[0];
    for (int w = 1; w < W; ++w)
      if (fatal(rcs[(size_t)w])) FAIL(h, rcs[(size_t)w], "bogp_nll_batch (helper handle %d): %s", w, bogp_last_error(team[(size_t)w]));
    return BOGP_OK;
  }
  std::vector<SlotPrep> prep((size_t)P);
  std::vector<double> theta((size_t)P * (d + 1));
  std::vector<int> valid;
  valid.reserve((size_t)P);
  for (int s = 0; s < P; ++s) {
    if (prep_slot(kernel, mode, d, par + (size_t)s * n_par, n_par, noise_var, estimate_trend, beta, N, theta.data() + (size_t)s * (d + 1), &prep[(size_t)s])) {
      valid.push_back(s);
    } else {  // the sequential call's BOGP_ERR_INVALID for this slot; the others go ahead
      info[s] = BOGP_ERR_INVALID;
      llf[s] = std::numeric_limits<double>::quiet_NaN();
      if (grad) for (int k = 0; k < n_par; ++k) grad[(size_t)s * n_par + k] = 0.0;
    }
  }
  // groups of slots bounded by the workspace budget (elimination path: ~2 ld^2 doubles a slot) and by one launch's grid
  size_t gmax = 4096;
  if (path == BOGP_NLL_PATH_ELIM) {
    const size_t per = slot_layout(h->ldr, d, N).total * sizeof(double);
    gmax = std::max<size_t>(1, std::min<size_t>(gmax, batch_max_bytes() / per));
  }
  for (size_t g0 = 0; g0 < valid.size(); g0 += gmax) {
    const std::vector<int> grp(valid.begin() + (long)g0, valid.begin() + (long)std::min(valid.size(), g0 + gmax));
    const int rc = run_group(h, path, kernel, mode, grp, par, n_par, noise_var, estimate_trend, beta, prep, theta, llf, grad, info);
    if (rc != BOGP_OK) return rc;
  }
  return BOGP_OK;
}
