// bogp_point.hip -- C ABI of the one-point / B-point consumption path (kernels_point.hip): bogp_point_eval,
// bogp_point_eval_batch, bogp_gradient_batch, bogp_polish.  See include/bogp.h for the contracts.
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bogp_handle.h"

using namespace bogp;

namespace {

struct PointPlan {
  int NC, npass, Npp, nRB, Nr32, rec_stride, rb, nsplit;
};

PointPlan plan_of(const bogp_handle* h, int B, int q, bool want_dacq) {
  PointPlan p;
  point_tri_geometry(h->N, h->d, B, &p.rb, &p.nsplit);
  p.NC = point_columns_per_pass(h->d);
  p.npass = point_passes(h->d);
  p.Npp = h->ldr;                         // rows of the right-hand sides (zero beyond N)
  p.nRB = (h->N + p.rb - 1) / p.rb;       // row blocks of V (rows beyond N are identity padding: C = rhs = 0)
  p.Nr32 = (h->N + 31) / 32 * 32;         // <= Np: gamma / w are zero padded up to there
  p.rec_stride = 2 + q + 2 * h->d + (want_dacq ? q * h->d : 0);
  return p;
}

int check_common(bogp_handle* h, const char* who, int q, const int* acq_id, const double* acq_par) {
  if (!h->committed) FAIL(h, BOGP_ERR_INVALID, "%s: no committed model", who);
  if (h->kernel == BOGP_KERNEL_CUBIC || h->kernel == BOGP_KERNEL_GENEXP || h->kernel == BOGP_KERNEL_MATERN_NU)
    FAIL(h, BOGP_ERR_UNSUPPORTED, "%s: the cubic correlation has no input-derivative (corr_dx leaves it undefined in the reference, gpr.py:655-658)", who);
  if (h->p > 1 && h->trend != BOGP_TREND_LINEAR)
    FAIL(h, BOGP_ERR_UNSUPPORTED, "%s: the quadratic trend has no Jacobian in the reference either (trend.py:138-139)", who);
  if (h->n_t > 1) FAIL(h, BOGP_ERR_UNSUPPORTED, "%s: one target (several: bogp_predict + bogp_gradient per target)", who);
  if (h->d > BOGP_POINT_MAX_D) FAIL(h, BOGP_ERR_UNSUPPORTED, "%s: at most %d input dimensions", who, BOGP_POINT_MAX_D);
  if (q < 0 || q > BOGP_MAX_Q || (q > 0 && !acq_id)) FAIL(h, BOGP_ERR_INVALID, "%s: 0 <= q <= %d with non-null acq_id", who, BOGP_MAX_Q);
  for (int i = 0; i < q; ++i) {
    if (acq_id[i] < 0 || acq_id[i] > 3) FAIL(h, BOGP_ERR_INVALID, "unknown acquisition id %d", acq_id[i]);
    const bool zero_ok = acq_id[i] == BOGP_ACQ_EPSILON_PI;
    if (acq_id[i] != BOGP_ACQ_EI && (!acq_par || !(acq_par[i] > 0 || (zero_ok && acq_par[i] == 0))))
      FAIL(h, BOGP_ERR_INVALID, "acquisition parameter %d must be > 0 (the reference asserts alpha/epsilon/t > 0)", i);
  }
  return BOGP_OK;
}

int ensure_counters(bogp_handle* h, size_t n) {
  if (h->pt_counter_cap >= n && h->dpt_counter) return BOGP_OK;
  if (h->dpt_counter) HIPCHK(h, hipFree(h->dpt_counter));
  h->dpt_counter = nullptr;
  h->pt_counter_cap = 0;
  HIPCHK(h, hipMalloc((void**)&h->dpt_counter, n * sizeof(unsigned int)));
  HIPCHK(h, hipMemsetAsync(h->dpt_counter, 0, n * sizeof(unsigned int), h->stream));
  h->pt_counter_cap = n;
  return BOGP_OK;
}

int ensure_pinned(bogp_handle* h, size_t n) {
  if (h->hpin_cap >= n && h->hpin) return BOGP_OK;
  if (h->hpin) HIPCHK(h, hipHostFree(h->hpin));
  h->hpin = h->hpin_dev = nullptr;
  h->hpin_cap = 0;
  HIPCHK(h, hipHostMalloc((void**)&h->hpin, n * sizeof(double), hipHostMallocMapped));
  HIPCHK(h, hipHostGetDevicePointer((void**)&h->hpin_dev, h->hpin, 0));
  h->hpin_cap = n;
  return BOGP_OK;
}

// queue k_point_rhs + k_point_tri for B points; the records land in `out` (device or device-mapped host memory)
int queue_point_eval(bogp_handle* h, const PointPlan& pl, const double* dXb, const double* x_host, int B, int q, const int* acq_id,
                     const double* acq_par, double plugin, int minimize, bool want_dacq, double* out,
                     unsigned long long* done_flag = nullptr, unsigned long long done_seq = 0) {
  hipStream_t st = h->stream;
  int e;
  // Enough right-hand sides: C = V rhs is the candidate sweep's triangular product -- FP64 MFMA through k_contract16's
  // cross-product epilogue against the packed V of the commit (kernels_point.hip, "MFMA flavour") instead of k_point_tri's
  // FP64-VALU loop.  "Enough" = the (64-column tile, 256-column group of V) workgroups fill the GPU's 512 slots one and a half
  // times: a workgroup of the last group walks all N rows whatever B is (220 us at N = 2048), so below that the VALU path's
  // finer split is faster -- C3: 128 points 515 us (VALU) vs 582 (MFMA); C5: 32 points 6.1 ms vs 3.0, 128 points 24.7 vs 11.0
  // (profiles/r03_point_call_latency.txt).  BOGP_POINT_MFMA_MIN = that workgroup count (default 768; 0 = never).
  const int ncp = point_mfma_columns(h->d);
  const char* e_min = getenv("BOGP_POINT_MFMA_MIN");  // read per call: the tests run both flavours in one process
  const long long mfma_min = e_min ? atoll(e_min) : 768LL;
  if (dXb && ncp > 0 && h->dVp && h->p == 1 && mfma_min > 0 && (((long long)B * ncp + 63) / 64) * ((h->Np + 255) / 256) >= mfma_min) {
    const int Np = h->Np, nJ = (Np + 255) / 256;
    const long long Mc = ((long long)B * ncp + 63) / 64 * 64;
    if ((e = ensure(h, &h->drT[0], &h->rT_cap[0], (size_t)Np * Mc))) return e;
    if ((e = ensure(h, &h->dpt_part, &h->pt_part_cap, (size_t)B * (nJ + 1) * 2 * ncp))) return e;
    PointRhsArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.X = h->dX; ra.theta = h->dtheta; ra.Xb = dXb; ra.N = h->N; ra.d = h->d;
    HIPCHK(h, launch_point_rhs_T(h->kernel, ra, ncp, h->drT[0], Mc, Np, B, st));
    HIPCHK(h, launch_point_gw(ncp, h->drT[0], Mc, h->dgamma, h->dw, pl.Nr32, B, nJ, h->dpt_part, st));
    ContractArgs ka;
    memset(&ka, 0, sizeof(ka));
    ka.rT = h->drT[0]; ka.Vp = h->dVp; ka.ss_part = h->dpt_part; ka.Mc = Mc; ka.nMt = (int)(Mc / 64); ka.nJ = nJ;
    ka.NJ16 = Np / 16; ka.NKP = Np / 8; ka.cross_B = B;
    HIPCHK(h, launch_contract_cross(ka, ncp, st));
    PointTriArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.part = h->dpt_part; fa.out = out; fa.nRB = nJ; fa.npass = 1; fa.d = h->d;
    fa.rec_stride = pl.rec_stride; fa.q = q; fa.want_dacq = want_dacq ? 1 : 0; fa.estimate_trend = h->estimate_trend;
    fa.minimize = minimize;
    for (int i = 0; i < q; ++i) { fa.acq_id[i] = acq_id[i]; fa.acq_par[i] = acq_par ? acq_par[i] : 0.0; }
    fa.plugin = plugin; fa.beta = h->beta; fa.G = h->G; fa.ftft = h->ftft; fa.sigma2 = h->sigma2;
    fa.done_flag = done_flag; fa.done_seq = done_seq;
    HIPCHK(h, launch_point_finish_mfma(fa, ncp, B, st));
    return BOGP_OK;
  }
  if ((e = ensure(h, &h->dpt_rhs, &h->pt_rhs_cap, (size_t)B * pl.npass * pl.Npp * pl.NC))) return e;
  if ((e = ensure(h, &h->dpt_part, &h->pt_part_cap, (size_t)B * pl.npass * (pl.nRB + 1) * 2 * pl.NC))) return e;
  const size_t nslots = (size_t)B * pl.npass * (pl.nRB + 1);
  if ((e = ensure_counters(h, (size_t)std::max(B, 64) + 1))) return e;
  if (pl.nsplit > 1) {
    if ((e = ensure(h, &h->dpt_split, &h->pt_split_cap, nslots * pl.nsplit * pl.rb * pl.NC))) return e;
    if (h->pt_splitc_cap < nslots || !h->dpt_splitc) {
      if (h->dpt_splitc) HIPCHK(h, hipFree(h->dpt_splitc));
      h->dpt_splitc = nullptr;
      h->pt_splitc_cap = 0;
      HIPCHK(h, hipMalloc((void**)&h->dpt_splitc, nslots * sizeof(unsigned int)));
      HIPCHK(h, hipMemsetAsync(h->dpt_splitc, 0, nslots * sizeof(unsigned int), st));
      h->pt_splitc_cap = nslots;
    }
  }
  PointRhsArgs ra;
  memset(&ra, 0, sizeof(ra));
  ra.X = h->dX; ra.theta = h->dtheta; ra.Xb = dXb; ra.rhs = h->dpt_rhs;
  ra.N = h->N; ra.d = h->d; ra.Npp = pl.Npp; ra.npass = pl.npass;
  if (!dXb) memcpy(ra.x, x_host, (size_t)h->d * sizeof(double));
  HIPCHK(h, launch_point_rhs(h->kernel, ra, B, st));
  const double* trend_rec = nullptr;
  if (h->p > 1) {  // linear basis: (Ft^T L^-1) [r | dr/dx] and the trend's share of mu, MSE and their gradients (kernels_point.hip)
    if ((e = ensure(h, &h->dpt_tw, &h->pt_tw_cap, (size_t)B * pl.npass * h->p * pl.NC))) return e;
    if ((e = ensure(h, &h->dpt_trec, &h->pt_trec_cap, (size_t)B * (2 + 2 * h->d)))) return e;
    HIPCHK(h, launch_point_trend(ra, h->dWp, h->Np, h->p, h->dbetav, h->dSinv, h->estimate_trend, h->dpt_tw, h->dpt_trec, B, st));
    trend_rec = h->dpt_trec;
  }
  PointTriArgs ta;
  memset(&ta, 0, sizeof(ta));
  ta.V = h->dV; ta.gamma = h->dgamma; ta.wvec = h->dw; ta.rhs = h->dpt_rhs; ta.part = h->dpt_part;
  ta.counter = h->dpt_counter; ta.out = out;
  ta.split_scratch = h->dpt_split; ta.split_counter = h->dpt_splitc; ta.rb = pl.rb; ta.nsplit = pl.nsplit;
  ta.ld = h->ldr; ta.Npp = pl.Npp; ta.Nr32 = pl.Nr32; ta.nRB = pl.nRB; ta.npass = pl.npass; ta.d = h->d;
  ta.rec_stride = pl.rec_stride; ta.q = q; ta.want_dacq = want_dacq ? 1 : 0; ta.estimate_trend = h->estimate_trend;
  ta.minimize = minimize;
  for (int i = 0; i < q; ++i) { ta.acq_id[i] = acq_id[i]; ta.acq_par[i] = acq_par ? acq_par[i] : 0.0; }
  ta.plugin = plugin; ta.beta = h->beta; ta.G = h->G; ta.ftft = h->ftft; ta.sigma2 = h->sigma2;
  ta.done_flag = done_flag; ta.done_seq = done_seq; ta.trend_rec = trend_rec;
  HIPCHK(h, launch_point_tri(ta, B, st));
  return BOGP_OK;
}

}  // namespace

namespace bogp {

void point_release(bogp_handle* h) {
  dfree(h->dpt_rhs); dfree(h->dpt_part); dfree(h->dpt_out); dfree(h->dpt_Xb); dfree(h->dpt_state); dfree(h->dpt_box);
  dfree(h->dpt_counter); dfree(h->dpt_split); dfree(h->dpt_splitc); dfree(h->dpt_tw); dfree(h->dpt_trec);
  h->pt_tw_cap = h->pt_trec_cap = 0;
  h->pt_split_cap = h->pt_splitc_cap = 0;
  h->pt_rhs_cap = h->pt_part_cap = h->pt_out_cap = h->pt_Xb_cap = h->pt_state_cap = h->pt_box_cap = h->pt_counter_cap = 0;
  if (h->hpin) (void)hipHostFree(h->hpin);
  h->hpin = h->hpin_dev = nullptr;
  h->hpin_cap = 0;
}

static int point_eval_chunk(bogp_handle* h, const double* Xb, int B, int q, const int* acq_id, const double* acq_par, double plugin,
                            int minimize, double* mu, double* mse, double* dmu, double* dmse, double* acq, double* dacq) {
  hipStream_t st = h->stream;
  const int d = h->d;
  const bool want_dacq = dacq != nullptr && q > 0;
  const PointPlan pl = plan_of(h, B, q, want_dacq);
  const size_t nrec = (size_t)B * pl.rec_stride;
  int e;
  // the finishing workgroups write straight into pinned host memory: one stream synchronisation, no copy command
  if ((e = ensure_pinned(h, std::max<size_t>(nrec + 8, 1024)))) return e;
  const double* dXb = nullptr;
  if (B > 1 || d > BOGP_POINT_ARG_D) {
    if ((e = ensure(h, &h->dpt_Xb, &h->pt_Xb_cap, (size_t)B * d))) return e;
    HIPCHK(h, hipMemcpyAsync(h->dpt_Xb, Xb, (size_t)B * d * sizeof(double), hipMemcpyHostToDevice, st));
    dXb = h->dpt_Xb;
  }
  // One point (the BFGS loop's call): completion is read off a sequence word the finishing workgroup stores behind the record
  // (pinned memory, system-scope release) -- polling it costs less than a stream synchronisation (profiles/r03_point_call_latency.txt);
  // bounded: after ~2 ms of polling the ordinary synchronisation takes over.
  if (B == 1) {
    volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(h->hpin + nrec);
    const unsigned long long seq = ++h->pt_seq;
    unsigned long long* dflag = reinterpret_cast<unsigned long long*>(h->hpin_dev + nrec);
    if ((e = queue_point_eval(h, pl, dXb, Xb, B, q, acq_id, acq_par, plugin, minimize, want_dacq, h->hpin_dev, dflag, seq))) return e;
    bool seen = false;
    for (int spin = 0; spin < 400000; ++spin) {
      if (*flag == seq) { seen = true; break; }
      __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!seen) HIPCHK(h, hipStreamSynchronize(st));
  } else {
    if ((e = queue_point_eval(h, pl, dXb, Xb, B, q, acq_id, acq_par, plugin, minimize, want_dacq, h->hpin_dev))) return e;
    HIPCHK(h, hipStreamSynchronize(st));
  }
  for (int b = 0; b < B; ++b) {
    const double* o = h->hpin + (size_t)b * pl.rec_stride;
    if (mu) mu[b] = o[0];
    if (mse) mse[b] = o[1];
    if (acq) memcpy(acq + (size_t)b * q, o + 2, (size_t)q * sizeof(double));
    if (dmu) memcpy(dmu + (size_t)b * d, o + 2 + q, (size_t)d * sizeof(double));
    if (dmse) memcpy(dmse + (size_t)b * d, o + 2 + q + d, (size_t)d * sizeof(double));
    if (want_dacq) memcpy(dacq + (size_t)b * q * d, o + 2 + q + 2 * d, (size_t)q * d * sizeof(double));
  }
  return BOGP_OK;
}

int point_eval_host(bogp_handle* h, const char* who, const double* Xb, int B, int q, const int* acq_id, const double* acq_par,
                    double plugin, int minimize, double* mu, double* mse, double* dmu, double* dmse, double* acq, double* dacq) {
  int e = check_common(h, who, q, acq_id, acq_par);
  if (e) return e;
  if (!Xb || B <= 0) FAIL(h, BOGP_ERR_INVALID, "%s: null points or B <= 0", who);
  if (q > 0 && !acq && !dacq) FAIL(h, BOGP_ERR_INVALID, "%s: q > 0 needs acq or dacq", who);
  HIPCHK(h, hipSetDevice(h->device));
  const int d = h->d;
  // points are served in chunks whose right-hand sides (8 ld NC bytes per point and pass) stay below 256 MB
  const size_t per_point = (size_t)point_passes(d) * h->ldr * point_columns_per_pass(d) * sizeof(double);
  const int chunk = (int)std::max<size_t>(1, std::min<size_t>(4096, ((size_t)256 << 20) / per_point));
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = std::min(chunk, B - b0);
    e = point_eval_chunk(h, Xb + (size_t)b0 * d, nb, q, acq_id, acq_par, plugin, minimize, mu ? mu + b0 : nullptr, mse ? mse + b0 : nullptr,
                         dmu ? dmu + (size_t)b0 * d : nullptr, dmse ? dmse + (size_t)b0 * d : nullptr, acq ? acq + (size_t)b0 * q : nullptr,
                         dacq ? dacq + (size_t)b0 * q * d : nullptr);
    if (e) return e;
  }
  return BOGP_OK;
}

}  // namespace bogp

extern "C" int bogp_point_eval(bogp_handle* h, const double* x, int q, const int* acq_id, const double* acq_par, double plugin,
                               int minimize, double* mu, double* mse, double* dmu, double* dmse, double* acq) {
  if (!h) return BOGP_ERR_INVALID;
  if (!x || !mu || !mse || !dmu || !dmse) FAIL(h, BOGP_ERR_INVALID, "bogp_point_eval: null pointer");
  if (q > 0 && !acq) FAIL(h, BOGP_ERR_INVALID, "bogp_point_eval: q > 0 with a null acq");
  return point_eval_host(h, "bogp_point_eval", x, 1, q, acq_id, acq_par, plugin, minimize, mu, mse, dmu, dmse, acq, nullptr);
}

extern "C" int bogp_point_eval_batch(bogp_handle* h, const double* Xb, int B, int q, const int* acq_id, const double* acq_par,
                                     double plugin, int minimize, double* mu, double* mse, double* dmu, double* dmse, double* acq,
                                     double* dacq) {
  if (!h) return BOGP_ERR_INVALID;
  return point_eval_host(h, "bogp_point_eval_batch", Xb, B, q, acq_id, acq_par, plugin, minimize, mu, mse, dmu, dmse, acq, dacq);
}

extern "C" int bogp_gradient_batch(bogp_handle* h, const double* Xb, int B, double* dmu, double* dmse) {
  if (!h) return BOGP_ERR_INVALID;
  if (!Xb || !dmu || !dmse || B <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_gradient_batch: null pointer or B <= 0");
  return point_eval_host(h, "bogp_gradient_batch", Xb, B, 0, nullptr, nullptr, 0.0, 1, nullptr, nullptr, dmu, dmse, nullptr, nullptr);
}

extern "C" int bogp_polish(bogp_handle* h, const double* X0, int B, const double* lo, const double* hi, int acq_id, double acq_par,
                           double plugin, int minimize, int max_evals, double pgtol, double factr, double* Xout, double* fout,
                           int* n_evals) {
  if (!h) return BOGP_ERR_INVALID;
  int e = check_common(h, "bogp_polish", 1, &acq_id, &acq_par);
  if (e) return e;
  if (!X0 || !lo || !hi || !Xout || !fout || B <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_polish: null pointer or B <= 0");
  const int d = h->d;
  if (max_evals <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_polish: max_evals must be positive");
  double wmin = INFINITY;
  for (int k = 0; k < d; ++k) {
    if (!(hi[k] >= lo[k])) FAIL(h, BOGP_ERR_INVALID, "bogp_polish: lo[%d] > hi[%d]", k, k);
    if (hi[k] > lo[k]) wmin = std::min(wmin, hi[k] - lo[k]);
  }
  if (!std::isfinite(wmin)) wmin = 1.0;
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const PointPlan pl = plan_of(h, B, 1, true);
  const size_t ss = polish_state_doubles(d);
  if ((e = ensure(h, &h->dpt_Xb, &h->pt_Xb_cap, (size_t)B * d))) return e;
  if ((e = ensure(h, &h->dpt_out, &h->pt_out_cap, (size_t)B * pl.rec_stride))) return e;
  if ((e = ensure(h, &h->dpt_state, &h->pt_state_cap, (size_t)B * ss))) return e;
  if ((e = ensure(h, &h->dpt_box, &h->pt_box_cap, (size_t)2 * d))) return e;
  if ((e = ensure_counters(h, (size_t)std::max(B, 64) + 1))) return e;
  if ((e = ensure_pinned(h, std::max<size_t>((size_t)B * (d + 2) + 16, 1024)))) return e;
  unsigned int* dn_done = h->dpt_counter + (h->pt_counter_cap - 1);
  // starting points clipped into the box (scipy's L-BFGS-B projects x0 the same way)
  std::vector<double> xs((size_t)B * d), box(2 * (size_t)d), st0((size_t)B * ss, 0.0);
  for (int b = 0; b < B; ++b)
    for (int k = 0; k < d; ++k) xs[(size_t)b * d + k] = std::min(std::max(X0[(size_t)b * d + k], lo[k]), hi[k]);
  for (int k = 0; k < d; ++k) { box[k] = lo[k]; box[d + k] = hi[k]; }
  for (int b = 0; b < B; ++b) { st0[(size_t)b * ss + 1] = 1.0; st0[(size_t)b * ss + 7] = 1.0; }  // alpha = 1, first = 1
  HIPCHK(h, hipMemcpyAsync(h->dpt_Xb, xs.data(), xs.size() * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(h->dpt_box, box.data(), box.size() * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(h->dpt_state, st0.data(), st0.size() * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemsetAsync(dn_done, 0, sizeof(unsigned int), st));
  PolishArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.state = h->dpt_state; pa.rec = h->dpt_out; pa.Xt = h->dpt_Xb; pa.lo = h->dpt_box; pa.hi = h->dpt_box + d; pa.n_done = dn_done;
  pa.d = d; pa.q = 1; pa.rec_stride = pl.rec_stride; pa.state_stride = (int)ss; pa.max_evals = max_evals;
  pa.pgtol = pgtol; pa.factr_eps = factr * 2.220446049250313e-16; pa.first_step = 0.05 * wmin;
  unsigned int* hdone = (unsigned int*)(h->hpin);
  // every iteration = one batched evaluation of the B trial points + one optimiser step, all queued; the host looks at the
  // count of finished starts every 8 iterations (one 4-byte read-back) to stop early
  int it = 0;
  while (it < max_evals) {
    const int burst = std::min(8, max_evals - it);
    for (int s = 0; s < burst; ++s, ++it) {
      if ((e = queue_point_eval(h, pl, h->dpt_Xb, nullptr, B, 1, &acq_id, &acq_par, plugin, minimize, true, h->dpt_out))) return e;
      HIPCHK(h, launch_polish_step(pa, B, st));
    }
    HIPCHK(h, hipMemcpyAsync(hdone, dn_done, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    if (*hdone >= (unsigned int)B) break;
  }
  std::vector<double> fin((size_t)B * ss);
  HIPCHK(h, hipMemcpyAsync(fin.data(), h->dpt_state, fin.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  for (int b = 0; b < B; ++b) {
    const double* s = &fin[(size_t)b * ss];
    fout[b] = s[0];
    if (n_evals) n_evals[b] = (int)s[6];
    memcpy(Xout + (size_t)b * d, s + 8, (size_t)d * sizeof(double));
  }
  return BOGP_OK;
}
