// bogp_api_sweep.hip -- the C ABI of libbogp.so (include/bogp.h), part 3 of 3: candidate sets (upload, lazy upload, device generators),
// the posterior / acquisition sweep (one launch for small training sets, chunked producer -> contraction -> acquisition otherwise),
// top-k, the one-point gradient / Hessian calls, timing and the debug / self-test exports.  Part 1 = bogp_api.hip, part 2 = bogp_api_fit.hip.
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
  constexpr int nblk_per_split = 8;  // (4 / 16 measured: profiles/r05_corr_split_ab.txt)
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
  const int small_m = 32;
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
  // (k_corr_chunk<K, PV>) and finished by ONE per-candidate launch (k_trend_small); (the tile products it replaces: git 3357059)
  // (a mean-only call of a trend-rows model keeps the producer the full call uses -- pv = 0 -- so that mu comes out bit-identical)
  const int pv = (!vx_model && h->p > 1 && h->estimate_trend) ? corr_trend_columns(h->p) : 0;
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
      const bool tiles128 = Mc % 128 == 0;
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

#if defined(CONTRACT_TRACE) || defined(CONTRACT_D_TRACE)
// (profiling builds only, `make EXTRA=-DCONTRACT_TRACE` / -DCONTRACT_D_TRACE: the stamps the last k_contract16<4> launch left -- tools/contract_trace.py)
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
