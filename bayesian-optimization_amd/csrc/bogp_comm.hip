// bogp_comm.hip -- the ONE cross-rank step of the sweep behind the C ABI: candidate shards are swept
// independently (one process per GPU), then every rank contributes its q winners -- (value, GLOBAL index, point) records
// packed ON THE DEVICE from the sweep's own result buffers -- to one ncclAllGather over RCCL/xGMI, and every rank applies
// the same deterministic reduce (np.argmax over the concatenation: NaN maximal, ties -> lowest global index).
// RCCL has no MAXLOC and an (f64, i64) pair does not fit a 64-bit max-reducible key, hence gather-then-reduce; the
// payload is q (2 + d) doubles per rank (1.4 KB at q = 8, d = 20): latency-bound, issued once per ask(), never per tile.
//
// librccl is resolved at run time (dlopen, preferring a copy the process already holds, e.g. PyTorch's bundled one):
// libbogp.so itself has no link-time dependency on it, single-GPU clients never load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "bogp_handle.h"

using namespace bogp;

namespace {
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  std::string err;
};
Rccl g_rccl;

bool rccl_load() {
  if (g_rccl.AllGather) return true;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  void* lib = nullptr;
  for (const char* n : names)
    if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;  // a copy already in the process wins
  if (!lib)
    for (const char* n : names)
      if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!lib) {
    g_rccl.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : "");
    return false;
  }
  g_rccl.lib = lib;
#define SYM(field, name)                                          \
  g_rccl.field = (decltype(g_rccl.field))dlsym(lib, name);        \
  if (!g_rccl.field) {                                            \
    g_rccl.err = std::string("librccl lacks symbol ") + name;     \
    return false;                                                 \
  }
  SYM(GetUniqueId, "ncclGetUniqueId")
  SYM(CommInitRank, "ncclCommInitRank")
  SYM(CommDestroy, "ncclCommDestroy")
  SYM(GetErrorString, "ncclGetErrorString")
  SYM(CommCount, "ncclCommCount")
  SYM(CommUserRank, "ncclCommUserRank")
  SYM(AllGather, "ncclAllGather")
#undef SYM
  return true;
}

// a > b in np.argmax order over the concatenated global array: NaN beats every number, ties -> lower global index
inline bool record_better(double av, int64_t ai, double bv, int64_t bi) {
  const bool an = std::isnan(av), bn = std::isnan(bv);
  if (an != bn) return an;
  if (!an && av != bv) return av > bv;
  return ai < bi;
}
}  // namespace

#define NCCLCHK(h, expr)                                                                                              \
  do {                                                                                                                \
    ncclResult_t _r = (expr);                                                                                         \
    if (_r != ncclSuccess) FAIL(h, BOGP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
  } while (0)

// record n of the send buffer = [value, global index (bit pattern), x_0 .. x_{d-1}] of winner n; an empty slot of a top-k
// list (index INT64_MAX from the masked argmax) becomes (-inf, -1, NaN ...)
__global__ void k_pack_winners(const double* __restrict__ val, const int64_t* __restrict__ idx, int n, int64_t offset,
                               const double* __restrict__ Xs, int d, int with_points, double* __restrict__ out) {
  const int i = blockIdx.x;
  const int rec = 2 + (with_points ? d : 0);
  const int64_t li = idx[i];
  const bool empty = li == INT64_MAX || li < 0;
  double* o = out + (size_t)i * rec;
  if (threadIdx.x == 0) {
    o[0] = empty ? -INFINITY : val[i];
    const int64_t g = empty ? (int64_t)-1 : li + offset;
    o[1] = __longlong_as_double(g);
  }
  if (with_points)
    for (int k = threadIdx.x; k < d; k += blockDim.x) o[2 + k] = empty ? __builtin_nan("") : Xs[(size_t)li * d + k];
}

extern "C" int bogp_comm_unique_id(unsigned char* id_out) {
  if (!id_out) return BOGP_ERR_INVALID;
  if (!rccl_load()) return BOGP_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) return BOGP_ERR_HIP;
  static_assert(sizeof(id) == BOGP_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id_out, &id, sizeof(id));
  return BOGP_OK;
}

extern "C" int bogp_comm_init(bogp_handle* h, const unsigned char* id_in, int rank, int world) {
  if (!h) return BOGP_ERR_INVALID;
  if (!id_in || world < 1 || rank < 0 || rank >= world) FAIL(h, BOGP_ERR_INVALID, "bogp_comm_init: need an id and 0 <= rank < world (got %d, %d)", rank, world);
  if (h->comm) FAIL(h, BOGP_ERR_INVALID, "bogp_comm_init: the handle already has a communicator (bogp_comm_destroy first)");
  if (!rccl_load()) FAIL(h, BOGP_ERR_UNSUPPORTED, "bogp_comm_init: %s", g_rccl.err.c_str());
  HIPCHK(h, hipSetDevice(h->device));
  ncclUniqueId id;
  memcpy(&id, id_in, sizeof(id));
  ncclComm_t c = nullptr;
  NCCLCHK(h, g_rccl.CommInitRank(&c, world, id, rank));
  h->comm = c;
  h->comm_owned = true;
  h->comm_rank = rank;
  h->comm_world = world;
  return BOGP_OK;
}

extern "C" int bogp_comm_attach(bogp_handle* h, void* nccl_comm) {
  if (!h) return BOGP_ERR_INVALID;
  if (!nccl_comm) FAIL(h, BOGP_ERR_INVALID, "bogp_comm_attach: null communicator");
  if (h->comm) FAIL(h, BOGP_ERR_INVALID, "bogp_comm_attach: the handle already has a communicator");
  if (!rccl_load()) FAIL(h, BOGP_ERR_UNSUPPORTED, "bogp_comm_attach: %s", g_rccl.err.c_str());
  int world = 0, rank = 0;
  NCCLCHK(h, g_rccl.CommCount((ncclComm_t)nccl_comm, &world));
  NCCLCHK(h, g_rccl.CommUserRank((ncclComm_t)nccl_comm, &rank));
  h->comm = nccl_comm;
  h->comm_owned = false;
  h->comm_rank = rank;
  h->comm_world = world;
  return BOGP_OK;
}

extern "C" int bogp_comm_info(const bogp_handle* h, int* rank, int* world) {
  if (!h) return BOGP_ERR_INVALID;
  if (rank) *rank = h->comm ? h->comm_rank : 0;
  if (world) *world = h->comm ? h->comm_world : 0;
  return BOGP_OK;
}

extern "C" int bogp_comm_destroy(bogp_handle* h) {
  if (!h) return BOGP_ERR_INVALID;
  if (h->comm && h->comm_owned) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    (void)g_rccl.CommDestroy((ncclComm_t)h->comm);
  }
  h->comm = nullptr;
  h->comm_owned = false;
  h->comm_rank = 0;
  h->comm_world = 0;
  return BOGP_OK;
}

namespace bogp {
void comm_release(bogp_handle* h) {  // from bogp_destroy
  (void)bogp_comm_destroy(h);
  dfree(h->dxchg_send);
  dfree(h->dxchg_recv);
}
}  // namespace bogp

// ---- the deterministic reduces (host; a few hundred bytes) ----------------------------------------------------------
extern "C" int bogp_reduce_pairs(int R, int q, int d, const double* gathered, double* val, int64_t* gidx, double* x) {
  if (R <= 0 || q <= 0 || d < 0 || !gathered || !val || !gidx) return BOGP_ERR_INVALID;
  const int rec = 2 + d;
  for (int c = 0; c < q; ++c) {
    int b = 0;
    for (int r = 1; r < R; ++r) {
      const double* a = gathered + ((size_t)r * q + c) * rec;
      const double* w = gathered + ((size_t)b * q + c) * rec;
      int64_t ai, wi;
      memcpy(&ai, a + 1, 8);
      memcpy(&wi, w + 1, 8);
      if (record_better(a[0], ai, w[0], wi)) b = r;
    }
    const double* w = gathered + ((size_t)b * q + c) * rec;
    val[c] = w[0];
    memcpy(&gidx[c], w + 1, 8);
    if (x && d) memcpy(x + (size_t)c * d, w + 2, (size_t)d * sizeof(double));
  }
  return BOGP_OK;
}

extern "C" int bogp_merge_topk(int R, int q, int k, int d, const double* gathered, double* val, int64_t* gidx, double* x) {
  if (R <= 0 || q <= 0 || k <= 0 || d < 0 || !gathered || !val || !gidx) return BOGP_ERR_INVALID;
  const int rec = 2 + d;
  struct Ent {
    double v;
    int64_t i;
    const double* p;
  };
  std::vector<Ent> ent;
  for (int c = 0; c < q; ++c) {
    ent.clear();
    for (int r = 0; r < R; ++r)
      for (int s = 0; s < k; ++s) {
        const double* a = gathered + (((size_t)r * q + c) * k + s) * rec;
        int64_t ai;
        memcpy(&ai, a + 1, 8);
        if (ai >= 0) ent.push_back({a[0], ai, a});
      }
    std::stable_sort(ent.begin(), ent.end(), [](const Ent& a, const Ent& b) { return record_better(a.v, a.i, b.v, b.i); });
    for (int s = 0; s < k; ++s) {
      const size_t o = (size_t)c * k + s;
      if (s < (int)ent.size()) {
        val[o] = ent[s].v;
        gidx[o] = ent[s].i;
        if (x && d) memcpy(x + o * d, ent[s].p + 2, (size_t)d * sizeof(double));
      } else {
        val[o] = -INFINITY;
        gidx[o] = -1;
        if (x && d)
          for (int j = 0; j < d; ++j) x[o * d + j] = NAN;
      }
    }
  }
  return BOGP_OK;
}

// pack this rank's n winners on the device -> all-gather -> ONE read-back of R x n records
static int exchange(bogp_handle* h, const double* dval, const int64_t* didx, int n, int64_t index_offset, int with_points,
                    std::vector<double>* gathered) {
  if (!h->comm) FAIL(h, BOGP_ERR_INVALID, "no communicator: call bogp_comm_init / bogp_comm_attach first");
  if (with_points && (!h->dXs || h->M <= 0)) FAIL(h, BOGP_ERR_INVALID, "no candidates to read the winning points from");
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const int d = h->d, R = h->comm_world;
  const int rec = 2 + (with_points ? d : 0);
  const size_t nsend = (size_t)n * rec;
  int e;
  if ((e = ensure(h, &h->dxchg_send, &h->xchg_send_cap, nsend))) return e;
  if ((e = ensure(h, &h->dxchg_recv, &h->xchg_recv_cap, nsend * R))) return e;
  hipLaunchKernelGGL(k_pack_winners, dim3(n), 64, 0, st, dval, didx, n, index_offset, h->dXs, d, with_points, h->dxchg_send);
  HIPCHK(h, hipGetLastError());
  NCCLCHK(h, g_rccl.AllGather(h->dxchg_send, h->dxchg_recv, nsend, ncclDouble, (ncclComm_t)h->comm, st));
  gathered->resize(nsend * R);
  HIPCHK(h, hipMemcpyAsync(gathered->data(), h->dxchg_recv, nsend * R * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  return BOGP_OK;
}

extern "C" int bogp_exchange_argmax(bogp_handle* h, int64_t index_offset, double* best_val, int64_t* best_gidx, double* best_x) {
  if (!h) return BOGP_ERR_INVALID;
  if (!best_val || !best_gidx) FAIL(h, BOGP_ERR_INVALID, "bogp_exchange_argmax: best_val / best_gidx must be non-null");
  if (h->last_q <= 0 || !h->dbest_val) FAIL(h, BOGP_ERR_INVALID, "bogp_exchange_argmax: no sweep result on this handle (call bogp_sweep first)");
  const int q = h->last_q;
  std::vector<double> g;
  int rc = exchange(h, h->dbest_val, h->dbest_idx, q, index_offset, best_x != nullptr, &g);
  if (rc) return rc;
  return bogp_reduce_pairs(h->comm_world, q, best_x ? h->d : 0, g.data(), best_val, best_gidx, best_x);
}

extern "C" int bogp_exchange_topk(bogp_handle* h, int64_t index_offset, double* best_val, int64_t* best_gidx, double* best_x) {
  if (!h) return BOGP_ERR_INVALID;
  if (!best_val || !best_gidx) FAIL(h, BOGP_ERR_INVALID, "bogp_exchange_topk: best_val / best_gidx must be non-null");
  if (h->last_topk_q <= 0 || h->last_topk_k <= 0 || !h->dtopk_val) FAIL(h, BOGP_ERR_INVALID, "bogp_exchange_topk: no top-k result on this handle (call bogp_sweep_topk first)");
  const int q = h->last_topk_q, k = h->last_topk_k;
  std::vector<double> g;
  int rc = exchange(h, h->dtopk_val, h->dtopk_idx, q * k, index_offset, best_x != nullptr, &g);
  if (rc) return rc;
  return bogp_merge_topk(h->comm_world, q, k, best_x ? h->d : 0, g.data(), best_val, best_gidx, best_x);
}
