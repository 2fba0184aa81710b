// kernels_elim.hip -- the fit path's ONE-PASS ELIMINATION at 64-block granularity (157 <= N <= 3072, constant trend): factor, inverse and
// solves of gpr.py:790-811 / 994-1038 as one block Gauss-Jordan sweep, for one evaluation (k_elim_step / k_elim_stepS) and for batches of
// P evaluations (k_elim_*_b).  Part of kernels_chol.hip until r05; split out in r06 (no code change).  The shared 64-block device routines
// are in bogp_chol_device.h.
#include <atomic>
#include <cstdlib>
#include "bogp_chol_device.h"

namespace bogp {

// =====================================================================================================================
// The likelihood's factor + inverse + solves as ONE elimination at 64-block granularity (157 <= N <= 1024, constant basis):
// the scheme of kernels_nllsmall.hip with 64 x 64 blocks and one workgroup a block.  E holds the block T(bi, bj), bi >= bj,
// of the bordered matrix [[R, .], [I, 0]] in place: R until block column bj is eliminated, then X(bj, bi)^T (X = L^-T) until step
// bi, then block (bi, bj) of -R^-1; one extra block row carries [y; 1] and ends as -(R^-1 y)^T, -(R^-1 1)^T.  A step k is ONE
// launch (k_chol_step's layout): every workgroup forms the two panel blocks it needs itself from the copied-out raw panel,
// X_i = M_i W_k^T, X_j = M_j W_k^T (M_k = I), and applies T <- (bi == k or bj == k ? 0 : T) - X_i X_j^T; the blocks of column k + 1
// (as they are) and of row k + 1 (transposed) go into the next raw panel, and the workgroup of block (k + 1, k + 1) factors and
// inverts it for the next step.  It replaces the Cholesky steps, the recursive-doubling inverse, U U^T and both matrix-vector
// passes: nb launches of ~27 us instead of ~4 nb + 8 kernels.
// =====================================================================================================================
namespace {
// tile (bi, bj) of the state: block row nb (the right-hand sides) lives in its own 64 x ld array
__device__ __forceinline__ double* elim_tile(const ElimArgs& a, int bi, int bj, int& ldt) {
  if (bi < a.nb) {
    ldt = a.ld;
    return a.E + (size_t)bj * CB * a.ld + (size_t)bi * CB;
  }
  ldt = CB;
  return a.Eb + (size_t)bj * CB * CB;
}
// the identity rows of block row kb in a raw panel (M_kb = I: the solved block of the pivot row is W_kb^T itself)
__device__ __forceinline__ void elim_identity_rows(double* __restrict__ Pn, int lde, int kb, int tid) {
  for (int e = tid; e < CB * CB; e += 256) {
    const int r = e & 63, col = e >> 6;
    Pn[(size_t)col * lde + (size_t)kb * CB + r] = r == col ? 1.0 : 0.0;
  }
}
// the diagonal block of an elimination step: diag_pipe + sum(log diag L) + info.  ALONE = true (k_elim_diag_b, the first block): also the
// identity rows of block row kb in the raw panel; otherwise the workgroup of block (kb + 1, kb) of the same launch writes them
// (elim_store_plain) and k_elim_first has filled the constant part of every W_k -- r05: the chain of diagonal blocks is what an
// evaluation waits for, and these 32 stores a lane were 1.7 us of its 25 us a step (tools/probes/run_variants.sh)
template <bool ALONE>
__device__ __forceinline__ void elim_diag2(const double* cs, double* scr, double* __restrict__ Wn, double* __restrict__ logpart,
                                           int* __restrict__ info, int base, int reset, int nlive, double* __restrict__ Pn, int lde,
                                           int kb, int tid) {
  if (ALONE) elim_identity_rows(Pn, lde, kb, tid);
  double ls = 0.0;
  const int bad = diag_pipe<false>(cs, scr, Wn, nullptr, 0, nlive, tid, &ls);
  if (tid == 0) {
    *logpart = ls;
    if (reset) *info = bad;
    else if (bad != 0 && *info == 0) *info = base + bad;
  }
}

}  // namespace

// workgroup 0: the first diagonal block; workgroups i >= 1: block (i, 0) into the raw panel as it is
__device__ __forceinline__ void elim_first_block(const ElimArgs& a, double* __restrict__ W0, double* __restrict__ Pn) {
  __shared__ __attribute__((aligned(16))) double cs[CB * (CB + 1)];
  __shared__ __attribute__((aligned(16))) double sb[ED_LDS];
  const int tid = threadIdx.x, bi = blockIdx.x;
  const int lde = a.ld + CB;
  int ldt;
  const double* T = elim_tile(a, bi, 0, ldt);
  if (bi == 0) {
    for (int e = tid; e < CB * CB; e += 256) {
      const int r = e & 63, c = e >> 6;
      cs[r * (CB + 1) + c] = T[(size_t)c * ldt + r];
    }
    __syncthreads();
    elim_diag2<true>(cs, sb, W0, a.logpart, a.info, 0, 1, max(0, min(CB, a.N)), Pn, lde, 0, tid);
  } else {
    for (int e = tid; e < CB * CB; e += 256) {
      const int r = e & 63, c = e >> 6;
      Pn[(size_t)c * lde + (size_t)bi * CB + r] = T[(size_t)c * ldt + r];
    }
    // the constant part of W_{bi-1} (zeros above the diagonal, identity padding), for every diagonal block of this evaluation
    diag_fill(W0 + (size_t)(bi - 1) * CB * CB, nullptr, 0, max(0, min(CB, a.N - CB * (bi - 1))), tid);
  }
}
__global__ __launch_bounds__(256) void k_elim_first(const ElimArgs a, double* __restrict__ W0, double* __restrict__ Pn) {
  elim_first_block(a, W0, Pn);
}
// bogp_nll_batch flavours of the three elimination kernels: blockIdx.y = the parameter vector, whose state / panels / factors come
// from its BatchSlot; the block routines are the one-evaluation kernels' own
__global__ __launch_bounds__(256) void k_elim_first_b(const BatchSlot* __restrict__ slots) {
  const BatchSlot& sl = slots[blockIdx.y];
  elim_first_block(sl.ea, sl.Winv, sl.panels);
}

#ifdef ELIM_PROFILE
// (profiling builds only, `make EXTRA=-DELIM_PROFILE`: wall-clock stamps (100 MHz, common to all CUs) of the fused step's workgroups --
// row k: [0..7] the workgroup of the next diagonal block, [8..9] entry / exit of workgroup 0, [10..11] of the last workgroup;
// tools/probes/elim_stamps.py)
__device__ unsigned long long g_elim_stamps[64 * 16];
#define ESTAMP(cond_, k_, slot_)                                                                      \
  if ((cond_) && threadIdx.x == 0) g_elim_stamps[((k_) & 63) * 16 + (slot_)] = wall_clock64();
#else
#define ESTAMP(cond_, k_, slot_)
#endif
// the updated block back into the state; the blocks of column / row k + 1 into the next raw panel; block (k + 1, k + 1) factored and
// inverted for the next step.  Shared by the fused step (k_elim_step) and the split one (k_elim_update_b).
__device__ __forceinline__ bool elim_store_plain(const ElimArgs& a, int k, int bi, int bj, const double (&acc)[4][4], double* __restrict__ Tb,
                                                 int ldt, double* __restrict__ Pnext) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lk = lane >> 4;
  const int lde = a.ld + CB;
  const int i0 = CB * bi, j0 = CB * bj;

#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) Tb[(size_t)(16 * mi + 4 * t + lk) * ldt + 16 * w + (lane & 15)] = -acc[mi][t];
  const int kn = k + 1;
  if (kn >= a.nb) return false;
  if (bj == kn && bi > kn) {  // column kn, as it is
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t) Pnext[(size_t)(16 * mi + 4 * t + lk) * lde + i0 + 16 * w + (lane & 15)] = -acc[mi][t];
    if (bi == kn + 1) elim_identity_rows(Pnext, lde, kn, tid);  // (for the workgroup of the diagonal block, which has the chain to carry)
  } else if (bi == kn && bj < kn) {  // row kn, transposed: raw row block bj
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t) Pnext[(size_t)(16 * w + (lane & 15)) * lde + j0 + 16 * mi + 4 * t + lk] = -acc[mi][t];
  }
  return bi == kn && bj == kn;  // the next diagonal block: the caller stages and factors it (elim_stage_diag, elim_diag2)
}
// the next diagonal block out of the accumulators into the 64 x 65 staging elim_diag2 reads
__device__ __forceinline__ void elim_stage_diag(const double (&acc)[4][4], double* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lk = lane >> 4;
  __syncthreads();  // every wave is done with what the tile held
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) lds[(16 * w + (lane & 15)) * (CB + 1) + 16 * mi + 4 * t + lk] = -acc[mi][t];
  __syncthreads();
}
__device__ __forceinline__ void elim_store_block(const ElimArgs& a, int k, int bi, int bj, double (&acc)[4][4], double* __restrict__ Tb, int ldt,
                                                 double* lds, double* sb, double* __restrict__ Pnext, double* __restrict__ Wn) {
  const int kn = k + 1;
  // the next diagonal block goes straight to its factorisation: its tile in the state is dead (step kn restarts the block from zero; only
  // k_elim_diag_b reads it from the state, behind the super-tile update's own stores), so the 16 stores a lane are off the chain
  if (!(kn < a.nb && bi == kn && bj == kn)) {
    elim_store_plain(a, k, bi, bj, acc, Tb, ldt, Pnext);
    return;
  }
  ESTAMP(true, k, 4)
  elim_stage_diag(acc, lds);
  ESTAMP(true, k, 5)
  elim_diag2<false>(lds, sb, Wn, a.logpart + kn, a.info, CB * kn, 0, max(0, min(CB, a.N - CB * kn)), Pnext, a.ld + CB, kn, threadIdx.x);
}

// XROW >= 0 (the group chain of a batch, k_elim_substep_b): this workgroup also leaves the solved block of block row XROW (= bi or bj) in
// `Xout` (the layout of k_elim_panel_b) and, for the right-hand sides' row, Yt / Ft of block k -- what the separate panel launch would have
// written for that row
__device__ __forceinline__ void elim_step_core(const ElimArgs& a, int k, int bi, int bj, const double* __restrict__ Wk, const double* __restrict__ Pcur,
                                               double* __restrict__ Pnext, double* __restrict__ Wn, int xrow, double* __restrict__ Xout,
                                               double* lds, double* sb) {  // lds: CB * CPITCH doubles, sb: ED_LDS doubles of the workgroup's LDS
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lk = lane >> 4;
  const int lde = a.ld + CB;
  const int i0 = CB * bi, j0 = CB * bj;
  const bool restart = bi == k || bj == k;
  const bool dg_ = bi == k + 1 && bj == k + 1;
  // the workgroup every other one of the NEXT step waits for: its waves issue ahead of the neighbours it shares a CU with
  if (dg_) __builtin_amdgcn_s_setprio(3);
  ESTAMP(dg_, k, 0)
  ESTAMP(blockIdx.x == 0, k, 8)
  ESTAMP(blockIdx.x == gridDim.x - 1, k, 10)

  stage_aside(lds, as_global(Wk), CB, tid);  // tile[kk][c] = W(c, kk)
  double bv[16], bvi[16];
  load_bside(bv, as_global(Pcur + j0), lde, w, lane);
  if (bi != bj) load_bside(bvi, as_global(Pcur + i0), lde, w, lane);
  double xj[4][4], xi[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) xj[mi][t] = xi[mi][t] = 0.0;
  int ldt;
  double* __restrict__ Tb = elim_tile(a, bi, bj, ldt);
  double acc[4][4];  // negated tile
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mi][t] = restart ? 0.0 : -Tb[(size_t)(16 * mi + 4 * t + lk) * ldt + 16 * w + (lane & 15)];
  __syncthreads();
  ESTAMP(dg_, k, 1)
  mma_64<true>(lds, bv, xj, lane);  // X_j = M_j W^T: rows 16 w .. of block row bj, element (row, col 16 mi + 4 t + lk)
  if (bi != bj) {
    mma_64<true>(lds, bvi, xi, lane);
  } else {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t) xi[mi][t] = xj[mi][t];
  }
  if (bi == a.nb && (bj == k || xrow == a.nb) && w == 0 && (lane & 15) < 2) {  // rows 0 / 1 of the solved right-hand sides: Yt, Ft of block k
    double* dst = (lane & 15) == 0 ? a.yt : a.ft;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t) dst[CB * k + 16 * mi + 4 * t + lk] = xi[mi][t];
  }
  if (xrow >= 0) {  // element (row, col) of X_xrow at Xout[row + 64 col]
    const bool from_i = xrow == bi;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t) Xout[(size_t)(16 * mi + 4 * t + lk) * CB + 16 * w + (lane & 15)] = from_i ? xi[mi][t] : xj[mi][t];
  }
  __syncthreads();  // every wave is done with the W tile
  // A side of the update: tile[kk][c] = X_j(c, kk)
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) lds[(16 * mi + 4 * t + lk) * CPITCH + 16 * w + (lane & 15)] = xj[mi][t];
  // B side: X_i(row 16 w + (lane & 15), kk = 4 ks + lk) is exactly xi[ks / 4][ks % 4] of this lane
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) bv[4 * mi + t] = xi[mi][t];
  asm volatile("s_nop 7\n\ts_nop 7"
               : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]), "+v"(bv[8]),
                 "+v"(bv[9]), "+v"(bv[10]), "+v"(bv[11]), "+v"(bv[12]), "+v"(bv[13]), "+v"(bv[14]), "+v"(bv[15]));
  __syncthreads();
  ESTAMP(dg_, k, 2)
  mma_64(lds, bv, acc, lane);  // -T_new = -T_old + X_i X_j^T
  ESTAMP(dg_, k, 3)
  elim_store_block(a, k, bi, bj, acc, Tb, ldt, lds, sb, Pnext, Wn);
  ESTAMP(dg_, k, 6)
  ESTAMP(blockIdx.x == 0, k, 9)
  ESTAMP(blockIdx.x == gridDim.x - 1, k, 11)
}
__device__ __forceinline__ void elim_step_block(const ElimArgs& a, int k, const double* __restrict__ Wk, const double* __restrict__ Pcur,
                                                double* __restrict__ Pnext, double* __restrict__ Wn) {
  int bi, bj;
  __shared__ __attribute__((aligned(16))) double lds[CB * CPITCH];
  __shared__ __attribute__((aligned(16))) double sb[ED_LDS];
  int b = (int)blockIdx.x;
  const int nblk = (a.nb + 1) * (a.nb + 2) / 2 - 1;
  if ((int)gridDim.x > nblk) {
    // launch_elim's layout for one evaluation above 512 blocks a step -- 31 / 32 block rows (gridDim.x = blocks + 2; at 299 blocks it LOSES: 592 -> 620 us): the workgroup of the next diagonal block is
    // dispatched FIRST and the two workgroups the dispatcher would put on its CU after it (it fills the CUs round by round: 256 and 512 land where
    // 0 did) do nothing -- the diagonal block has its CU to itself (r05: operands in 1.4 - 2.8 us instead of 4 - 5, factorisation 12 us instead of 14 - 17)
    const int D = (k + 1) * (k + 2) / 2 + k + 1;  // its place in the triangular order (>= nblk in the last step: no such block)
    if (b == 256 || b == 512) return;
    b -= (b > 256) + (b > 512);                   // 0 .. nblk - 1
    if (b >= nblk) return;                        // (a grid below 513 holds one idle workgroup only)
    if (D < nblk) b = b == 0 ? D : (b <= D ? b - 1 : b);
  }
  tri_index(b, bi, bj);  // bi <= nb (block row nb = the right-hand sides); (nb, nb) is not launched
  elim_step_core(a, k, bi, bj, Wk, Pcur, Pnext, Wn, -1, nullptr, lds, sb);
}
__global__ __launch_bounds__(256) void k_elim_step(const ElimArgs a, int k, const double* __restrict__ Wk, const double* __restrict__ Pcur,
                                                   double* __restrict__ Pnext, double* __restrict__ Wn) {
  elim_step_block(a, k, Wk, Pcur, Pnext, Wn);
}
__global__ __launch_bounds__(256) void k_elim_step_b(const BatchSlot* __restrict__ slots, int k) {
  const BatchSlot& sl = slots[blockIdx.y];
  const size_t lde = (size_t)sl.ea.ld + CB;
  double* P0 = sl.panels;
  double* P1 = sl.panels + lde * CB;
  elim_step_block(sl.ea, k, sl.Winv + (size_t)k * CB * CB, (k & 1) ? P1 : P0, (k & 1) ? P0 : P1, sl.Winv + (size_t)(k + 1) * CB * CB);
}

// ---- the fused step on ROW PAIRS (r05: one evaluation with 14 .. 30 block rows, N = 833 .. 1920) ------------------------------------------------
// k_elim_step at N = 2048 is 560 workgroups on 256 CUs: two or three tenants a CU, each with three 64^3 products, and the workgroup of the
// next diagonal block -- the one the next step waits for -- shares its CU's matrix pipe, LDS and memory queue with them: 24.7 - 29.5 us a
// step against 19.3 at N = 1024, where every workgroup has a CU to itself (profiles/r05_elim_chain.txt).  Here a workgroup owns the two
// blocks (r0, c), (r0 + 1, c) of a pair of block rows: the solved block of column c is formed once for both (2 or 2.5 products a block
// instead of 3, the zero half of W skipped), every operand is requested before the first product, and the next diagonal block is left out
// and taken by workgroup 0 ALONE on its CU through the ordinary block routine (its operands arrive in 1.4 us instead of 4 - 5).  ~290
// workgroups at N = 2048.  Every block sees the same mma_64 calls on the same operands as in k_elim_step: the same bits.
// (Measured first, and replaced: 2 x 2 super-tiles -- 154 workgroups of 256 threads, every product behind its own global load, 36 us a
// step; of 512 threads in two groups, 24 us: the matrix-pipe work of a step sits on 154 of the 256 CUs.)
// workgroup q >= 0 of the pair grid: pair BI holds the block rows r0 = 2 BI - o, r0 + 1 (o = 1 when nb is even: row 0 alone, so that the
// last pair is (nb - 1, nb)), column c <= min(r0 + 1, nb - 1)
__device__ __forceinline__ void elim_pair_index(int q, int o, int& BI, int& c) {
  if (o) {
    BI = (int)sqrt((double)q);
    while ((BI + 1) * (BI + 1) <= q) ++BI;
    while (BI * BI > q) --BI;
    c = q - BI * BI;
  } else {
    BI = (int)((sqrt(4.0 * q + 1.0) - 1.0) * 0.5);
    while ((BI + 1) * (BI + 2) <= q) ++BI;
    while (BI * (BI + 1) > q) --BI;
    c = q - BI * (BI + 1);
  }
}
__device__ __forceinline__ int elim_pair_grid_dev(int nb) {  // = elim_pair_grid(nb) in closed form
  const int o = (nb & 1) ? 0 : 1;
  const int last = (nb + o) / 2;  // the last pair: (nb - 1, nb)
  return (o ? last * last : last * (last + 1)) + nb;
}
static int elim_pair_grid(int nb) {  // workgroups of the pair grid: sum over the pairs of min(r0 + 2, nb)
  const int o = (nb & 1) ? 0 : 1;
  int tot = 0;
  for (int BI = 0; 2 * BI - o <= nb; ++BI) tot += min(2 * BI - o + 2, nb);
  return tot;
}
__device__ __forceinline__ void elim_step_pair(const ElimArgs& a, int k, int q, const double* __restrict__ Wk, const double* __restrict__ Pcur,
                                               double* __restrict__ Pnext, double* tile) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lk = lane >> 4;
  const int lde = a.ld + CB;
  const int o = (a.nb & 1) ? 0 : 1;
  int BI, c;
  elim_pair_index(q, o, BI, c);
  const int r0 = 2 * BI - o;
  if (c > a.nb - 1) return;  // (cannot happen inside elim_pair_grid's range)
  // the two blocks: (r0, c) if r0 >= max(c, 0), (r0 + 1, c) if r0 + 1 <= nb -- and not the next diagonal block, which is workgroup 0's
  bool live[2];
  int rows[2], ldts[2];
  double* Tbs[2];
  double accs[2][4][4];  // negated tiles
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    const int bi = r0 + ii;
    rows[ii] = bi;
    live[ii] = bi >= 0 && bi >= c && bi <= a.nb && !(bi == k + 1 && c == k + 1 && k + 1 < a.nb);
    if (!live[ii]) rows[ii] = c;  // (a placeholder inside the arrays)
    Tbs[ii] = elim_tile(a, rows[ii], c, ldts[ii]);
  }
  if (!live[0] && !live[1]) return;  // (the pair of the next diagonal block's column that holds nothing else)
  ESTAMP(blockIdx.x == gridDim.x / 2, k, 8)
  ESTAMP(blockIdx.x == gridDim.x - 1, k, 10)
  // every operand requested at once: W, the raw panel blocks of column c and of both rows, both state tiles
  stage_aside(tile, as_global(Wk), CB, tid);  // tile[kk][cc] = W(cc, kk)
  const bool own_c = (live[0] ? rows[0] : rows[1]) != c;  // no row of the pair is block row c: X_c takes a product of its own
  double bvc[16], bvr[2][16];
  if (own_c) load_bside(bvc, as_global(Pcur + CB * c), lde, w, lane);
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
    if (live[ii]) load_bside(bvr[ii], as_global(Pcur + CB * rows[ii]), lde, w, lane);
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    const bool fetch = live[ii] && !(rows[ii] == k || c == k);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t) accs[ii][mi][t] = fetch ? -Tbs[ii][(size_t)(16 * mi + 4 * t + lk) * ldts[ii] + 16 * w + (lane & 15)] : 0.0;
  }
  double xc[4][4], xr[2][4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) xc[mi][t] = xr[0][mi][t] = xr[1][mi][t] = 0.0;
  __syncthreads();
  ESTAMP(blockIdx.x == gridDim.x / 2, k, 12)
  if (own_c) mma_64<true>(tile, bvc, xc, lane);  // X_c = M_c W^T
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
    if (live[ii]) mma_64<true>(tile, bvr[ii], xr[ii], lane);  // X of block row rows[ii]
  if (!own_c) {  // block row c is the first live row of the pair
    const int ic = live[0] ? 0 : 1;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t) xc[mi][t] = ic ? xr[1][mi][t] : xr[0][mi][t];
  }
  // rows 0 / 1 of the solved right-hand sides: Yt, Ft of block k -- by the workgroup that holds block (nb, k)
  if (c == k && w == 0 && (lane & 15) < 2) {
    double* dst = (lane & 15) == 0 ? a.yt : a.ft;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
      if (live[ii] && rows[ii] == a.nb) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int t = 0; t < 4; ++t) dst[CB * k + 16 * mi + 4 * t + lk] = xr[ii][mi][t];
      }
  }
  __syncthreads();  // every wave is done with the W tile
  ESTAMP(blockIdx.x == gridDim.x / 2, k, 13)
  // A side of both updates: tile[kk][cc] = X_c(cc, kk)
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) tile[(16 * mi + 4 * t + lk) * CPITCH + 16 * w + (lane & 15)] = xc[mi][t];
  __syncthreads();
  ESTAMP(blockIdx.x == gridDim.x / 2, k, 14)
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    if (!live[ii]) continue;
    // B side: X_i(row 16 w + (lane & 15), kk = 4 ks + lk) is exactly xr[ks / 4][ks % 4] of this lane
    double bv[16];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t) bv[4 * mi + t] = xr[ii][mi][t];
    asm volatile("s_nop 7\n\ts_nop 7"
                 : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]), "+v"(bv[8]),
                   "+v"(bv[9]), "+v"(bv[10]), "+v"(bv[11]), "+v"(bv[12]), "+v"(bv[13]), "+v"(bv[14]), "+v"(bv[15]));
    mma_64(tile, bv, accs[ii], lane);  // -T_new = -T_old + X_i X_c^T
    ESTAMP(blockIdx.x == gridDim.x / 2 && ii == 1, k, 15)
    elim_store_plain(a, k, rows[ii], c, accs[ii], Tbs[ii], ldts[ii], Pnext);
  }
  ESTAMP(blockIdx.x == gridDim.x / 2, k, 9)
  ESTAMP(blockIdx.x == gridDim.x - 1, k, 11)
}
__device__ __forceinline__ void elim_step_pair_wg(const ElimArgs& a, int k, const double* __restrict__ Wk, const double* __restrict__ Pcur,
                                                  double* __restrict__ Pnext, double* __restrict__ Wn) {
  __shared__ __attribute__((aligned(16))) double lds[CB * CPITCH];
  __shared__ __attribute__((aligned(16))) double sb[ED_LDS];
  if (blockIdx.x == 0) {  // the chain: the next diagonal block through the ordinary block routine
    if (k + 1 < a.nb) elim_step_core(a, k, k + 1, k + 1, Wk, Pcur, Pnext, Wn, -1, nullptr, lds, sb);
    return;
  }
  if (blockIdx.x == 256) return;  // (the workgroup that would land on the diagonal block's CU: idle, as in elim_step_block)
  const int q = (int)blockIdx.x - 1 - (blockIdx.x > 256);
  if (q >= elim_pair_grid_dev(a.nb)) return;
  elim_step_pair(a, k, q, Wk, Pcur, Pnext, lds);
}
__global__ __launch_bounds__(256) void k_elim_stepS(const ElimArgs a, int k, const double* __restrict__ Wk, const double* __restrict__ Pcur,
                                                    double* __restrict__ Pnext, double* __restrict__ Wn) {
  elim_step_pair_wg(a, k, Wk, Pcur, Pnext, Wn);
}
__global__ __launch_bounds__(256) void k_elim_stepS_b(const BatchSlot* __restrict__ slots, int k) {
  const BatchSlot& sl = slots[blockIdx.y];
  const size_t lde = (size_t)sl.ea.ld + CB;
  double* P0 = sl.panels;
  double* P1 = sl.panels + lde * CB;
  elim_step_pair_wg(sl.ea, k, sl.Winv + (size_t)k * CB * CB, (k & 1) ? P1 : P0, (k & 1) ? P0 : P1, sl.Winv + (size_t)(k + 1) * CB * CB);
}
// one evaluation's fused steps on row pairs when a step has BOGP_ELIM_STEP_PAIR_MIN .. BOGP_ELIM_STEP_PAIR_MAX 64 x 64 blocks (defaults 110 .. 500:
// nb = 14 .. 30, N = 833 .. 1920; MIN = 0: never).  Measured per step, blocks / pairs (us; tools/probes/time_elim_pairmin.py, time_elim_ld.py): nb = 3 34.4 / 37.1,
// 8: 25.9 / 26.8, 10 - 12: equal, 14: 24.3 / 23.9, 16: 23.9 / 23.5, 20: 24.1 / 23.4, 22: 25.0 / 23.4, 23: 25.8 / 23.4, 24 - 30: 26.1 - 28.6 / 23.3 - 23.6.
// Above -- nb = 31, 32: 272 / 289 pair workgroups -- some CUs hold two of them, the diagonal block's among them, and a step costs 31 / 34 us where the block
// grid costs 29.7 / 30.3 (27.3 with the diagonal block's CU kept free) and the pair grid 23.6 at nb = 30 (profiles/r05_elim_chain.txt)
static bool elim_step_pairs(int grid) {
  // BOGP_ELIM_STEP_PAIRS = 0 / 1: never / at every size (tests/test_gpu_nll_batch_variants.py runs both against the batch kernels' bits)
  static const int forced = [] { const char* e = getenv("BOGP_ELIM_STEP_PAIRS"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
  if (forced >= 0) return forced == 1;
  constexpr int lo = 110, hi = 500;
  return grid >= lo && grid <= hi;
}

// ---- the step split in two launches (batches whose blocks outnumber the workgroup slots) --------------------------------------------
// k_elim_step lets every workgroup form the two solved panel blocks it needs itself (three 64^3 products a block and step: free while
// the machine has idle slots, 3 x the flops once P matrices fill it).  Here the nb + 1 solved blocks X_i = M_i W_k^T of a step are
// formed ONCE (k_elim_panel_b: the same mma_64 on the same operands, stored as plain 64 x 64 tiles) and the update reads them back
// in the layouts the fused kernel built in registers / LDS (k_elim_update_b: stage_aside / load_bside on the stored tiles): one
// product a block and step, bit-identical results.
// the solved panel block of block row bi at step k: X = M W_k^T from the raw panel, into the slot's solved panel k mod 4
__device__ __forceinline__ void elim_panel_row(const BatchSlot& sl, int k, int bi, double* lds) {
  const ElimArgs& a = sl.ea;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lk = lane >> 4;
  const size_t lde = (size_t)a.ld + CB;
  const double* __restrict__ Pcur = sl.panels + ((k & 1) ? lde * CB : 0);
  stage_aside(lds, as_global(sl.Winv + (size_t)k * CB * CB), CB, tid);  // tile[kk][c] = W(c, kk)
  double bv[16];
  load_bside(bv, as_global(Pcur + (size_t)CB * bi), (int)lde, w, lane);
  double x[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) x[mi][t] = 0.0;
  __syncthreads();
  mma_64<true>(lds, bv, x, lane);  // X_i = M_i W^T: rows 16 w .. of block row bi, element (row, col 16 mi + 4 t + lk)
  if (bi == a.nb && w == 0 && (lane & 15) < 2) {  // rows 0 / 1 of the solved right-hand sides: Yt, Ft of block k
    double* dst = (lane & 15) == 0 ? a.yt : a.ft;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t) dst[CB * k + 16 * mi + 4 * t + lk] = x[mi][t];
  }
  // (four solved panels are kept, by k mod 4: a grouped step needs X of up to four consecutive steps at once)
  double* __restrict__ Xs = sl.xpanel + (size_t)(k & 3) * ((size_t)a.nb + 1) * CB * CB + (size_t)bi * CB * CB;  // element (row, col) at Xs[row + 64 col]
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) Xs[(size_t)(16 * mi + 4 * t + lk) * CB + 16 * w + (lane & 15)] = x[mi][t];
}
__global__ __launch_bounds__(256) void k_elim_panel_b(const BatchSlot* __restrict__ slots, int k) {
  __shared__ __attribute__((aligned(16))) double lds[CB * CPITCH];
  elim_panel_row(slots[blockIdx.y], k, (int)blockIdx.x, lds);
}
// step k on ONE block from the stored solved panel: T <- (restart ? 0 : T) - X_i X_j^T, then elim_store_block
__device__ __forceinline__ void elim_update_one(const BatchSlot& sl, int k, int bi, int bj, double* lds, double* sb) {
  const ElimArgs& a = sl.ea;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lk = lane >> 4;
  const size_t lde = (size_t)a.ld + CB;
  const bool restart = bi == k || bj == k;
  if (bi == k + 1 && bj == k + 1) __builtin_amdgcn_s_setprio(3);  // (as in elim_step_core)
  const double* __restrict__ xp = sl.xpanel + (size_t)(k & 3) * ((size_t)a.nb + 1) * CB * CB;
  stage_aside(lds, as_global(xp + (size_t)bj * CB * CB), CB, tid);  // A side: tile[kk][c] = X_j(c, kk)
  double bv[16];
  load_bside(bv, as_global(xp + (size_t)bi * CB * CB), CB, w, lane);  // B side: X_i(row 16 w + (lane & 15), kk = 4 ks + lk)
  int ldt;
  double* __restrict__ Tb = elim_tile(a, bi, bj, ldt);
  double acc[4][4];  // negated tile
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mi][t] = restart ? 0.0 : -Tb[(size_t)(16 * mi + 4 * t + lk) * ldt + 16 * w + (lane & 15)];
  __syncthreads();
  mma_64(lds, bv, acc, lane);  // -T_new = -T_old + X_i X_j^T
  elim_store_block(a, k, bi, bj, acc, Tb, ldt, lds, sb, sl.panels + ((k & 1) ? 0 : lde * CB), sl.Winv + (size_t)(k + 1) * CB * CB);
}
// xcd != 0: a 1-D grid of G * P workgroups whose linear id L is dealt XCD-locally -- the hardware hands workgroup L to XCD L % 8, so XCD x
// is given the x-th eighth of the slot-major work list (unit u = slot * G + block): a slot's solved panel (1 MB at N = 2048) is then read by
// the workgroups of one or two XCDs only and stays in their L2 instead of being fetched by all eight.
// xcd < 0 (inside a GROUPED step, below): only the blocks of the -xcd columns / rows c = G, G + 1, ... -- for column / row c the nb + 1 blocks
// (c, t) for t <= c and (t, c) above -- i.e. the blocks whose state after step k the raw panels and diagonal factors of the group's later steps
// are made of.
__global__ __launch_bounds__(256) void k_elim_update_b(const BatchSlot* __restrict__ slots, int k, int xcd, int G, int P) {
  __shared__ __attribute__((aligned(16))) double lds[CB * CPITCH];
  __shared__ __attribute__((aligned(16))) double sb[ED_LDS];
  int slot = (int)blockIdx.y, blk = (int)blockIdx.x;
  if (xcd < 0) {  // sub mode: -xcd columns / rows starting at c0 = G, blockIdx.x = ci (nb + 1) + t
    const int nb1 = slots[slot].ea.nb + 1;
    const int ci = (int)blockIdx.x / nb1, t = (int)blockIdx.x % nb1;
    const int c0 = G, c = c0 + ci;
    if (t >= c0 && t < c) return;  // block (c, t) already belongs to the earlier column / row t of this launch
    const int sbi = t <= c ? c : t, sbj = t <= c ? t : c;
    blk = sbi * (sbi + 1) / 2 + sbj;
  } else if (xcd) {
    const long U = (long)G * P, L = (long)blockIdx.x;
    const long per = (U + 7) / 8;           // units per XCD (the last ones may run short)
    const long u = (L % 8) * per + L / 8;   // L / 8 < per by the grid size 8 * per
    if (u >= U) return;
    slot = (int)(u / G);
    blk = (int)(u % G);
  }
  int bi, bj;
  tri_index(blk, bi, bj);
  elim_update_one(slots[slot], k, bi, bj, lds, sb);
}

// The group chain's step in ONE launch: the sub-mode update with the solved panel formed by the workgroups themselves (elim_step_core: X_i, X_j
// from the raw panel and W_k, the fused kernel's three products a block -- free here, the launch is a handful of blocks per slot waiting for its
// diagonal block's factorisation).  The workgroups of the first column (ci = 0: one per block row t) leave X^k_t in the solved panel for the
// whole-state update that follows the group.  Saves the k_elim_panel_b launch of every step but the group's last (13 us each).
__global__ __launch_bounds__(256) void k_elim_substep_b(const BatchSlot* __restrict__ slots, int k, int ncol) {
  const BatchSlot& sl = slots[blockIdx.y];
  const ElimArgs& a = sl.ea;
  const int nb1 = a.nb + 1;
  const int ci = (int)blockIdx.x / nb1, t = (int)blockIdx.x % nb1;
  const int c0 = k + 1, c = c0 + ci;
  if (t >= c0 && t < c) return;  // block (c, t) already belongs to the earlier column / row t of this launch
  const size_t lde = (size_t)a.ld + CB;
  double* P0 = sl.panels;
  double* P1 = sl.panels + lde * CB;
  double* Xout = sl.xpanel + (size_t)(k & 3) * ((size_t)a.nb + 1) * CB * CB + (size_t)t * CB * CB;
  __shared__ __attribute__((aligned(16))) double lds[CB * CPITCH];
  __shared__ __attribute__((aligned(16))) double sb[ED_LDS];
  elim_step_core(a, k, t <= c ? c : t, t <= c ? t : c, sl.Winv + (size_t)k * CB * CB, (k & 1) ? P1 : P0, (k & 1) ? P0 : P1,
                 sl.Winv + (size_t)(k + 1) * CB * CB, ci == 0 ? t : -1, Xout, lds, sb);
}

// ---- GROUPED steps: two or four block columns per pass over the state (r04) ----------------------------------------------------------
// A batch that fills the GPU is bound by the read-modify-write of the N x N state, once per 64-column step (~1.1 GB of traffic per
// evaluation at N = 2048).  Steps k and k + 1 are therefore applied in ONE pass:
//   k_elim_panel_b(k)                X^k  (as for a split step)
//   k_elim_update_b(k, sub mode)     step k on the nb + 1 blocks of column / row k + 1 only: they make the raw panel of step k + 1 and its
//                                    diagonal factor W_{k+1} -- the ordinary block routine, nothing new
//   k_elim_panel_b(k + 1)            X^{k+1}
//   k_elim_updateG_b(k, 2)           every block once: T <- step k (unless done above) then step k + 1, the intermediate state kept in the
//                                    accumulators instead of a store + reload -- the same mma_64 calls on the same values in the same order,
//                                    (-(-x) = x exactly), so the bits of the two separate steps; then what step k + 1 publishes (raw panel and
//                                    diagonal factor of k + 2).
// With ng = 4 the same, one level deeper: after panel(k + g) the sub-mode launch applies step k + g to the blocks of the columns / rows
// k + g + 1 .. k + 3 (each block once: a block lying in two of them is taken by the earlier one), and k_elim_updateG_b(k, 4) applies to every
// block the steps from its last restart inside the group (or from k) to k + 3.
__device__ __forceinline__ void elim_group_block(const BatchSlot& sl, int k, int ng, int bi, int bj, double* lds, double* sb) {
  const ElimArgs& a = sl.ea;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lk = lane >> 4;
  const size_t lde = (size_t)a.ld + CB;
  const size_t xsz = ((size_t)a.nb + 1) * CB * CB;
  const int klast = k + ng - 1;
  if (bi == klast + 1 && bj == klast + 1) __builtin_amdgcn_s_setprio(3);  // (as in elim_step_core)
  // the LAST step of the group that restarts this block (its row or column index); the sub-mode launches have carried the blocks of
  // columns / rows k + 1 .. klast up to their restart, which zeroes them anyway: what such a block still needs are the steps from there on
  int first = -1;
  if (bi >= k && bi <= klast) first = bi;
  if (bj >= k && bj <= klast) first = max(first, bj);
  const bool restart = first >= 0;
  if (!restart) first = k;
  int ldt;
  double* __restrict__ Tb = elim_tile(a, bi, bj, ldt);
  double bv[16];
  double acc[4][4];  // negated tile
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mi][t] = restart ? 0.0 : -Tb[(size_t)(16 * mi + 4 * t + lk) * ldt + 16 * w + (lane & 15)];
  for (int sidx = first; sidx <= klast; ++sidx) {
    const double* __restrict__ xs = sl.xpanel + (size_t)(sidx & 3) * xsz;
    if (sidx > first) __syncthreads();  // every wave is done with the previous step's X_j tile
    stage_aside(lds, as_global(xs + (size_t)bj * CB * CB), CB, tid);
    load_bside(bv, as_global(xs + (size_t)bi * CB * CB), CB, w, lane);
    __syncthreads();
    mma_64(lds, bv, acc, lane);  // step sidx:  -T <- -T + X_i X_j^T, the intermediate state never leaves the accumulators
  }
  elim_store_block(a, klast, bi, bj, acc, Tb, ldt, lds, sb, sl.panels + ((klast & 1) ? 0 : lde * CB), sl.Winv + (size_t)(klast + 1) * CB * CB);
}
__global__ __launch_bounds__(256) void k_elim_updateG_b(const BatchSlot* __restrict__ slots, int k, int ng, int xcd, int G, int P) {
  __shared__ __attribute__((aligned(16))) double lds[CB * CPITCH];
  __shared__ __attribute__((aligned(16))) double sb[ED_LDS];
  int slot = (int)blockIdx.y, blk = (int)blockIdx.x;
  if (xcd) {
    const long U = (long)G * P, L = (long)blockIdx.x;
    const long per = (U + 7) / 8;
    const long u = (L % 8) * per + L / 8;
    if (u >= U) return;
    slot = (int)(u / G);
    blk = (int)(u % G);
  }
  int bi, bj;
  tri_index(blk, bi, bj);
  elim_group_block(slots[slot], k, ng, bi, bj, lds, sb);
}

// ---- the grouped update on 128 x 128 SUPER-TILES (r04) --------------------------------------------------------------------------------
// k_elim_updateG_b fetches two 32-KB solved tiles per 64^3 product: with P matrices in flight the solved panels of a slot (4 MB at
// N = 2048) do not stay in an XCD's 4-MB L2 next to the streaming state, and the kernel sits at ~47 % of the matrix peak on those
// fetches (profiles/r04_nll_batch_group.txt).  Here a workgroup owns the 2 x 2 blocks (2 BI + a, 2 BJ + b): per step it fetches the TWO
// A-side tiles X_{2BJ}, X_{2BJ+1} and the TWO B-side fragments X_{2BI}, X_{2BI+1} for FOUR products -- half the fetches per
// flop.  Every block still sees the accumulations of k_elim_updateG_b on the same operands in the same order (its own first step ..
// klast), then elim_store_block's stores: the same bits.  Blocks above
// the diagonal or outside the nb + 1 block rows / nb block columns are computed and discarded, a block that restarts inside the group is
// zeroed when the loop reaches its step (no per-block control flow around the MFMA chains: see the kernel).  The next diagonal block is
// factored by k_elim_diag_b.
// (buffer accesses: a uniform descriptor, ONE 32-bit lane offset and a scalar offset per access.  The slot's pointers come out of memory,
// i.e. generic, and a generic / global access costs a 64-bit address pair per load that the compiler keeps live across the step loop --
// ~100 VGPRs in this kernel, which has none to spare)
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const double* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), 0, 0x7fffffff, 0x00020000);
}
#ifndef BOGP_STATE_AUX
#define BOGP_STATE_AUX 2
#endif
// AUX = 2: non-temporal (the state tiles: touched once per pass, they should not push the solved panels out of the L2)
template <int AUX = 0>
__device__ __forceinline__ double buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff_doubles, unsigned soff_doubles) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, 8 * voff_doubles, 8 * soff_doubles, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ void buf_store(double v, __amdgpu_buffer_rsrc_t r, unsigned voff_doubles, unsigned soff_doubles) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, 8 * voff_doubles, 8 * soff_doubles, AUX);
}
// r06, last session: NO LDS and NO barrier (what k_contract16d did for the sweep and mm128_tile_direct for the tile products).  The A side -- the two
// solved tiles X_{2BJ}, X_{2BJ+1}, k-major with the row index contiguous -- comes straight from global memory as row PAIRS per lane (lane (k, i) takes
// rows 2 i, 2 i + 1 of a 32-row half with one 128-bit buffer load: fragment mi = 2 half + e stands for rows 32 half + 2 i + e of the tile, i.e. for the
// COLUMNS 32 (mi >> 1) + 8 t + 2 lk + (mi & 1) of the output block in register t of lane group lk), the B side one value a k-step as before; a k-pair's
// twelve loads go out as ONE block in front of the 32 MFMAs of the k-pair before it (two register slots), across step boundaries too.  A block that
// restarts at a step is zeroed between that step's first load block and its first MFMA, behind a drain.  Per output element the same accumulations in
// the same order as k_elim_updateG_b and as the staged version of r04-r06 (two LDS tiles + two barriers a step; git de191e8): the same bits
// (tests/test_gpu_nll_batch_variants.py).  A slot of a batch of 16 at N = 2048: 334 -> 321 us, of 8 at N = 3072: 947 -> 899 us (profiles/r06_elim_super_direct_ab.txt).
__global__ __launch_bounds__(256, 2) void k_elim_updateS_b(const BatchSlot* __restrict__ slots, int k, int ng, int xcd, int G, int P) {
  int slot = (int)blockIdx.y, blk = (int)blockIdx.x;
  if (xcd) {
    const long U = (long)G * P, L = (long)blockIdx.x;
    const long per = (U + 7) / 8;
    const long u = (L % 8) * per + L / 8;
    if (u >= U) return;
    slot = (int)(u / G);
    blk = (int)(u % G);
  }
  slot = __builtin_amdgcn_readfirstlane(slot);
  const BatchSlot& sl = slots[slot];
  const ElimArgs& a = sl.ea;
  int BI, BJ;
  tri_index(blk, BI, BJ);
  BI = __builtin_amdgcn_readfirstlane(BI);
  BJ = __builtin_amdgcn_readfirstlane(BJ);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lk = lane >> 4, li = lane & 15;
  const unsigned lde = (unsigned)a.ld + CB;
  const size_t xsz = ((size_t)a.nb + 1) * CB * CB;
  const int klast = k + ng - 1;
  bool valid[2][2];
  int first[2][2];
  bool any = false;
#pragma unroll
  for (int ia = 0; ia < 2; ++ia)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      const int bi = 2 * BI + ia, bj = 2 * BJ + ib;
      valid[ia][ib] = bi <= a.nb && bj < a.nb && bj <= bi;
      int f = -1;
      if (bi >= k && bi <= klast) f = bi;
      if (bj >= k && bj <= klast) f = max(f, bj);
      first[ia][ib] = f;  // -1: no restart inside the group, the block comes from the state
      any = any || valid[ia][ib];
    }
  if (!any) return;  // nothing of this super-tile is there
  // output block element (row 16 w + li, column colmap(mi, t) + 2 lk): accumulator [mi] component t
#define BOGP_ES_COL(mi, t) (32 * ((mi) >> 1) + 8 * (t) + ((mi) & 1))
  d4 acc[2][2][4];
#pragma unroll
  for (int ia = 0; ia < 2; ++ia)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      if (valid[ia][ib] && first[ia][ib] < 0) {
        int ldt;
        const __amdgpu_buffer_rsrc_t Tb = tile_rsrc(elim_tile(a, 2 * BI + ia, 2 * BJ + ib, ldt));
        const unsigned voff = (unsigned)(2 * lk) * ldt + 16 * w + li;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[ia][ib][mi][t] = -buf_load<BOGP_STATE_AUX>(Tb, voff, (unsigned)BOGP_ES_COL(mi, t) * ldt);
      } else {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ia][ib][mi] = (d4){0.0, 0.0, 0.0, 0.0};
      }
    }
  const int tj1 = 2 * BJ + (2 * BJ + 1 <= a.nb ? 1 : 0), ti1 = 2 * BI + (2 * BI + 1 <= a.nb ? 1 : 0);  // (past the edge: the tile before, discarded)
  const unsigned voffA = 8u * ((unsigned)lk * CB + 2 * li);        // bytes: row pair 2 li of half 0 at k = lk of the k-step; half 1: + 256 bytes
  const unsigned voffB = (unsigned)lk * CB + 16 * w + li;          // doubles: row 16 w + li at k = lk of the k-step
  v4u av[2][2][2][2];  // [slot][tile ib][k-step][half]: two doubles (rows 2 i, 2 i + 1)
  double bv[2][2][2];  // [slot][tile ia][k-step]
  // the twelve loads of k-pair kp_ (0 .. 7) of step st_: tiles of the solved panel of that step
#define BOGP_ES_LOADS(sl_, st_, kp_)                                                                                   \
  do {                                                                                                                 \
    const double* xs_ = sl.xpanel + (size_t)((st_) & 3) * xsz;                                                          \
    const __amdgpu_buffer_rsrc_t a0_ = tile_rsrc(xs_ + (size_t)(2 * BJ) * CB * CB), a1_ = tile_rsrc(xs_ + (size_t)tj1 * CB * CB); \
    const __amdgpu_buffer_rsrc_t b0_ = tile_rsrc(xs_ + (size_t)(2 * BI) * CB * CB), b1_ = tile_rsrc(xs_ + (size_t)ti1 * CB * CB); \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                                 \
      const unsigned so_ = (unsigned)(8 * (kp_) + 4 * h_) * CB;  /* doubles */                                          \
      av[sl_][0][h_][0] = __builtin_amdgcn_raw_buffer_load_b128(a0_, voffA, 8 * so_, 0);                               \
      av[sl_][0][h_][1] = __builtin_amdgcn_raw_buffer_load_b128(a0_, voffA + 256, 8 * so_, 0);                         \
      av[sl_][1][h_][0] = __builtin_amdgcn_raw_buffer_load_b128(a1_, voffA, 8 * so_, 0);                               \
      av[sl_][1][h_][1] = __builtin_amdgcn_raw_buffer_load_b128(a1_, voffA + 256, 8 * so_, 0);                         \
      bv[sl_][0][h_] = buf_load(b0_, voffB, so_);                                                                      \
      bv[sl_][1][h_] = buf_load(b1_, voffB, so_);                                                                      \
    }                                                                                                                  \
  } while (0)
#define BOGP_ES_AVAL(sl_, ib_, h_, mi_) __builtin_bit_cast(double, (v2u){av[sl_][ib_][h_][(mi_) >> 1][2 * ((mi_) & 1)], av[sl_][ib_][h_][(mi_) >> 1][2 * ((mi_) & 1) + 1]})
#define BOGP_ES_MFMAS(sl_)                                                                                             \
  do {                                                                                                                 \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                                   \
      _Pragma("unroll") for (int ib_ = 0; ib_ < 2; ++ib_)                                                              \
        _Pragma("unroll") for (int mi_ = 0; mi_ < 4; ++mi_) {                                                          \
          const double a_ = BOGP_ES_AVAL(sl_, ib_, h_, mi_);                                                           \
          mfma16(a_, bv[sl_][0][h_], acc[0][ib_][mi_]);                                                                \
          mfma16(a_, bv[sl_][1][h_], acc[1][ib_][mi_]);                                                                \
        }                                                                                                              \
  } while (0)
#define BOGP_ES_ALL_ACC                                                                                                                    \
  "+v"(acc[0][0][0]), "+v"(acc[0][0][1]), "+v"(acc[0][0][2]), "+v"(acc[0][0][3]), "+v"(acc[0][1][0]), "+v"(acc[0][1][1]), "+v"(acc[0][1][2]), \
      "+v"(acc[0][1][3]), "+v"(acc[1][0][0]), "+v"(acc[1][0][1]), "+v"(acc[1][0][2]), "+v"(acc[1][0][3]), "+v"(acc[1][1][0]), "+v"(acc[1][1][1]), \
      "+v"(acc[1][1][2]), "+v"(acc[1][1][3])
  BOGP_ES_LOADS(0, k, 0);
  // the VALU moves that built the accumulators must retire before the first MFMA reads them as SrcC (inline-asm MFMAs: wait states by hand)
  asm volatile("s_nop 7\n\ts_nop 7" : BOGP_ES_ALL_ACC);
  for (int sidx = k; sidx <= klast; ++sidx) {
    bool restart = false;
#pragma unroll
    for (int ia = 0; ia < 2; ++ia)
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) restart = restart || sidx == first[ia][ib];
    if (restart) {  // (uniform) T <- 0 - X_i X_j^T from this step on: behind the drain of what the block gathered so far, in front of the step's first MFMA
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : BOGP_ES_ALL_ACC);
#pragma unroll
      for (int ia = 0; ia < 2; ++ia)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
          if (sidx == first[ia][ib]) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[ia][ib][mi] = (d4){0.0, 0.0, 0.0, 0.0};
          }
      asm volatile("s_nop 7\n\ts_nop 7" : BOGP_ES_ALL_ACC);
    }
    const int snext = min(sidx + 1, klast);
#pragma unroll
    for (int kp = 0; kp < 8; ++kp) {
      if (kp < 7) BOGP_ES_LOADS((kp + 1) & 1, sidx, kp + 1);
      else BOGP_ES_LOADS(0, snext, 0);  // (the last step asks for its own first k-pair again: nothing uses it)
      __builtin_amdgcn_sched_barrier(0);
      BOGP_ES_MFMAS(kp & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : BOGP_ES_ALL_ACC);
#undef BOGP_ES_LOADS
#undef BOGP_ES_MFMAS
#undef BOGP_ES_AVAL
  // elim_store_block's stores: the block back into the state; column kn as it is / row kn transposed into the next raw panel (the next
  // diagonal block is factored by k_elim_diag_b)
  const __amdgpu_buffer_rsrc_t Pnext = tile_rsrc(sl.panels + ((klast & 1) ? 0 : (size_t)lde * CB));
  const int kn = klast + 1;
#pragma unroll
  for (int ia = 0; ia < 2; ++ia)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      if (!valid[ia][ib]) continue;
      const int bi = 2 * BI + ia, bj = 2 * BJ + ib;
      int ldt;
      const __amdgpu_buffer_rsrc_t Tb = tile_rsrc(elim_tile(a, bi, bj, ldt));
      const unsigned voff = (unsigned)(2 * lk) * ldt + 16 * w + li;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int t = 0; t < 4; ++t) buf_store<BOGP_STATE_AUX>(-acc[ia][ib][mi][t], Tb, voff, (unsigned)BOGP_ES_COL(mi, t) * ldt);
      if (kn >= a.nb) continue;
      if (bj == kn && bi > kn) {
        const unsigned vo = (unsigned)(2 * lk) * lde + 16 * w + li;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int t = 0; t < 4; ++t) buf_store(-acc[ia][ib][mi][t], Pnext, vo, (unsigned)BOGP_ES_COL(mi, t) * lde + (unsigned)(CB * bi));
      } else if (bi == kn && bj < kn) {
        const unsigned vo = (unsigned)(16 * w + li) * lde + 2 * lk;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int t = 0; t < 4; ++t) buf_store(-acc[ia][ib][mi][t], Pnext, vo, (unsigned)(CB * bj + BOGP_ES_COL(mi, t)));
      }
    }
#undef BOGP_ES_ALL_ACC
#undef BOGP_ES_COL
}

// the next diagonal block after a super-tile update: factored and inverted from the state by a workgroup of its own launch (inlined
// into k_elim_updateS_b, the factorisation takes that kernel's allocation past 256 VGPRs: one workgroup a CU instead of two).  The tile
// read back is the value elim_store_block stages from the accumulators, so W, log-determinant part and info are the same bits.
__global__ __launch_bounds__(256) void k_elim_diag_b(const BatchSlot* __restrict__ slots, int kn) {
  __shared__ __attribute__((aligned(16))) double cs[CB * (CB + 1)];
  __shared__ __attribute__((aligned(16))) double sb[ED_LDS];
  const BatchSlot& sl = slots[blockIdx.x];
  const ElimArgs& a = sl.ea;
  const int tid = threadIdx.x;
  const size_t lde = (size_t)a.ld + CB;
  int ldt;
  const double* __restrict__ T = elim_tile(a, kn, kn, ldt);
  for (int e = tid; e < CB * CB; e += 256) {
    const int r = e & 63, c = e >> 6;
    cs[r * (CB + 1) + c] = T[(size_t)c * ldt + r];
  }
  __syncthreads();
  elim_diag2<true>(cs, sb, sl.Winv + (size_t)kn * CB * CB, a.logpart + kn, a.info, CB * kn, 0, max(0, min(CB, a.N - CB * kn)),
             sl.panels + ((kn & 1) ? lde * CB : 0), (int)lde, kn, tid);
}

// R^-1 = -T into Rinv (lower triangle, column-major, ldr) and, by the last workgroup, the likelihood's scalars (k_fit_rho's
// expressions), the gradient's two weights and gamma = R^-1 y - beta R^-1 1
__device__ __forceinline__ void elim_finish_block(const ElimArgs& a, double* __restrict__ Rinv, int ldr, double* __restrict__ gamma,
                                                  double* __restrict__ scal, double* __restrict__ coefw, int estimate_trend, int mode,
                                                  double beta, double s2t_host) {
  const int tid = threadIdx.x;
  const int ntiles = a.nb * (a.nb + 1) / 2;
  if ((int)blockIdx.x < ntiles) {
    int bi, bj;
    tri_index((int)blockIdx.x, bi, bj);
    for (int e = tid; e < CB * CB; e += 256) {
      const int r = e & 63, c = e >> 6;
      const int row = CB * bi + r, col = CB * bj + c;
      if (row < a.N && col <= row) Rinv[(size_t)col * ldr + row] = -a.E[(size_t)col * a.ld + row];
    }
    return;
  }
  __shared__ double red[256];
  auto block_sum = [&](double v) {
    red[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
  };
  const int N = a.N;
  double sff = 0.0, sfy = 0.0;
  for (int i = tid; i < N; i += 256) {
    const double f = a.ft[i];
    sff = __builtin_fma(f, f, sff);
    sfy = __builtin_fma(f, a.yt[i], sfy);
  }
  sff = block_sum(sff);
  sfy = block_sum(sfy);
  const double nrm = sqrt(sff);
  double coef;
  if (estimate_trend) {
    const double G = -nrm, qty = sfy / G;
    coef = -(qty / G);
  } else {
    coef = -beta;
  }
  double srr = 0.0;
  for (int i = tid; i < N; i += 256) {
    const double r = __builtin_fma(coef, a.ft[i], a.yt[i]);
    srr = __builtin_fma(r, r, srr);
  }
  srr = block_sum(srr);
  for (int i = tid; i < N; i += 256) {  // block row nb, rows 0 / 1: -(R^-1 y)_i, -(R^-1 1)_i
    const double gy = a.Eb[(size_t)i * CB + 0], g1 = a.Eb[(size_t)i * CB + 1];
    gamma[i] = -__builtin_fma(coef, g1, gy);
  }
  if (tid == 0) {
    double ld_ = 0.0;
    for (int b = 0; b < a.nb; ++b) ld_ += a.logpart[b];
    scal[0] = ld_;
    scal[1] = nrm;
    scal[2] = sfy;
    scal[3] = srr;
    double iw = 0.0;
    int info = *a.info;
    memcpy(&iw, &info, sizeof(info));
    scal[62] = iw;
    const double s2t = mode == BOGP_MODE_NOISY ? s2t_host : (mode == BOGP_MODE_NOISELESS ? srr / (N - (estimate_trend ? 1 : 0)) : srr / N);
    coefw[0] = 1.0 / s2t;
    coefw[8] = 1.0 / s2t;
  }
}
__global__ __launch_bounds__(256) void k_elim_finish(const ElimArgs a, double* __restrict__ Rinv, int ldr, double* __restrict__ gamma,
                                                     double* __restrict__ scal, double* __restrict__ coefw, int estimate_trend, int mode,
                                                     double beta, double s2t_host) {
  elim_finish_block(a, Rinv, ldr, gamma, scal, coefw, estimate_trend, mode, beta, s2t_host);
}
__global__ __launch_bounds__(256) void k_elim_finish_b(const BatchSlot* __restrict__ slots, int estimate_trend, int mode, double beta) {
  const BatchSlot& sl = slots[blockIdx.y];
  elim_finish_block(sl.ea, sl.Rinv, sl.ea.ld, sl.gamma, sl.scal, sl.scal + 4 * BOGP_MAX_TARGETS, estimate_trend, mode, beta, sl.par[3]);
}

// [y; 1] into block row nb and identity padding of E outside its leading N x N block (k_build_R wrote the lower 64-tiles)
__device__ __forceinline__ void elim_init_column(const ElimArgs& a, const double* __restrict__ y) {
  const int j = blockIdx.x;  // column
  const int N = a.N, ld = a.ld;
  for (int r = threadIdx.x; r < CB; r += blockDim.x) a.Eb[(size_t)j * CB + r] = (j < N && r == 0) ? y[j] : ((j < N && r == 1) ? 1.0 : 0.0);
  if (j >= N) {
    for (int i = threadIdx.x; i < ld; i += blockDim.x) a.E[(size_t)j * ld + i] = i == j ? 1.0 : 0.0;
  } else {
    for (int i = N + threadIdx.x; i < ld; i += blockDim.x) a.E[(size_t)j * ld + i] = 0.0;
  }
}
__global__ void k_elim_init(const ElimArgs a, const double* __restrict__ y) { elim_init_column(a, y); }
__global__ void k_elim_init_b(const BatchSlot* __restrict__ slots, const double* __restrict__ y) { elim_init_column(slots[blockIdx.y].ea, y); }

// the elimination of P matrices at once (bogp_nll_batch): the launches of launch_elim with a second grid dimension over the slots
hipError_t launch_elim_batch(const BatchSlot* slots, int P, int ld, const double* y, int estimate_trend, int mode, double beta, hipStream_t st) {
  const int nb = ld / CB;
  hipLaunchKernelGGL(k_elim_init_b, dim3(ld, P), 64, 0, st, slots, y);
  hipLaunchKernelGGL(k_elim_first_b, dim3(nb + 1, P), 256, 0, st, slots);
  const int grid = (nb + 1) * (nb + 2) / 2 - 1;
  // fused steps while the batch leaves workgroup slots idle, split steps (a third of the matrix-core work, one more launch a step)
  // once it does not: same bits either way.  BOGP_ELIM_SPLIT_BLOCKS: blocks per step from which the split is taken
  static const long split_from = [] { const char* e = getenv("BOGP_ELIM_SPLIT_BLOCKS"); return e ? atol(e) : 600L; }();
  const bool split = (long)grid * P >= split_from;
  static const bool xcd_local = [] { const char* e = getenv("BOGP_ELIM_XCD"); return !(e && atoi(e) == 0); }();
  static const int group = [] {  // block columns per pass over the state: 4 (default), 2 (pair steps), 1
    const char* e = getenv("BOGP_ELIM_GROUP");
    const int g = e ? atoi(e) : 4;
    return g >= 4 ? 4 : (g >= 2 ? 2 : 1);
  }();
  // the whole-state update of a grouped step on 128 x 128 super-tiles (k_elim_updateS_b)
  // -- from BOGP_ELIM_SUPER super-tile workgroups a launch (default 1500: N = 2048 from P = 10; below that the 64 x 64 kernel's finer grain
  // fills the GPU better; 0 = never)
  static const long super_from = [] { const char* e = getenv("BOGP_ELIM_SUPER"); return e ? atol(e) : 1500L; }();
  const int SR = (nb + 2) / 2, sgrid = SR * (SR + 1) / 2;
  const bool super_tiles = super_from > 0 && (long)sgrid * P >= super_from;
  const unsigned gs = xcd_local ? (unsigned)(8 * (((long)sgrid * P + 7) / 8)) : (unsigned)sgrid;
  // the panel chain of the group k .. k + ng - 1: X^(k+g), then step k + g on the blocks of the columns / rows k + g + 1 .. k + ng - 1
  // the chain's steps as ONE launch each (k_elim_substep_b: the panel formed by the update's own workgroups) while the first of them is at most
  // BOGP_ELIM_SUBSTEP blocks (default 600; 0 = never): -4 % a batch at P = 2 .. 4, +2 % at P = 16 (N = 2048), where the tripled products cost
  // more than the saved launch
  static const long substep_upto = [] { const char* e = getenv("BOGP_ELIM_SUBSTEP"); return e ? atol(e) : 600L; }();
  auto chain = [&](int k, int ng, hipStream_t s) {
    if ((long)(ng - 1) * (nb + 1) * P <= substep_upto && ng > 1) {
      for (int g = 0; g + 1 < ng; ++g)
        hipLaunchKernelGGL(k_elim_substep_b, dim3((unsigned)((ng - 1 - g) * (nb + 1)), P), 256, 0, s, slots, k + g, ng - 1 - g);
      hipLaunchKernelGGL(k_elim_panel_b, dim3(nb + 1, P), 256, 0, s, slots, k + ng - 1);
      return;
    }
    for (int g = 0; g < ng; ++g) {
      hipLaunchKernelGGL(k_elim_panel_b, dim3(nb + 1, P), 256, 0, s, slots, k + g);
      if (g + 1 < ng) hipLaunchKernelGGL(k_elim_update_b, dim3((unsigned)((ng - 1 - g) * (nb + 1)), P), 256, 0, s, slots, k + g, -(ng - 1 - g), k + g + 1, P);
    }
  };
  for (int k = 0; k < nb; ++k) {
    const int ng = !split ? 1 : (group >= 4 && k + 3 < nb ? 4 : (group >= 2 && k + 1 < nb ? 2 : 1));
    if (ng > 1) {
      const unsigned g1 = xcd_local ? (unsigned)(8 * (((long)grid * P + 7) / 8)) : (unsigned)grid;
      chain(k, ng, st);
      if (super_tiles) {
        hipLaunchKernelGGL(k_elim_updateS_b, dim3(gs, xcd_local ? 1 : P), 256, 0, st, slots, k, ng, xcd_local ? 1 : 0, sgrid, P);
        if (k + ng < nb) hipLaunchKernelGGL(k_elim_diag_b, dim3(P), 256, 0, st, slots, k + ng);
      } else {
        hipLaunchKernelGGL(k_elim_updateG_b, dim3(g1, xcd_local ? 1 : P), 256, 0, st, slots, k, ng, xcd_local ? 1 : 0, grid, P);
      }
      k += ng - 1;
      continue;
    }
    if (split) {
      hipLaunchKernelGGL(k_elim_panel_b, dim3(nb + 1, P), 256, 0, st, slots, k);
      if (xcd_local) hipLaunchKernelGGL(k_elim_update_b, dim3((unsigned)(8 * (((long)grid * P + 7) / 8))), 256, 0, st, slots, k, 1, grid, P);
      else hipLaunchKernelGGL(k_elim_update_b, dim3(grid, P), 256, 0, st, slots, k, 0, grid, P);
    } else {
      if (P == 1 && elim_step_pairs(grid))
        hipLaunchKernelGGL(k_elim_stepS_b, dim3(elim_pair_grid(nb) + 2, P), 256, 0, st, slots, k);
      else
        hipLaunchKernelGGL(k_elim_step_b, dim3(P == 1 && grid > 512 ? grid + 2 : grid, P), 256, 0, st, slots, k);  // (+ 2: launch_elim's free-CU layout, elim_step_block)
    }
  }
  hipLaunchKernelGGL(k_elim_finish_b, dim3(nb * (nb + 1) / 2 + 1, P), 256, 0, st, slots, estimate_trend, mode, beta);
  return hipGetLastError();
}

hipError_t launch_elim(const ElimArgs& a, const double* y, double* Winv, double* panels, double* Rinv, int ldr, double* gamma, double* scal,
                       double* coefw, int estimate_trend, int mode, double beta, double s2t_host, hipStream_t st) {
  const int nb = a.nb, lde = a.ld + CB;
  double* P[2] = {panels, panels + (size_t)lde * CB};
  hipLaunchKernelGGL(k_elim_init, dim3(a.ld), 64, 0, st, a, y);
  hipLaunchKernelGGL(k_elim_first, dim3(nb + 1), 256, 0, st, a, Winv, P[0]);
  const int grid = (nb + 1) * (nb + 2) / 2 - 1;
  const bool super_step = elim_step_pairs(grid);
  for (int k = 0; k < nb; ++k) {
    if (super_step)
      hipLaunchKernelGGL(k_elim_stepS, dim3(elim_pair_grid(nb) + 2), 256, 0, st, a, k, Winv + (size_t)k * CB * CB, P[k & 1], P[(k + 1) & 1],
                         Winv + (size_t)(k + 1) * CB * CB);
    else
      hipLaunchKernelGGL(k_elim_step, dim3(grid > 512 ? grid + 2 : grid), 256, 0, st, a, k, Winv + (size_t)k * CB * CB, P[k & 1], P[(k + 1) & 1],
                         Winv + (size_t)(k + 1) * CB * CB);  // (+ 2: the diagonal block's CU kept free, elim_step_block)
  }
  hipLaunchKernelGGL(k_elim_finish, dim3(nb * (nb + 1) / 2 + 1), 256, 0, st, a, Rinv, ldr, gamma, scal, coefw, estimate_trend, mode, beta,
                     s2t_host);
  return hipGetLastError();
}

#ifdef ELIM_PROFILE
hipError_t debug_elim_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_elim_stamps), sizeof(g_elim_stamps)); }
#endif
}  // namespace bogp
