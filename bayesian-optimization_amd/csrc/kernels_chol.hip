// kernels_chol.hip -- L = chol(R) for the likelihood path of libbogp (gfx950), replacing rocsolver_dpotrf.
//
// Reference call: scipy.linalg.cholesky(R, lower=True) in GaussianProcess._compute_aux_var (gpr.py:795); it runs once
// per likelihood evaluation, i.e. up to 100*d times per fit (gpr.py:1058-1197), so at N ~ 2048 its LATENCY, not its
// N^3/3 flops, decides the fit time: rocSOLVER spends 4.4 of its 5 ms in 17 single-workgroup potf2 / forward-
// substitution launches (profiles/r01_nll_rocsolver_kernel_stats.csv).
//
// Layout: A is column-major with ld = 64*nb (the engine pads R with an identity block, so no kernel has an edge
// case); only the lower triangle (and the diagonal blocks) is read or written.  Right-looking, 64-wide block columns,
// two launches per block column:
//   k_chol_panel(k)   X = A[k+1:, k] W_k^T with W_k = L_kk^-1 (explicit 64x64 inverse): a 64x64x64 product per 64 rows
//                     on v_mfma_f64_16x16x4_f64 -- no serial substitution in the panel
//   k_chol_update(k)  workgroup 0:  A_{k+1,k+1} -= X_{k+1} X_{k+1}^T (MFMA), then factors that block AND inverts the
//                                   factor, blocked by 4 columns (256 threads, 4x4 elements each in registers, strips
//                                   exchanged through 4 KB of LDS, two barriers per 4 columns); the serial chain of
//                                   the factorisation runs BESIDE the trailing update
//                     others:       A_ij -= X_i X_j^T (64x64 tiles, K = 64), same MFMA micro-kernel
// Measured alternatives for the diagonal block (tools/probes/ubench_potf2.hip, profiles/r01_ubench_potf2.txt): one wave with
// a row per lane and v_readlane / ds_bpermute / LDS broadcasts needs 50-300 us per block (SGPR pressure and spills).
#include "bogp_device.h"
#include "bogp_internal.h"

#include "bogp_chol_device.h"

namespace bogp {

// ---------------------------------------------------------------------------------------------------------------
// First diagonal block (nothing to subtract yet).
// ---------------------------------------------------------------------------------------------------------------
// (base, reset): the two-level variant factors later diagonal blocks with this kernel too -- `base` = first row of the
// block (LAPACK's info counts from the matrix origin), reset = 0 keeps an earlier failure.
// Factor + invert the 64 x 64 block at A (global, column-major) with the calling 256-thread workgroup; cs / sb: LDS scratch of
// CB * (CB + 1) and DIAG_SB doubles.
__device__ __forceinline__ void diag_from_global(double* __restrict__ A, int ld, double* __restrict__ W0, int* __restrict__ info, int base,
                                                 int reset, double* cs, double* sb, int nlive) {
  const int tid = threadIdx.x;
  for (int e = tid; e < CB * CB; e += 256) {
    const int r = e & 63, c = e >> 6;
    cs[r * (CB + 1) + c] = A[(size_t)c * ld + r];
  }
  __syncthreads();
  const int bad = diag_pipe(cs, sb, W0, A, ld, nlive, tid, nullptr);
  if (tid == 0) {
    if (reset)
      *info = bad;  // also resets the flag of the previous factorisation
    else if (bad != 0 && *info == 0)
      *info = base + bad;
  }
}

__global__ __launch_bounds__(256) void k_chol_first(double* __restrict__ A, int ld, double* __restrict__ W0, int* __restrict__ info,
                                                    int base = 0, int reset = 1, int nlive = CB) {
  __shared__ __attribute__((aligned(16))) double cs[CB * (CB + 1)];
  __shared__ __attribute__((aligned(16))) double sb[ED_LDS];
  diag_from_global(A, ld, W0, info, base, reset, cs, sb, nlive);
}

// ---------------------------------------------------------------------------------------------------------------
// Panel: rows below diagonal block k, 64 per workgroup.  P -> A[k0 + 64 + k0*ld] (in place), Wk = L_kk^-1.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chol_panel(const double* __restrict__ Wk, double* __restrict__ P, int ld) {
  __shared__ __attribute__((aligned(16))) double lds[CB * CPITCH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* __restrict__ Pb = P + (size_t)blockIdx.x * CB;
  stage_aside(lds, Wk, CB, tid);  // tile[kk][c] = W(c, kk)
  double bv[16];
  load_bside(bv, Pb, ld, w, lane);
  double acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mi][t] = 0.0;
  __syncthreads();
  mma_64(lds, bv, acc, lane);
  const int lk = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) Pb[(size_t)(16 * mi + 4 * t + lk) * ld + 16 * w + (lane & 15)] = acc[mi][t];
}

// ---------------------------------------------------------------------------------------------------------------
// Update with `np` consecutive, already solved block columns (panels) starting at column kp0: for every 64 x 64 output
// block (bi, bj), bj <= bi, of the region whose first row / column is o0:  A_ij -= sum_p X_p(i) X_p(j)^T, X_p = A[:, kp0 +
// 64 p ..].  The output block is read and written ONCE for the np rank-64 terms: at N = 8192 the plain right-looking sweep
// (np = 1) is bound by exactly that read-modify-write of a 512 MB trailing matrix.  Workgroup 0 also FACTORS and INVERTS
// the region's first diagonal block, so the serial chain of the factorisation runs beside the update.
// Grid: m * nc workgroups (m block rows, nc block columns; nc = m: the whole trailing triangle, nc = 1: only the next
// block column -- what a pair step applies first).  The panels are read-only here: disjoint from everything written.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chol_update(double* __restrict__ A, int ld, int kp0, int np, int o0, int m, int nc,
                                                     double* __restrict__ Wn, int* __restrict__ info, double* __restrict__ Pnext,
                                                     int tri_grid, int nlive) {
  __shared__ __attribute__((aligned(16))) double lds[CB * CPITCH];  // 40 KB: A-side tile, then the 64 x 65 block staging
  __shared__ __attribute__((aligned(16))) double sb[ED_LDS];
  int bi, bj;
  if (nc == m && tri_grid) {
    tri_index((int)blockIdx.x, bi, bj);
  } else {
    bi = blockIdx.x / nc;
    bj = blockIdx.x % nc;
    if (bj > bi) return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = o0 + CB * bi, j0 = o0 + CB * bj;  // first row / column of the output block
  const int lk = lane >> 4;
  const double* __restrict__ Xc = A + (size_t)kp0 * ld;

  stage_aside(lds, Xc + j0, ld, tid);  // tile[kk][c] = X(j0 + c, kk)
  double bv[16];
  load_bside(bv, Xc + i0, ld, w, lane);
  double acc[4][4];  // negated output tile: the MFMA accumulates X X^T - A
  double* __restrict__ Ab = A + (size_t)j0 * ld + i0;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mi][t] = -Ab[(size_t)(16 * mi + 4 * t + lk) * ld + 16 * w + (lane & 15)];
  __syncthreads();
  for (int p = 1; p < np; ++p) {
    // the next panel's B side is requested before this panel's MFMAs, its A side restaged after them
    double bn[16];
    const double* __restrict__ Xn = Xc + (size_t)p * CB * ld;
    load_bside(bn, Xn + i0, ld, w, lane);
    mma_64(lds, bv, acc, lane);
    __syncthreads();  // every wave is done reading the A-side tile
    stage_aside(lds, Xn + j0, ld, tid);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) bv[ks] = bn[ks];
    __syncthreads();
  }
  mma_64(lds, bv, acc, lane);

  if (bi != 0) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t) Ab[(size_t)(16 * mi + 4 * t + lk) * ld + 16 * w + (lane & 15)] = -acc[mi][t];
    if (Pnext != nullptr && bj == 0) {  // a copy of the next (still unsolved) panel for k_chol_step
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int t = 0; t < 4; ++t) Pnext[(size_t)(16 * mi + 4 * t + lk) * ld + i0 + 16 * w + (lane & 15)] = -acc[mi][t];
    }
    return;
  }
  // ---- next diagonal block: stage the updated block in LDS (row-major, pitch 65), factor, invert -------------
  __syncthreads();  // every wave is done reading the A-side tile
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) lds[(16 * w + (lane & 15)) * (CB + 1) + 16 * mi + 4 * t + lk] = -acc[mi][t];
  __syncthreads();
  const int bad = diag_pipe(lds, sb, Wn, Ab, ld, nlive, tid, nullptr);
  if (tid == 0 && bad != 0 && *info == 0) *info = i0 + bad;
}

// ---------------------------------------------------------------------------------------------------------------
// The diagonal chain as ONE resident workgroup beside the block-column kernels (r03; BOGP_CHOL_CHAIN=1).
//
// With k_chol_step a block column costs ~31.6 us of which the 64-pivot diagonal routine is 19.5: the rest is the launch
// boundary and workgroup (0, 0)'s way to its tile (stage W_k, load the panel tile, two 64^3 products, stores).  Here the
// critical path of the factorisation -- factor(k) -> X = A(k+1, k) W_k^T -> A(k+1, k+1) -= X X^T -> factor(k+1) -- never leaves
// ONE workgroup: W_k stays in LDS between steps, the two tiles a step needs from the previous block column's kernel arrive
// through a counter (flagT[t] == 2: tiles (1, 0) and (1, 1) of k_chol_step(t - 1) stored), and W_{t+1} goes out to the next
// block-column kernel through flagW[t + 1].  The block-column kernels (k_chol_step with flags) are launched back to back on
// the main stream and wait for their W inside; they no longer touch tile (0, 0).  No circular wait: k_chol_step(t) needs only
// flagW[t] (published before the chain starts step t), step t needs only k_chol_step(t - 1), which ran on flagW[t - 1].
// Every wait is bounded (chain_wait): on expiry *info = -7 and the work goes on with whatever is there -- a launch can fail, not hang.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool chain_wait(const unsigned int* flag, unsigned int target) {
  for (int spin = 0; spin < (1 << 18); ++spin) {  // ~0.3 s; a legitimate wait is below a millisecond
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      return true;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  return false;
}

__global__ void k_chain_init(unsigned int* __restrict__ flags, int nb, int kf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 2 * nb) flags[i] = i == kf ? 1u : (i == nb + kf ? 2u : 0u);  // flagW[kf] = 1, flagT[kf] = 2: step kf finds everything in place
}

__global__ __launch_bounds__(256) void k_chol_chain(double* __restrict__ A, int ld, int kf, int nb, double* __restrict__ Winv,
                                                    const double* __restrict__ scratch, int* __restrict__ info,
                                                    unsigned int* __restrict__ flags, int N) {
  __shared__ __attribute__((aligned(16))) double lds[CB * CPITCH];
  __shared__ __attribute__((aligned(16))) double sb[DIAG_SB];
  unsigned int* flagW = flags;
  unsigned int* flagT = flags + nb;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lk = lane >> 4;
  const int tr = tid >> 4, tc = tid & 15;
  stage_aside(lds, Winv + (size_t)kf * CB * CB, CB, tid);  // W_kf comes from the step before the chain (or k_chol_first)
  for (int t = kf; t + 1 < nb; ++t) {
    const int i0 = (t + 1) * CB;
    if (tid == 0 && !chain_wait(flagT + t, 2u) && *info == 0) *info = -7;
    __syncthreads();  // ... and W_t is staged
    const double* __restrict__ Pcur = scratch + (size_t)(t & 1) * ld * CB;
    double bv[16];
    load_bside(bv, Pcur + i0, ld, w, lane);  // the unsolved tile A(t + 1, t)
    double* __restrict__ Ab = A + (size_t)i0 * ld + i0;
    double acc[4][4], x[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[mi][q] = -Ab[(size_t)(16 * mi + 4 * q + lk) * ld + 16 * w + (lane & 15)];
        x[mi][q] = 0.0;
      }
    mma_64(lds, bv, x, lane);  // X = A(t + 1, t) W_t^T
    __syncthreads();           // every wave is done with the W tile
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        lds[(16 * mi + 4 * q + lk) * CPITCH + 16 * w + (lane & 15)] = x[mi][q];
        bv[4 * mi + q] = x[mi][q];
      }
    asm volatile("s_nop 7\n\ts_nop 7"
                 : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]), "+v"(bv[8]),
                   "+v"(bv[9]), "+v"(bv[10]), "+v"(bv[11]), "+v"(bv[12]), "+v"(bv[13]), "+v"(bv[14]), "+v"(bv[15]));
    __syncthreads();
    mma_64(lds, bv, acc, lane);  // -(A(t + 1, t + 1) - X X^T)
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) lds[(16 * w + (lane & 15)) * (CB + 1) + 16 * mi + 4 * q + lk] = -acc[mi][q];
    __syncthreads();
    double a[4][4], ww[4][4];
    const int bad = diag_factor_invert(lds, sb, a, ww, tid, max(0, min(CB, N - i0)));
    // L's diagonal tile and W_{t+1} go out with write-through (agent-scope) stores drained by every storing wave, then ONE relaxed
    // flag: an agent-scope release fence here is a write-back of the XCD's L2 (several us) on the critical path of every step
    {
      double* __restrict__ Wn = Winv + (size_t)(t + 1) * CB * CB;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * tr + i, col = 4 * tc + c;
          if (r >= col) __hip_atomic_store(Ab + (size_t)col * ld + r, a[i][c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(Wn + col * CB + r, tc <= tr ? ww[i][c] : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0 && bad != 0 && *info == 0) *info = i0 + bad;
    // W_{t+1} for the next step straight from the registers (tile[kk][c] = W(c, kk), zeros above the diagonal)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) lds[(4 * tc + cc) * CPITCH + 4 * tr + i] = tc <= tr ? ww[i][cc] : 0.0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flagW + t + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// One block column in ONE launch (small trailing matrices): panel solve + trailing update + next diagonal block.
// Every workgroup (bi, bj) recomputes the two panel tiles it needs, X_i = A[i, k] W_k^T and X_j, from the UNSOLVED panel
// (three 64^3 products per workgroup instead of one -- free while the trailing matrix has fewer tiles than the GPU has
// workgroup slots) and k_chol_panel with its launch boundary disappears from the serial chain (~6 of ~33 us per block column).
// The unsolved panel is read from a scratch copy Pcur (element (row, kk) at Pcur[row + kk * ld]): the workgroups of block
// column bj = 0 store the solved X_i into A (the final L) while others still read the unsolved tile, and they write the
// NEXT unsolved panel twice, into A and into Pnext (two scratch panels alternate between steps).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chol_step(double* __restrict__ A, int ld, int k0, int m, const double* __restrict__ Wk,
                                                   const double* __restrict__ Pcur, double* __restrict__ Pnext,
                                                   double* __restrict__ Wn, int* __restrict__ info, int tri_grid, int nlive,
                                                   unsigned int* __restrict__ flagW, unsigned int* __restrict__ flagT) {
  __shared__ __attribute__((aligned(16))) double lds[CB * CPITCH];
  __shared__ __attribute__((aligned(16))) double sb[ED_LDS];
  int bi, bj;
  if (tri_grid) {
    tri_index((int)blockIdx.x, bi, bj);
  } else {
    bi = blockIdx.x / m;
    bj = blockIdx.x % m;
    if (bj > bi) return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = k0 + CB * (1 + bi), j0 = k0 + CB * (1 + bj);
  const int lk = lane >> 4;
  const bool chained = flagW != nullptr;  // the diagonal chain runs in k_chol_chain: W_k arrives through a flag (see there)
  if (chained) {
    if (tid == 0 && !chain_wait(flagW + k0 / CB, 1u) && *info == 0) *info = -7;
    __syncthreads();
  }

  stage_aside(lds, Wk, CB, tid);  // tile[kk][c] = W(c, kk)
  double bv[16], bvi[16];
  load_bside(bv, Pcur + j0, ld, w, lane);
  if (bi != bj) load_bside(bvi, Pcur + i0, ld, w, lane);
  double xj[4][4], xi[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) xj[mi][t] = xi[mi][t] = 0.0;
  double acc[4][4];  // negated output tile, requested early
  double* __restrict__ Ab = A + (size_t)j0 * ld + i0;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mi][t] = -Ab[(size_t)(16 * mi + 4 * t + lk) * ld + 16 * w + (lane & 15)];
  __syncthreads();
  mma_64(lds, bv, xj, lane);  // X_j: rows 16 w .. of block j, element (row, col 16 mi + 4 t + lk)
  if (bi != bj) {
    mma_64(lds, bvi, xi, lane);
  } else {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t) xi[mi][t] = xj[mi][t];
  }
  __syncthreads();  // every wave is done with the W tile
  // A side of the update: tile[kk][c] = X_j(c, kk)
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) lds[(16 * mi + 4 * t + lk) * CPITCH + 16 * w + (lane & 15)] = xj[mi][t];
  if (bj == 0) {  // the solved panel tile of block row i is final: store L
    double* __restrict__ Lb = A + (size_t)k0 * ld + i0;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t) Lb[(size_t)(16 * mi + 4 * t + lk) * ld + 16 * w + (lane & 15)] = xi[mi][t];
  }
  if (chained && bi == 0) return;  // the chain workgroup updates and factors A(k + 1, k + 1) itself
  // B side: X_i(row 16 w + (lane & 15), kk = 4 ks + lk) is exactly xi[ks / 4][ks % 4] of this lane
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) bv[4 * mi + t] = xi[mi][t];
  // (register moves feeding inline-asm MFMAs: the hazard recogniser does not see the consumer)
  asm volatile("s_nop 7\n\ts_nop 7"
               : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]), "+v"(bv[8]),
                 "+v"(bv[9]), "+v"(bv[10]), "+v"(bv[11]), "+v"(bv[12]), "+v"(bv[13]), "+v"(bv[14]), "+v"(bv[15]));
  __syncthreads();
  mma_64(lds, bv, acc, lane);

  if (bi != 0) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t) Ab[(size_t)(16 * mi + 4 * t + lk) * ld + 16 * w + (lane & 15)] = -acc[mi][t];
    if (bj == 0) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int t = 0; t < 4; ++t) Pnext[(size_t)(16 * mi + 4 * t + lk) * ld + i0 + 16 * w + (lane & 15)] = -acc[mi][t];
    }
    if (chained && bi == 1) {  // tiles (1, 0) and (1, 1) are what the chain's next step consumes: hand them over
      __threadfence();
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(flagT + k0 / CB + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  __syncthreads();
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) lds[(16 * w + (lane & 15)) * (CB + 1) + 16 * mi + 4 * t + lk] = -acc[mi][t];
  __syncthreads();
  const int bad = diag_pipe(lds, sb, Wn, Ab, ld, nlive, tid, nullptr);
  if (tid == 0 && bad != 0 && *info == 0) *info = i0 + bad;
}

// ---------------------------------------------------------------------------------------------------------------
// V = L^-1, U = V^T and R^-1 = U U^T from the diagonal-block inverses the factorisation leaves behind.
//
// Recursive doubling instead of a sequential block substitution: with L = [[L11, 0], [L21, L22]],
//     V21 = -V22 (L21 V11),
// so level s (blocks of 64 * 2^s) is two batched launches of the same "C = B A^T" tile product the factorisation uses
// -- both operands with their non-contracted index contiguous in memory, which is why the transposed copy U is kept:
//     step 1   Tt = U11 L21^T           (Tt(c, r) = sum_k V11(k, c) L21(r, k);  U11 upper: k-blocks >= the row block)
//     step 2   V21 = -V22 Tt^T,  U12 = -Tt V22^T   (one launch, grid.z = 2;  V22 lower: k-blocks <= the row block)
// log2(nb) levels of fully parallel tile products replace rocSOLVER's trtri / potri and the three dependent
// triangular solves of the likelihood (Yt = V y, Ft = V 1, gamma = U rho are plain matrix-vector products).
// ---------------------------------------------------------------------------------------------------------------
struct TriArgs {
  const double* L;
  double* V;
  double* U;
  double* T;     // scratch, same shape
  double* Rinv;  // lower triangle (full diagonal tiles)
  int ld, nb, level;
};
enum { TG_TRTRI_T = 0, TG_TRTRI_V = 1, TG_TRTRI_U = 2, TG_UUT = 3 };

__global__ __launch_bounds__(256) void k_tri_gemm(TriArgs a, int mode0) {
  __shared__ __attribute__((aligned(16))) double lds[CB * CPITCH];
  const int mode = mode0 + (int)blockIdx.z;
  const int ld = a.ld;
  const double *Bs, *As;
  double* out;
  int kb0, kb1;
  double alpha;
  if (mode == TG_UUT) {
    const int bi = blockIdx.x / a.nb, bj = blockIdx.x % a.nb;
    if (bj > bi) return;
    Bs = a.U + (size_t)bi * CB;
    As = a.U + (size_t)bj * CB;
    // K-slice blockIdx.y of UUT_PARTS: the k-blocks [bi, nb) of a tile are long for small bi (one workgroup would walk
    // all nb of them); slices go to separate matrices that the consumers add, so the result stays deterministic
    const int per = (a.nb + UUT_PARTS - 1) / UUT_PARTS;
    out = a.Rinv + (size_t)blockIdx.y * ld * ld + (size_t)bi * CB + (size_t)bj * CB * ld;
    kb0 = max(bi, (int)blockIdx.y * per);
    kb1 = min(a.nb, ((int)blockIdx.y + 1) * per);
    alpha = 1.0;
  } else {
    const int nbb = 1 << a.level;
    const int o11 = 2 * (int)blockIdx.y * nbb, o22 = o11 + nbb;
    const int n22 = min(nbb, a.nb - o22);
    const int ti = blockIdx.x / nbb, tj = blockIdx.x % nbb;
    if (mode == TG_TRTRI_T) {  // rows: c in block 11, columns: r in block 22, k over block 11
      if (tj >= n22) return;
      Bs = a.U + (size_t)(o11 + ti) * CB + (size_t)o11 * CB * ld;
      As = a.L + (size_t)(o22 + tj) * CB + (size_t)o11 * CB * ld;
      out = a.T + (size_t)(o11 + ti) * CB + (size_t)(o22 + tj) * CB * ld;
      kb0 = ti;
      kb1 = nbb;
      alpha = 1.0;
    } else if (mode == TG_TRTRI_V) {  // rows: r in block 22, columns: c in block 11, k over block 22
      if (ti >= n22) return;
      Bs = a.V + (size_t)(o22 + ti) * CB + (size_t)o22 * CB * ld;
      As = a.T + (size_t)(o11 + tj) * CB + (size_t)o22 * CB * ld;
      out = a.V + (size_t)(o22 + ti) * CB + (size_t)(o11 + tj) * CB * ld;
      kb0 = 0;
      kb1 = ti + 1;
      alpha = -1.0;
    } else {  // TG_TRTRI_U: rows: c in block 11, columns: r in block 22, k over block 22
      if (tj >= n22) return;
      Bs = a.T + (size_t)(o11 + ti) * CB + (size_t)o22 * CB * ld;
      As = a.V + (size_t)(o22 + tj) * CB + (size_t)o22 * CB * ld;
      out = a.U + (size_t)(o11 + ti) * CB + (size_t)(o22 + tj) * CB * ld;
      kb0 = 0;
      kb1 = tj + 1;
      alpha = -1.0;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  double acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mi][t] = 0.0;
  // no explicit software pipeline: prefetching the next k-block into registers (measured) costs 222 VGPRs, halves the
  // occupancy and makes the launch 1.5x slower -- four resident workgroups per CU hide the load latency better
  for (int kb = kb0; kb < kb1; ++kb) {
    const size_t koff = (size_t)kb * CB * ld;
    __syncthreads();  // the previous tile has been consumed
    stage_aside(lds, As + koff, ld, tid);
    double bv[16];
    load_bside(bv, Bs + koff, ld, w, lane);
    __syncthreads();
    mma_64(lds, bv, acc, lane);
  }
  const int lk = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) out[(size_t)(16 * mi + 4 * t + lk) * ld + 16 * w + (lane & 15)] = alpha * acc[mi][t];
}

// diagonal blocks: V_kk = W_k, U_kk = W_k^T (dense 64 x 64 blocks, zeros included)
__global__ __launch_bounds__(256) void k_tri_base(const double* __restrict__ Winv, double* __restrict__ V, double* __restrict__ U, int ld) {
  const int k0 = blockIdx.x * CB;
  const double* Wk = Winv + (size_t)blockIdx.x * CB * CB;
  for (int e = threadIdx.x; e < CB * CB; e += 256) {
    const int r = e & 63, c = e >> 6;
    const double v = Wk[c * CB + r];
    V[(size_t)(k0 + c) * ld + k0 + r] = v;
    U[(size_t)(k0 + r) * ld + k0 + c] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Large matrices (ld >= BIG_LD = 3200, i.e. every N > 3072): the same products on 128 x 128 tiles.
//
// The 64 x 64 tile product above moves 64 KB of operands per 0.5 MFLOP (8 flop/B) and, at N = 8192, none of the 512 MB
// matrices stays in a cache: the inverse ran at 39 TF/s and the rank-64 trailing updates of the factorisation at
// 22 TF/s (profiles/r01_nll_n8192_kernel_stats.csv).  k_mm128 forms out(128 x 128) = alpha * Rside Cside^T (+ out):
//   * 4 waves x (64 x 64) outputs = 16 d4 accumulators per wave, two workgroups per CU; the operands come straight from global
//     memory (mm128_tile_direct below; through two LDS stages with a barrier per 16 k until r06);
//   * the MFMA's A operand is the COLUMN side, so that D's lane index (lane & 15) runs along the rows of the column-major
//     output: every store instruction writes whole row segments;
//   * workgroups walk the live tiles of a triangular product in order of decreasing K (or are dealt XCD-aware, mm_tile_of, where the
//     tiles are equally long);
//   * structure is exploited at tile level: tiles above the diagonal leave at once, K ranges follow the triangles.
// Wide first panels of the Cholesky (launch_chol_lower): the panel's block columns are factored by the 64-block kernels above with their
// updates confined to the panel, the trailing matrix then receives ONE rank-64 w update from k_mm128 (MM_SYRK) instead of w rank-64 ones.
// ---------------------------------------------------------------------------------------------------------------
namespace {
constexpr int MB = 128;        // tile edge
// r02-r05: 6144 (the staged tile core: 4096 5.1 vs 4.8 ms, 6144 9.9 vs 10.2, 8192 17.6 vs 20.2).  With the LDS-free core (mm128_tile_direct) the 128-tile products win wherever
// the general path runs at all (profiles/r06_mm128_direct_ab.txt: llf + gradient 3.52 -> 3.40 ms at N = 3584, 4.51 -> 4.00 at 4096, 6.86 -> 5.83 at 5120, 9.07 -> 7.64 at 6016), so
// every training set above the elimination's limit (N > 3072: bogp_set_train rounds the leading dimension to 128 there) takes them.
constexpr int BIG_LD = 3200;

struct MmTile {
  const double* Rs;  // row-side operand: element (row, k) at Rs[row + k * ldr]
  const double* Cs;  // column-side operand: element (col, k) at Cs[col + k * ldc]
  double* out;       // out(row, col) at out[row + col * ldo]
  int ldr, ldc, ldo;
  int k0, k1;        // K range, multiples of 16
  double alpha;
  int beta;          // 1: out += alpha * product
};

// ---------------------------------------------------------------------------------------------------------------
// The tile core: NO LDS and NO barrier (r06, last session) -- what k_contract16d did for the sweep (kernels_posterior.hip, DESIGN 5.2'').
// Both operands are k-major with the non-contracted index contiguous, so a wave fetches its own fragments straight from global memory:
// lane (k, i) takes the PAIR of rows 2 i, 2 i + 1 of a 32-row half with one global_load_dwordx4 (four 256-byte runs an instruction) --
// fragment f = 2 half + e stands for rows 32 half + 2 i + e, on both sides, and the epilogue stores pairs accordingly (256-byte runs,
// half as many store instructions).  A k-pair (8 k values) is 8 loads, issued as ONE block in front of its 32 MFMAs, two k-pairs in
// flight through three register slots; every address is a scalar base + the lane's constant 32-bit offset (kept opaque in the loop: no
// 64-bit VALU adds on the MFMAs' pipe).  The two waves that share a row (column) half of the tile meet in L1.  231 VGPRs, no spills,
// two workgroups a CU.  (The staged core of r02-r06 -- both operands through two LDS stages of 16 x 128, one barrier a k-block -- is in git
// de191e8 behind -DMM128_LDS; the two give the SAME BITS, per output element the products run over k in the same order from the same
// starting value, and this one is 7-20 % faster per launch: profiles/r06_mm128_direct_ab.txt.)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mm128_tile_direct(const MmTile& t) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w & 1, wn = w >> 1;  // this wave: rows 64 wm .., columns 64 wn ..
  const int lk = lane >> 4, li = lane & 15;
  d4 acc[4][4];  // [column fragment][row fragment]; fragment f = 2 half + e <-> index 32 half + 2 i + e of the wave's 64
  // D[i][j] of (ci, rj): lane 16 (i % 4) + j, register i / 4 -> column 64 wn + 32 (ci >> 1) + 2 (4 r + lk) + (ci & 1), rows 64 wm + 32 (rj >> 1) + 2 li + (rj & 1)
  if (t.beta) {
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double* __restrict__ o = t.out + (size_t)(64 * wn + 32 * (ci >> 1) + 2 * (4 * r + lk) + (ci & 1)) * t.ldo + 64 * wm + 2 * li;
#pragma unroll
        for (int hr = 0; hr < 2; ++hr) {
          const double2 v = *reinterpret_cast<const double2*>(o + 32 * hr);
          acc[ci][2 * hr][r] = t.alpha * v.x;
          acc[ci][2 * hr + 1][r] = t.alpha * v.y;
        }
      }
  } else {
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int rj = 0; rj < 4; ++rj) acc[ci][rj] = (d4){0.0, 0.0, 0.0, 0.0};
  }
  const int nkp = (t.k1 - t.k0) / 8;  // k-pairs (K ranges are multiples of 16)
  if (nkp > 0) {
    constexpr int DD = 2, DR = DD + 1;  // k-pairs in flight, register slots
    const int kp_last = nkp - 1;
    const char* const rbase = reinterpret_cast<const char*>(t.Rs + (size_t)t.k0 * t.ldr + 64 * wm);
    const char* const cbase = reinterpret_cast<const char*>(t.Cs + (size_t)t.k0 * t.ldc + 64 * wn);
    const size_t rstride = (size_t)8 * t.ldr * sizeof(double), cstride = (size_t)8 * t.ldc * sizeof(double);  // a k-pair
    unsigned voffR0 = (unsigned)(((size_t)lk * t.ldr + 2 * li) * sizeof(double));  // k-step 0 of the pair
    unsigned voffR1 = voffR0 + (unsigned)(4 * (size_t)t.ldr * sizeof(double));   // k-step 1
    unsigned voffC0 = (unsigned)(((size_t)lk * t.ldc + 2 * li) * sizeof(double));
    unsigned voffC1 = voffC0 + (unsigned)(4 * (size_t)t.ldc * sizeof(double));
    double2 rv[DR][2][2], cv[DR][2][2];  // [slot][k-step][half]
#define BOGP_MMD_LOADS(slot, kpc_)                                                        \
  do {                                                                                    \
    const char* pr_ = rbase + (size_t)(kpc_) * rstride;                                   \
    const char* pc_ = cbase + (size_t)(kpc_) * cstride;                                   \
    asm volatile("" : "+v"(voffR0), "+v"(voffR1), "+v"(voffC0), "+v"(voffC1));            \
    cv[slot][0][0] = *reinterpret_cast<const double2*>(pc_ + voffC0);                     \
    cv[slot][0][1] = *reinterpret_cast<const double2*>(pc_ + voffC0 + 256);               \
    rv[slot][0][0] = *reinterpret_cast<const double2*>(pr_ + voffR0);                     \
    rv[slot][0][1] = *reinterpret_cast<const double2*>(pr_ + voffR0 + 256);               \
    cv[slot][1][0] = *reinterpret_cast<const double2*>(pc_ + voffC1);                     \
    cv[slot][1][1] = *reinterpret_cast<const double2*>(pc_ + voffC1 + 256);               \
    rv[slot][1][0] = *reinterpret_cast<const double2*>(pr_ + voffR1);                     \
    rv[slot][1][1] = *reinterpret_cast<const double2*>(pr_ + voffR1 + 256);               \
  } while (0)
#define BOGP_MMD_KPAIR(u, kp_)                                                            \
  do {                                                                                    \
    BOGP_MMD_LOADS(((u) + DD) % DR, min((kp_) + DD, kp_last));                            \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    _Pragma("unroll") for (int h = 0; h < 2; ++h)                                         \
      _Pragma("unroll") for (int ci = 0; ci < 4; ++ci) {                                  \
        const double c_ = (ci & 1) ? cv[u][h][ci >> 1].y : cv[u][h][ci >> 1].x;           \
        _Pragma("unroll") for (int rj = 0; rj < 4; ++rj)                                  \
          mfma16(c_, (rj & 1) ? rv[u][h][rj >> 1].y : rv[u][h][rj >> 1].x, acc[ci][rj]);  \
      }                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
#pragma unroll
    for (int tt = 0; tt < DD; ++tt) BOGP_MMD_LOADS(tt, min(tt, kp_last));
    int kp = 0;
    for (; kp + DR <= nkp; kp += DR) {
#pragma unroll
      for (int u = 0; u < DR; ++u) BOGP_MMD_KPAIR(u, kp + u);
    }
    for (; kp < nkp; kp += DR) {
#pragma unroll
      for (int u = 0; u < DR; ++u)
        if (kp + u < nkp) BOGP_MMD_KPAIR(u, kp + u);
    }
#undef BOGP_MMD_KPAIR
#undef BOGP_MMD_LOADS
    BOGP_CHOL_DRAIN();
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int rj = 0; rj < 4; ++rj) asm volatile("" : "+v"(acc[ci][rj]));  // the stores' reads of the accumulators stay behind the drain
  }
  // (out = alpha * acc: with beta the accumulators started from alpha * out, and alpha^2 = 1)
#pragma unroll
  for (int ci = 0; ci < 4; ++ci)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double* __restrict__ o = t.out + (size_t)(64 * wn + 32 * (ci >> 1) + 2 * (4 * r + lk) + (ci & 1)) * t.ldo + 64 * wm + 2 * li;
#pragma unroll
      for (int hr = 0; hr < 2; ++hr)
        *reinterpret_cast<double2*>(o + 32 * hr) = make_double2(t.alpha * acc[ci][2 * hr][r], t.alpha * acc[ci][2 * hr + 1][r]);
    }
}


enum { MM_UUT = 0, MM_T = 1, MM_V = 2, MM_U = 3, MM_SYRK = 4, MM_GEN = 5 };
struct MmArgs {
  const double* L;
  double* V;
  double* U;
  double* T;
  double* Rinv;
  double* A;   // MM_SYRK: the matrix being factored
  int ld, nt;  // nt = ld / 128
  int nb2;     // MM_T / V / U: 128-tiles per diagonal block of this level
  int t0;      // MM_SYRK: first trailing tile;  kp0 / kp1: the panel's columns;  cj0 / cj1: column tiles updated
  int kp0, kp1;
  int cj0, cj1;
  int TI, TJ;  // logical tile grid of one z / y slice
  int fixed;   // experiment BOGP_MM128_FIXED=1
  const double* gR;  // MM_GEN: plain product out = Rs Cs^T over k < gK (launch_mm128_gen)
  const double* gC;
  double* gO;
  int gldr, gldc, gldo, gK;
  int order;   // 1: workgroups walk the live tiles in order of decreasing K (pairs flattened into x)
  int fp, nl;  // MM_T / V / U with order: full pairs of the level, live 128-tiles of block 22 in the partial last pair (0: none)
};

// workgroup -> tile.  Hardware deals consecutive workgroups round-robin to the 8 XCDs, each with its own L2.  The grid is
// walked in 8 x 8 SUPER tiles of 64 consecutive workgroups; inside one, XCD x = b % 8 gets the 2 x 4 block of tiles
// (rows 2 (x >> 1) .., columns 4 (x & 1) ..): its 8 workgroups share 2 row panels and 4 column panels through that L2
// (6 panel streams for 8 tiles), and EVERY XCD takes an eighth of every super tile -- the triangular products have rows
// whose K differs by 64x, and a first version that gave whole super tiles to XCDs left one XCD with 2.5x the average
// work (and small grids on a single XCD).
__device__ __forceinline__ bool mm_tile_of(int TI, int TJ, int& ti, int& tj) {
  const int SJ = (TJ + 7) / 8;
  const int b = blockIdx.x;
  const int s = b >> 6, r = b & 63;
  const int x = r & 7, y = r >> 3;
  const int si = s / SJ, sj = s - si * SJ;
  ti = si * 8 + (x >> 1) * 2 + (y >> 2);
  tj = sj * 8 + (x & 1) * 4 + (y & 3);
  return ti < TI && tj < TJ;
}
}  // namespace

__global__ __launch_bounds__(256, 2) void k_mm128(MmArgs a, int mode0) {
  const int mode = mode0 + (int)blockIdx.z;
  int ti, tj, pair = (int)blockIdx.y;
  if (mode == MM_GEN) {  // every tile is live and equally long; consecutive workgroups share the row panel
    MmTile g;
    ti = (int)blockIdx.x / a.TJ;
    tj = (int)blockIdx.x - ti * a.TJ;
    g.Rs = a.gR + (size_t)ti * MB;
    g.Cs = a.gC + (size_t)tj * MB;
    g.out = a.gO + (size_t)ti * MB + (size_t)tj * MB * a.gldo;
    g.ldr = a.gldr; g.ldc = a.gldc; g.ldo = a.gldo;
    g.k0 = 0; g.k1 = a.gK;
    g.alpha = 1.0; g.beta = 0;
    mm128_tile_direct(g);
    return;
  }
  if (a.order) {
    // Longest K first, only tiles that exist: the dispatcher hands out workgroups in index order as slots free up, and
    // consecutive indices go to different XCDs -- every XCD sees the same mix of long and short tiles.  (The products are not
    // bound by operand traffic -- identical times with every k-row at one address -- so the XCD-local super tiles of
    // mm_tile_of bought nothing and cost balance: profiles/r03_mm128_order_ab.txt.)  Empty workgroups in the queue would
    // steer the real ones onto a subset of the CUs (measured at N = 6144), hence the compaction over the partial last pair:
    // fp full pairs, and nl < nb2 live tile rows / columns of block 22 in one more pair.
    const int q = (int)blockIdx.x, nb2 = a.nb2, fp = a.fp, nl = a.nl;
    if (mode == MM_UUT || mode == MM_SYRK) {  // row ti has ti + 1 tiles (MM_UUT: K shrinks with ti)
      ti = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5);
      while ((ti + 1) * (ti + 2) / 2 <= q) ++ti;
      while (ti * (ti + 1) / 2 > q) --ti;
      tj = q - ti * (ti + 1) / 2;
    } else if (mode == MM_T) {  // K = (nb2 - ti) tiles; tj runs over the live columns of block 22
      const int per = fp * nb2 + nl;
      ti = q / per;
      const int rem = q - ti * per;
      if (rem < fp * nb2) { pair = rem / nb2; tj = rem - pair * nb2; }
      else { pair = fp; tj = rem - fp * nb2; }
    } else {  // MM_V (MM_U): K = (nb2 - prim) tiles with row (column) nb2 - 1 - prim of block 22, live in the last pair from prim >= nb2 - nl
      const int p0 = nb2 - nl, perA = fp * nb2, perB = perA + (nl > 0 ? nb2 : 0);
      const int cntA = p0 * perA;
      int rem;
      if (q < cntA) { ti = q / perA; rem = q - ti * perA; }
      else { const int q2 = q - cntA; ti = p0 + q2 / perB; rem = q2 - (ti - p0) * perB; }
      pair = rem / nb2;
      tj = rem - pair * nb2;
    }
  } else if (!mm_tile_of(a.TI, a.TJ, ti, tj)) return;
  const int ld = a.ld;
  MmTile t;
  t.ldr = t.ldc = t.ldo = ld;
  t.alpha = 1.0;
  t.beta = 0;
  if (mode == MM_UUT) {  // R^-1(i, j) = sum_{k >= i} U(i, k) U(j, k), lower tiles
    if (tj > ti) return;
    t.Rs = a.U + (size_t)ti * MB;
    t.Cs = a.U + (size_t)tj * MB;
    t.out = a.Rinv + (size_t)ti * MB + (size_t)tj * MB * ld;
    t.k0 = ti * MB;
    t.k1 = ld;
  } else if (mode == MM_SYRK) {  // A22(i, j) -= sum_{k in panel} P(i, k) P(j, k), lower tiles (diagonal tiles in full)
    if (tj > ti || tj < a.cj0 || tj >= a.cj1) return;
    t.Rs = a.A + (size_t)(a.t0 + ti) * MB;
    t.Cs = a.A + (size_t)(a.t0 + tj) * MB;
    t.out = a.A + (size_t)(a.t0 + ti) * MB + (size_t)(a.t0 + tj) * MB * ld;
    t.k0 = a.kp0;
    t.k1 = a.kp1;
    t.alpha = -1.0;
    t.beta = 1;
  } else {
    const int nb2 = a.nb2;
    const int o11 = 2 * pair * nb2, o22 = o11 + nb2;
    const int n22 = min(nb2, a.nt - o22);
    if (mode == MM_T) {  // Tt(c, r) = sum_{k >= c} U11(c, k) L21(r, k): c in block 11 (ti), r in block 22 (tj)
      if (tj >= n22) return;
      t.Rs = a.U + (size_t)(o11 + ti) * MB;
      t.Cs = a.L + (size_t)(o22 + tj) * MB;
      t.out = a.T + (size_t)(o11 + ti) * MB + (size_t)(o22 + tj) * MB * ld;
      t.k0 = (o11 + ti) * MB;
      t.k1 = o22 * MB;
    } else if (mode == MM_V) {  // V21(r, c) = -sum_{k <= r} V22(r, k) Tt(c, k): r in block 22 (ti), c in block 11 (tj)
      ti = nb2 - 1 - ti;  // K grows with r: the long tiles are dispatched first
      if (ti >= n22) return;
      t.Rs = a.V + (size_t)(o22 + ti) * MB;
      t.Cs = a.T + (size_t)(o11 + tj) * MB;
      t.out = a.V + (size_t)(o22 + ti) * MB + (size_t)(o11 + tj) * MB * ld;
      t.k0 = o22 * MB;
      t.k1 = (o22 + ti + 1) * MB;
      t.alpha = -1.0;
    } else {  // MM_U: U12(c, r) = -sum_{k <= r} Tt(c, k) V22(r, k): c in block 11 (ti), r in block 22 (tj)
      {  // K grows with r = the COLUMN here: walk the grid transposed and reversed so that the long tiles go first
        const int a_ = nb2 - 1 - ti;
        ti = tj;
        tj = a_;
      }
      if (tj >= n22) return;
      t.Rs = a.T + (size_t)(o11 + ti) * MB;
      t.Cs = a.V + (size_t)(o22 + tj) * MB;
      t.out = a.U + (size_t)(o11 + ti) * MB + (size_t)(o22 + tj) * MB * ld;
      t.k0 = o22 * MB;
      t.k1 = (o22 + tj + 1) * MB;
      t.alpha = -1.0;
    }
  }
  if (a.fixed) t.ldr = t.ldc = 0;  // experiment: every k-row of the operands at one address (no L2 / HBM traffic)
  mm128_tile_direct(t);
}

// U12 = V21^T for every pair of a level (the third product of the recursive doubling is a transposition of the second):
// 64 x 64 tiles through LDS, both sides coalesced.  grid (tiles of r in block 22, tiles of c in block 11, pairs).
__global__ __launch_bounds__(256) void k_transpose_v21(const double* __restrict__ V, double* __restrict__ U, int ld, int nt, int nb2) {
  __shared__ double tl[64][65];
  const int o11 = 2 * (int)blockIdx.z * nb2 * 2, o22 = o11 + nb2 * 2;  // in 64-blocks
  const int n22 = min(nb2 * 2, 2 * nt - o22);
  if ((int)blockIdx.x >= n22) return;
  const int r0 = (o22 + blockIdx.x) * 64, c0 = (o11 + blockIdx.y) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int c = ty; c < 64; c += 4) tl[c][tx] = V[(size_t)(c0 + c) * ld + r0 + tx];  // V21(r0 + tx, c0 + c)
  __syncthreads();
  for (int r = ty; r < 64; r += 4) U[(size_t)(r0 + r) * ld + c0 + tx] = tl[tx][r];  // U12(c0 + tx, r0 + r)
}

static bool big_path(int ld) {
  const char* e = getenv("BOGP_NO_BIG_FIT");
  return ld >= BIG_LD && ld % MB == 0 && !(e && atoi(e) != 0);
}

// Wide first panels of the large-N Cholesky (r06, launch_chol_lower): block columns per wide panel, in order; the block columns behind them run
// the one-level chain.  BOGP_BIG_CHOL = "0": none; a comma list ("32", "16,16", ...): that schedule (A/B runs, tools/ab/ab_big_chol.sh); unset: by size.
static int wide_panels(int ld, int* widths, int cap) {
  if (!big_path(ld)) return 0;
  const char* e = getenv("BOGP_BIG_CHOL");
  int n = 0;
  if (e) {
    while (*e && n < cap) {
      const int v = atoi(e);
      if (v > 0) widths[n++] = v;
      while (*e && *e != ',') ++e;
      if (*e == ',') ++e;
    }
    return n;
  }
  // by size: panels of 24 block columns while at least 72 block columns (4608 rows) stay behind them -- N = 6144 ... 7552: one, 7680 ... 9088: two, ...
  // (measured, profiles/r06_wide_panels_ab.txt: 16 ... 32 columns a panel and one panel more or less are within 0.1 ms of each other at every size;
  // behind 72 block columns the one-level chain's updates are within a third of the diagonal chain's floor and a wide panel buys nothing)
  const int nb = ld / CB;
  for (int k = 0; n < cap && nb - (k + 24) >= 72; k += 24) widths[n++] = 24;
  return n;
}

// (until r06 k_mm128 needed 73.7 KB of dynamic LDS, granted once per device through an atomic mask -- ADVICE r05; the tile core takes no LDS any more)
static hipError_t launch_mm128(MmArgs a, int mode, int TI, int TJ, int ny, int nz, hipStream_t st) {
  a.TI = TI;
  a.TJ = TJ;
  constexpr int order = 1;  // (the tile orders that lost: tools/ab/ab_mm128_order.sh, EXPERIMENTS.md)
  a.fixed = 0;
  a.order = order;
  if (order && nz == 1 && (mode != MM_SYRK || (a.cj0 == 0 && a.cj1 >= TI && TI == TJ))) {
    unsigned count;
    if (mode == MM_UUT || mode == MM_SYRK) {  // (SYRK: every live tile has the same K -- the compact triangular grid only drops the workgroups that would exit)
      count = (unsigned)(TI * (TI + 1) / 2);
    } else {  // ny pairs of diagonal blocks of nb2 tiles each; block 22 of the last pair may be cut by the matrix edge
      const int nb2 = a.nb2;
      const int n22_last = max(0, min(nb2, a.nt - (2 * (ny - 1) * nb2 + nb2)));
      a.fp = n22_last == nb2 ? ny : ny - 1;
      a.nl = n22_last == nb2 ? 0 : n22_last;
      count = mode == MM_T ? (unsigned)(nb2 * (a.fp * nb2 + a.nl))
                           : (unsigned)((nb2 - a.nl) * a.fp * nb2 + a.nl * (a.fp + (a.nl > 0 ? 1 : 0)) * nb2);
    }
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_mm128, dim3(count, 1, 1), 256, 0, st, a, mode);
    return hipGetLastError();
  }
  a.order = 0;
  const int nsuper = ((TI + 7) / 8) * ((TJ + 7) / 8);
  hipLaunchKernelGGL(k_mm128, dim3((unsigned)(nsuper * 64), ny, nz), 256, 0, st, a, mode);
  return hipGetLastError();
}

hipError_t launch_mm128_gen(const double* Rs, int ldr, const double* Cs, int ldc, double* out, int ldo, int TI, int TJ, int K,
                            hipStream_t st) {
  if (TI <= 0 || TJ <= 0) return hipSuccess;
  MmArgs a{};
  a.gR = Rs; a.gC = Cs; a.gO = out;
  a.gldr = ldr; a.gldc = ldc; a.gldo = ldo; a.gK = K;
  a.TI = TI; a.TJ = TJ;
  hipLaunchKernelGGL(k_mm128, dim3((unsigned)(TI * TJ), 1, 1), 256, 0, st, a, (int)MM_GEN);
  return hipGetLastError();
}

// V = L^-1 (lower) and U = V^T (upper), both ld x ld column-major; their other triangles must be zero on entry and stay
// zero.  T: scratch of the same shape.
hipError_t launch_tri_inverse(const double* L, const double* Winv, double* V, double* U, double* T, int ld, hipStream_t st) {
  const int nb = ld / CB;
  hipLaunchKernelGGL(k_tri_base, dim3(nb), 256, 0, st, Winv, V, U, ld);
  TriArgs a{L, V, U, T, nullptr, ld, nb, 0};
  const bool big = big_path(ld);
  for (int level = 0; (1 << level) < nb; ++level) {
    const int nbb = 1 << level;
    const int pairs = (nb + 2 * nbb - 1) / (2 * nbb);
    a.level = level;
    // Levels that merge blocks of >= 1024 use the 128 x 128 tile product; below that a level is too few 128-tiles to fill
    // the GPU and the 64 x 64 kernel wins -- measured per level at N = 8192 (merging blocks of 128 / 256 / 512 / 1024 / 2048 /
    // 4096): k_tri_gemm 25 / 50 / 134 / 439 / 1272 / 4432 us against k_mm128 153 / 273 / 278 / 322 / 1075 / 2683 us.
    constexpr int mm_min_level = 4;
    if (big && level >= mm_min_level) {
      MmArgs m{};
      m.L = L; m.V = V; m.U = U; m.T = T; m.ld = ld; m.nt = ld / MB; m.nb2 = nbb / 2;
      hipError_t e = launch_mm128(m, MM_T, m.nb2, m.nb2, pairs, 1, st);
      if (e != hipSuccess) return e;
      if ((e = launch_mm128(m, MM_V, m.nb2, m.nb2, pairs, 1, st)) != hipSuccess) return e;
      hipLaunchKernelGGL(k_transpose_v21, dim3(nbb, nbb, pairs), 256, 0, st, V, U, ld, m.nt, m.nb2);
      continue;
    }
    hipLaunchKernelGGL(k_tri_gemm, dim3(nbb * nbb, pairs, 1), 256, 0, st, a, (int)TG_TRTRI_T);
    hipLaunchKernelGGL(k_tri_gemm, dim3(nbb * nbb, pairs, 2), 256, 0, st, a, (int)TG_TRTRI_V);
  }
  return hipGetLastError();
}

// sum of the *nparts slices at Rinv + q*ld*ld (lower triangle, full diagonal tiles) = U U^T = L^-T L^-1
hipError_t launch_uut(const double* U, double* Rinv, int ld, hipStream_t st, int* nparts) {
  const int nb = ld / CB;
  if (big_path(ld)) {  // one slice: the long-K tiles are spread over the XCDs by the tile order instead of K-slicing
    MmArgs m{};
    m.U = const_cast<double*>(U); m.Rinv = Rinv; m.ld = ld; m.nt = ld / MB;
    *nparts = 1;
    return launch_mm128(m, MM_UUT, m.nt, m.nt, 1, 1, st);
  }
  *nparts = UUT_PARTS;
  TriArgs a{nullptr, nullptr, const_cast<double*>(U), nullptr, Rinv, ld, nb, 0};
  hipLaunchKernelGGL(k_tri_gemm, dim3(nb * nb, UUT_PARTS, 1), 256, 0, st, a, (int)TG_UUT);
  return hipGetLastError();
}

// the schedule launch_chol_lower takes for a matrix of leading dimension ld (bogp_chol_wide_panels; no device call)
int chol_wide_panels(int ld, int* widths, int cap) {
  int w[16];
  const int n = wide_panels(ld, w, 16), nb = ld / CB;
  int kept = 0, kend = 0;
  for (int p = 0; p < n; ++p) {  // launch_chol_lower's own conditions
    kend += w[p];
    if ((kend & 1) || nb - kend - 1 <= 32 + 1) break;
    if (kept < cap && widths) widths[kept] = w[p];
    ++kept;
  }
  return kept;
}

// ---------------------------------------------------------------------------------------------------------------
// identity padding of an ld x ld column-major matrix outside its leading N x N block
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_pad_identity(double* __restrict__ A, int N, int ld) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // runs over a whole row / column
  const int j = N + blockIdx.y;                  // a padding row / column
  if (i >= ld) return;
  const double v = i == j ? 1.0 : 0.0;
  A[(size_t)j * ld + i] = v;  // column j
  A[(size_t)i * ld + j] = v;  // row j
}
hipError_t launch_pad_identity(double* A, int N, int ld, hipStream_t st) {
  if (ld > N) hipLaunchKernelGGL(k_pad_identity, dim3((ld + 255) / 256, ld - N), 256, 0, st, A, N, ld);
  return hipGetLastError();
}

// L = chol(A) in place (lower, column-major, ld a multiple of 64 with identity padding).  Winv: ld x 64 doubles; block
// k holds L_kk^-1 (64 x 64 column-major) afterwards.  *info (device) = 0 or 1 + the first column with a non-positive
// pivot, as LAPACK reports it.
hipError_t launch_chol_lower(double* A, int ld, double* Winv, int* info, hipStream_t st, hipStream_t st2, hipEvent_t* ev,
                             double* scratch, int N, unsigned int* chain_flags) {
  if (N <= 0 || N > ld) N = ld;
  auto live = [&](int col0) { return max(0, min(CB, N - col0)); };  // data columns of the 64-block that starts at col0
  const int nb = ld / CB;
  hipLaunchKernelGGL(k_chol_first, dim3(1), 256, 0, st, A, ld, Winv, info, 0, 1, live(0));
  // BOGP_CHOL_GROUP=G > 1 (experiment, default 1): block columns in groups of G -- inside a group the update after panel k
  // touches only block column k + 1 (with all the group's panels so far), the group's LAST panel triggers ONE rank-64 G
  // update of everything to the right, so the trailing matrix is read and written nb / G times instead of nb times.
  // Measured 13.0 -> 13.0 / 13.2 / 13.4 ms per likelihood at N = 8192 for G = 2 / 3 / 4: the 64 x 64 tile kernel is bound
  // by its un-pipelined operand loads, not by that read-modify-write.  (Also measured, and removed again: the group's
  // trailing update as ONE k_mm128 SYRK with the diagonal duty in its tile (0, 0): 38-43 TF/s at K = 128 against 28-33,
  // but a longer chain per block column -- 13.5-13.8 ms for every threshold tried; tools/ab/ab_chol_group.sh.)
  constexpr int G = 1;
  // Block columns whose trailing matrix has at most `fuse_max` block rows run as ONE launch (k_chol_step: panel solve
  // recomputed per workgroup); needs `scratch` (2 * ld * 64 doubles) for the copies of the unsolved panels.
  constexpr int fuse_max = 32;  // (tools/ab/ab_chol_fuse.sh)
  const bool can_fuse = G == 1 && scratch != nullptr && fuse_max > 0;
  auto panel_copy = [&](int k) { return scratch + (size_t)(k & 1) * ld * CB; };
  if (can_fuse && nb - 1 >= 1 && nb - 1 <= fuse_max) {  // step 0 is fused already: prime its panel copy from A
    hipError_t e = hipMemcpy2DAsync(panel_copy(0) + CB, (size_t)ld * sizeof(double), A + CB, (size_t)ld * sizeof(double),
                                    (size_t)(ld - CB) * sizeof(double), CB, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return e;
  }
  constexpr int tri = 1;  // triangular grids (the r02 m x m grids whose upper half exits at once lost: tools/ab/ab_chol_tri.sh)
  // (r03 experiment, switch removed in r06) the fused block columns with their diagonal
  // chain in ONE resident workgroup (k_chol_chain) on the second stream instead of workgroup (0, 0) of every k_chol_step.
  // Measured (tools/ab/ab_chol_chain.sh, profiles/r03_chol_chain_ab.txt): with agent-scope release fences at the hand-overs every
  // block column got 1.3-2.4 us SLOWER (a fence is a write-back of the XCD's L2: what a kernel boundary costs anyway); with
  // write-through stores + relaxed flags the per-column time equals k_chol_step's (31 us: flag poll + acquire + two tile
  // loads + two 64^3 products + staging + stores are the same ~11 us around the 19.5-us diagonal routine whoever runs them)
  // and the two events + init kernel + second-stream launch add ~30 us per factorisation: N = 128 132 -> 165 us,
  // N = 2048 1.42 -> 1.44 ms, N = 4096 3.60 -> 3.69 ms per likelihood.  Same bits as the default path.
  constexpr int chain_on = 0;  // (the switch went in r06: the A/B above lost; k_chol_chain stays in the source for the record of what was measured)
  const int kf = can_fuse ? max(0, nb - 1 - fuse_max) : nb;  // first fused block column
  const bool chain = chain_on && chain_flags != nullptr && st2 != nullptr && ev != nullptr && can_fuse && tri && kf + 1 < nb;
  unsigned int* flagW = chain ? chain_flags : nullptr;
  unsigned int* flagT = chain ? chain_flags + nb : nullptr;
  // Wide first panels (r06).  The one-level chain reads and writes the whole trailing triangle once per block column: 537 MB at the first column of
  // N = 8192, 143 us = 3.75 TB/s -- its early block columns are HBM bound (profiles/r05_big_chol_trace.txt).  Inside a wide panel of w block columns
  // the rank-64 updates are confined to the panel's own columns (the strip: ld x 64 w doubles, it fits the Infinity Cache), and ONE rank-64 w product
  // on 128 x 128 tiles (k_mm128, compute bound at K >= 1024) brings the rest of the trailing matrix up to date; the diagonal chain of the panel runs
  // beside the strip updates (workgroup 0 of k_chol_update) as everywhere else.  Only the FIRST columns are worth it -- later the trailing triangle is
  // small enough for the one-level chain to be bound by its diagonal blocks -- so the schedule is a short list of widths, not a uniform panel size
  // (the uniform two-level variant of r02-r05, 8 block columns a panel with a look-ahead stream, lost to the one-level chain and is gone).
  int kstart = 0;
  {
    int widths[16];
    const int nw = wide_panels(ld, widths, 16);
    for (int p = 0; p < nw; ++p) {
      const int kbeg = kstart, kend = kbeg + widths[p];
      // k_mm128 works on 128-tiles; the chain behind the last wide panel must start with an unfused step (a fused one reads the copy of the unsolved
      // panel that the step before it leaves)
      if ((kend & 1) || nb - kend - 1 <= fuse_max + 1 || kend * CB >= N) break;
      for (int k = kbeg; k < kend; ++k) {
        const int m = nb - k - 1, k0 = k * CB, nc = kend - 1 - k;
        hipLaunchKernelGGL(k_chol_panel, dim3(m), 256, 0, st, Winv + (size_t)k * CB * CB, A + (size_t)k0 * ld + k0 + CB, ld);
        if (nc > 0)
          hipLaunchKernelGGL(k_chol_update, dim3(m * nc), 256, 0, st, A, ld, k0, 1, k0 + CB, m, nc, Winv + (size_t)(k + 1) * CB * CB, info,
                             (double*)nullptr, 0, live(k0 + CB));
      }
      MmArgs a{};
      a.A = A; a.ld = ld; a.nt = ld / MB;
      a.t0 = kend / 2; a.kp0 = kbeg * CB; a.kp1 = kend * CB;
      const int TT = a.nt - a.t0;
      a.cj0 = 0; a.cj1 = TT;
      hipError_t e = launch_mm128(a, MM_SYRK, TT, TT, 1, 1, st);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(k_chol_first, dim3(1), 256, 0, st, A + (size_t)kend * CB * (ld + 1), ld, Winv + (size_t)kend * CB * CB, info, kend * CB, 0,
                         live(kend * CB));
      kstart = kend;
    }
  }
  for (int kbeg = kstart; kbeg + 1 < nb; kbeg += G) {
    for (int k = kbeg; k < kbeg + G && k + 1 < nb; ++k) {
      const int k0 = k * CB;
      const int m = nb - k - 1;
      const bool next_fused = can_fuse && m - 1 >= 1 && m - 1 <= fuse_max;
      if (chain && k == kf) {  // everything before column kf is queued on st: start the chain behind it
        hipError_t e;
        hipLaunchKernelGGL(k_chain_init, dim3((2 * nb + 255) / 256), 256, 0, st, chain_flags, nb, kf);
        if ((e = hipEventRecord(ev[0], st)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(st2, ev[0], 0)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_chol_chain, dim3(1), 256, 0, st2, A, ld, kf, nb, Winv, scratch, info, chain_flags, N);
        if ((e = hipEventRecord(ev[1], st2)) != hipSuccess) return e;
      }
      if (can_fuse && m <= fuse_max) {
        hipLaunchKernelGGL(k_chol_step, dim3(tri ? m * (m + 1) / 2 : m * m), 256, 0, st, A, ld, k0, m, Winv + (size_t)k * CB * CB,
                           panel_copy(k), panel_copy(k + 1), Winv + (size_t)(k + 1) * CB * CB, info, tri, live(k0 + CB), flagW, flagT);
        continue;
      }
      hipLaunchKernelGGL(k_chol_panel, dim3(m), 256, 0, st, Winv + (size_t)k * CB * CB, A + (size_t)k0 * ld + k0 + CB, ld);
      const bool last = k == kbeg + G - 1;
      hipLaunchKernelGGL(k_chol_update, dim3(last ? (tri ? m * (m + 1) / 2 : m * m) : m), 256, 0, st, A, ld, kbeg * CB, k - kbeg + 1,
                         k0 + CB, m, last ? m : 1, Winv + (size_t)(k + 1) * CB * CB, info,
                         next_fused ? panel_copy(k + 1) : (double*)nullptr, tri, live(k0 + CB));
    }
  }
  if (chain) {  // the factor is complete when the chain has left its last step
    hipError_t e = hipStreamWaitEvent(st, ev[1], 0);
    if (e != hipSuccess) return e;
  }
  return hipGetLastError();
}

}  // namespace bogp
