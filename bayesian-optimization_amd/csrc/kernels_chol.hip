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
// Measured alternatives for the diagonal block (tools/ubench_potf2.hip, profiles/r01_ubench_potf2.txt): one wave with
// a row per lane and v_readlane / ds_bpermute / LDS broadcasts needs 50-300 us per block (SGPR pressure and spills).
#include "bogp_device.h"
#include "bogp_internal.h"

namespace bogp {

namespace {

constexpr int CB = 64;           // block size
constexpr int CPITCH = 64 + 16;  // LDS pitch (doubles) of a k-major tile: conflict-free rotated A-fragment reads

// v_mfma_f64_16x16x4_f64 accumulating in place in ARCHITECTURAL VGPRs: 64-cycle issue = the FP64 matrix peak (with AGPR
// accumulators the same instruction takes 130 cycles, tools/ubench_mfma16.hip); one A and one B register per 2048 flop.
// Lanes: A = 16 k + i, B = 16 k + j, D[i][j] in lane 16 (i % 4) + j, component i / 4.
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma16(double a, double b, d4& c) {
  asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// 16 passes: nothing may read the last results before they have left the pipe
#define BOGP_CHOL_DRAIN() asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory")

// 1/sqrt(x): hardware estimate + two Newton-Raphson steps (FMA form), ~1 ulp
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double e = __builtin_fma(-x * y, y, 1.0);
  y = __builtin_fma(y * 0.5, e, y);
  e = __builtin_fma(-x * y, y, 1.0);
  y = __builtin_fma(y * 0.5, e, y);
  return y;
}

// ---- 64x64x64 product on the matrix cores ---------------------------------------------------------------------
// acc[mi][t] += sum_kk Bside(row, kk) * Aside(col, kk) for the calling wave's 16 rows (16 w .. 16 w + 15) and all 64
// columns.  Aside is staged by the whole workgroup into LDS as tile[kk][col] (k-major, pitch CPITCH) from a
// column-major source with element (col, kk) at As[col + kk*lda]; Bside comes straight from global, element (row, kk)
// at Bs[row + kk*ldb].  Result element acc[mi][t] of this lane: row 16 w + (lane & 15), column 16 mi + 4 t + (lane >> 4).
__device__ __forceinline__ void stage_aside(double* lds, const double* __restrict__ As, int lda, int tid) {
  const int srow = tid >> 5, scol = (tid & 31) * 2;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int kk = srow + 8 * p;
    const double2 v = *reinterpret_cast<const double2*>(As + (size_t)kk * lda + scol);
    *reinterpret_cast<double2*>(&lds[kk * CPITCH + scol]) = v;
  }
}
__device__ __forceinline__ void load_bside(double (&bv)[16], const double* __restrict__ Bs, int ldb, int w, int lane) {
  const int lk = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) bv[ks] = Bs[(size_t)(4 * ks + lk) * ldb + 16 * w + (lane & 15)];
}
__device__ __forceinline__ void mma_64(const double* lds, const double (&bv)[16], double (&acc)[4][4], int lane) {
  const int aoff = (lane >> 4) * CPITCH + (lane & 15);  // MFMA-A = the LDS-staged side: lane (k, i) reads tile[k][16 mi + i]
  d4 c[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) c[mi] = (d4){acc[mi][0], acc[mi][1], acc[mi][2], acc[mi][3]};
  // The MFMAs are inline asm, invisible to the compiler's hazard recogniser: the VALU moves that build c (and whatever
  // register they recycle) must retire before the first MFMA reads c as SrcC -- these wait states are placed by hand.
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const double* trow = &lds[4 * ks * CPITCH];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) mfma16(trow[aoff + 16 * mi], bv[ks], c[mi]);
  }
  // the drain names the accumulators as in/out operands so that no read of them can be scheduled above it
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
               : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mi][t] = c[mi][t];
}

// ---- diagonal block: Cholesky factor and its inverse, blocked by 4 columns -----------------------------------------
// Thread (tr, tc) = (tid >> 4, tid & 15) owns ONE 4x4 register tile z of the symmetric block (rows 4 tr.., columns
// 4 tc..; both triangles are kept).  With M = (4x4 diagonal factor)^-1 of block step jb, the block row
//     Y = M z(jb, :)                                            (published by the 16 threads tr == jb)
// is at the same time the strip of L (L[4 tc + c][4 jb + k] = Y[k][4 tc + c] for tc > jb, by symmetry) and, left of the
// diagonal, the final block row of W = L^-1 -- so every thread below applies the SAME update z -= lr yc with
// lr[i][k] = Y[k][4 tr + i], yc[k][c] = Y[k][4 tc + c], whether its tile still belongs to the trailing block (tc > jb) or
// already accumulates W (tc <= jb; the tile switches role at jb == tc, where lr is its final piece of L).
// Two barriers per 4 columns; the serial part is the diagonal thread's 4x4 potf2 + inverse.
// cs: 64 x 65 staging of the input block; sb: DIAG_SB doubles.  On return `lo` holds the L tile (tc <= tr) and z the
// W tile (tc <= tr).  Returns 0 or 1 + the first column with a non-positive pivot (LAPACK's info), workgroup-uniform.
constexpr int DIAG_SB = 16 + 4 * CB + 2;
__device__ __forceinline__ int diag_factor_invert(const double* cs, double* sb, double (&lo)[4][4], double (&z)[4][4], int tid) {
  const int tr = tid >> 4, tc = tid & 15;
  double* mini = sb;            // [4][4] row-major, lower
  double* Y = sb + 16;          // [4][64]
  double* flag = sb + 16 + 256; // 1 + first bad column (as a double), 0 if none
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      z[i][c] = cs[(4 * tr + i) * (CB + 1) + 4 * tc + c];
      lo[i][c] = 0.0;
    }
  if (tid == 0) flag[0] = 0.0;
  for (int jb = 0; jb < 16; ++jb) {
    // ---- A: 4x4 potf2 + inverse by the diagonal thread ---------------------------------------------------
    if (tr == jb && tc == jb) {
      double l[4][4], iv[4];
      int bad = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double piv = z[j][j];
        if (!(piv > 0.0)) {
          if (bad == 0) bad = 4 * jb + j + 1;
          piv = 1.0;
        }
        const double inv = rsqrt_nr(piv);
        double sq = piv * inv;
        sq = __builtin_fma(__builtin_fma(-sq, sq, piv), 0.5 * inv, sq);  // sqrt(piv) to ~1 ulp
        l[j][j] = sq;
        iv[j] = inv;
#pragma unroll
        for (int i = j + 1; i < 4; ++i) l[i][j] = z[i][j] * inv;
#pragma unroll
        for (int i = j + 1; i < 4; ++i)
#pragma unroll
          for (int c = j + 1; c <= i; ++c) z[i][c] = __builtin_fma(-l[i][j], l[c][j], z[i][c]);
      }
      // M = l^-1 (lower), by forward substitution on the identity
      double mm[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < c) {
            mm[i][c] = 0.0;
          } else if (i == c) {
            mm[i][c] = iv[i];
          } else {
            double sacc = 0.0;
#pragma unroll
            for (int k = c; k < i; ++k) sacc = __builtin_fma(l[i][k], mm[k][c], sacc);
            mm[i][c] = -sacc * iv[i];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          lo[i][c] = c <= i ? l[i][c] : 0.0;
          z[i][c] = mm[i][c];
          mini[4 * i + c] = mm[i][c];
          Y[i * CB + 4 * tc + c] = mm[i][c];
        }
      if (bad != 0 && flag[0] == 0.0) flag[0] = (double)bad;
    }
    __syncthreads();
    // ---- B: block row Y = M z(jb, :) ---------------------------------------------------------------------
    if (tr == jb && tc != jb) {
      double y[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double sacc = 0.0;
#pragma unroll
          for (int k = 0; k <= i; ++k) sacc = __builtin_fma(mini[4 * i + k], z[k][c], sacc);
          y[i][c] = sacc;
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          z[i][c] = y[i][c];
          Y[i * CB + 4 * tc + c] = y[i][c];
        }
    }
    __syncthreads();
    // ---- C: rank-4 update of every tile below the block row ----------------------------------------------
    if (tr > jb) {
      double lr[4][4], yc[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          lr[i][k] = Y[k * CB + 4 * tr + i];
          yc[k][i] = Y[k * CB + 4 * tc + i];
        }
      if (tc == jb) {  // this tile's piece of L is final; from here on the registers accumulate W
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            lo[i][k] = lr[i][k];
            z[i][k] = 0.0;
          }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int k = 0; k < 4; ++k) z[i][c] = __builtin_fma(-lr[i][k], yc[k][c], z[i][c]);
    }
  }
  __syncthreads();
  return (int)flag[0];
}

// lower triangle of L into the matrix, W (dense, zeros above the diagonal) into its 64 x 64 column-major buffer
__device__ __forceinline__ void diag_store(double* __restrict__ Ad /* &A[i0 + i0*ld] */, int ld, double* __restrict__ Wk,
                                           const double (&a)[4][4], const double (&w)[4][4], int tid) {
  const int tr = tid >> 4, tc = tid & 15;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * tr + i, col = 4 * tc + c;
      if (r >= col) Ad[(size_t)col * ld + r] = a[i][c];
      Wk[col * CB + r] = tc <= tr ? w[i][c] : 0.0;  // W tiles carry their own zeros above the diagonal
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// First diagonal block (nothing to subtract yet).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chol_first(double* __restrict__ A, int ld, double* __restrict__ W0, int* __restrict__ info) {
  __shared__ __attribute__((aligned(16))) double cs[CB * (CB + 1)];
  __shared__ __attribute__((aligned(16))) double sb[DIAG_SB];
  const int tid = threadIdx.x;
  for (int e = tid; e < CB * CB; e += 256) {
    const int r = e & 63, c = e >> 6;
    cs[r * (CB + 1) + c] = A[(size_t)c * ld + r];
  }
  __syncthreads();
  double a[4][4], w[4][4];
  const int bad = diag_factor_invert(cs, sb, a, w, tid);
  diag_store(A, ld, W0, a, w, tid);
  if (tid == 0) *info = bad;  // also resets the flag of the previous factorisation
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
// Trailing update with block column k (X = A[k0+64:, k0:k0+64], already solved) + factorisation / inversion of the
// next diagonal block.  Grid: m*m workgroups, m = nb - k - 1; workgroup (bi, bj) with bj > bi leaves at once.
// Xc -> A[k0*ld] is its own read-only argument: that block column is disjoint from everything this kernel writes.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chol_update(double* __restrict__ A, const double* __restrict__ Xc, int ld, int k0, int m,
                                                     double* __restrict__ Wn, int* __restrict__ info) {
  __shared__ __attribute__((aligned(16))) double lds[CB * CPITCH];  // 40 KB: A-side tile, then the 64 x 65 block staging
  __shared__ __attribute__((aligned(16))) double sb[DIAG_SB];
  const int bi = blockIdx.x / m, bj = blockIdx.x % m;
  if (bj > bi) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = k0 + CB * (1 + bi), j0 = k0 + CB * (1 + bj);  // first row / column of the output block
  const int lk = lane >> 4;

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
  mma_64(lds, bv, acc, lane);

  if (bi != 0) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int t = 0; t < 4; ++t) Ab[(size_t)(16 * mi + 4 * t + lk) * ld + 16 * w + (lane & 15)] = -acc[mi][t];
    return;
  }
  // ---- next diagonal block: stage the updated block in LDS (row-major, pitch 65), factor, invert -------------
  __syncthreads();  // every wave is done reading the A-side tile
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int t = 0; t < 4; ++t) lds[(16 * w + (lane & 15)) * (CB + 1) + 16 * mi + 4 * t + lk] = -acc[mi][t];
  __syncthreads();
  double a[4][4], ww[4][4];
  const int bad = diag_factor_invert(lds, sb, a, ww, tid);
  diag_store(Ab, ld, Wn, a, ww, tid);
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

// V = L^-1 (lower) and U = V^T (upper), both ld x ld column-major; their other triangles must be zero on entry and stay
// zero.  T: scratch of the same shape.
hipError_t launch_tri_inverse(const double* L, const double* Winv, double* V, double* U, double* T, int ld, hipStream_t st) {
  const int nb = ld / CB;
  hipLaunchKernelGGL(k_tri_base, dim3(nb), 256, 0, st, Winv, V, U, ld);
  TriArgs a{L, V, U, T, nullptr, ld, nb, 0};
  for (int level = 0; (1 << level) < nb; ++level) {
    const int nbb = 1 << level;
    const int pairs = (nb + 2 * nbb - 1) / (2 * nbb);
    a.level = level;
    hipLaunchKernelGGL(k_tri_gemm, dim3(nbb * nbb, pairs, 1), 256, 0, st, a, (int)TG_TRTRI_T);
    hipLaunchKernelGGL(k_tri_gemm, dim3(nbb * nbb, pairs, 2), 256, 0, st, a, (int)TG_TRTRI_V);
  }
  return hipGetLastError();
}

// sum of the UUT_PARTS slices at Rinv + q*ld*ld (lower triangle, full diagonal tiles) = U U^T = L^-T L^-1
hipError_t launch_uut(const double* U, double* Rinv, int ld, hipStream_t st) {
  const int nb = ld / CB;
  TriArgs a{nullptr, nullptr, const_cast<double*>(U), nullptr, Rinv, ld, nb, 0};
  hipLaunchKernelGGL(k_tri_gemm, dim3(nb * nb, UUT_PARTS, 1), 256, 0, st, a, (int)TG_UUT);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// identity padding of an ld x ld column-major matrix outside its leading N x N block
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_pad_identity(double* __restrict__ A, int N, int ld) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // row
  const int j = blockIdx.y;                      // column
  if (i >= ld || (i < N && j < N)) return;
  A[(size_t)j * ld + i] = i == j ? 1.0 : 0.0;
}
hipError_t launch_pad_identity(double* A, int N, int ld, hipStream_t st) {
  hipLaunchKernelGGL(k_pad_identity, dim3((ld + 255) / 256, ld), 256, 0, st, A, N, ld);
  return hipGetLastError();
}

// L = chol(A) in place (lower, column-major, ld a multiple of 64 with identity padding).  Winv: ld x 64 doubles; block
// k holds L_kk^-1 (64 x 64 column-major) afterwards.  *info (device) = 0 or 1 + the first column with a non-positive
// pivot, as LAPACK reports it.
hipError_t launch_chol_lower(double* A, int ld, double* Winv, int* info, hipStream_t st) {
  const int nb = ld / CB;
  hipLaunchKernelGGL(k_chol_first, dim3(1), 256, 0, st, A, ld, Winv, info);
  for (int k = 0; k + 1 < nb; ++k) {
    const int k0 = k * CB;
    const int m = nb - k - 1;
    hipLaunchKernelGGL(k_chol_panel, dim3(m), 256, 0, st, Winv + (size_t)k * CB * CB, A + (size_t)k0 * ld + k0 + CB, ld);
    hipLaunchKernelGGL(k_chol_update, dim3(m * m), 256, 0, st, A, A + (size_t)k0 * ld, ld, k0, m,
                       Winv + (size_t)(k + 1) * CB * CB, info);
  }
  return hipGetLastError();
}

}  // namespace bogp
