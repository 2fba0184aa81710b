// bogp_chol_device.h -- the 64-block device routines the factorisation kernels (kernels_chol.hip) and the elimination kernels
// (kernels_elim.hip) share: the 64 x 64 x 64 product on the matrix cores and the diagonal block's factor + inverse.  One file with
// kernels_chol.hip until r05; split out in r06 (no code change).
#pragma once
#include <hip/hip_runtime.h>

#include "bogp_device.h"
#include "bogp_internal.h"

namespace bogp {

namespace {

constexpr int CB = 64;           // block size
constexpr int CPITCH = 64 + 16;  // LDS pitch (doubles) of a k-major tile: conflict-free rotated A-fragment reads

// v_mfma_f64_16x16x4_f64 accumulating in place in ARCHITECTURAL VGPRs: 64-cycle issue = the FP64 matrix peak (with AGPR
// accumulators the same instruction takes 130 cycles, tools/probes/ubench_mfma16.hip); one A and one B register per 2048 flop.
// Lanes: A = 16 k + i, B = 16 k + j, D[i][j] in lane 16 (i % 4) + j, component i / 4.
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma16(double a, double b, d4& c) {
  asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// 16 passes: nothing may read the last results before they have left the pipe
#define BOGP_CHOL_DRAIN() asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory")

// 1/sqrt(x): hardware estimate + ONE third-order (Halley) step, e = 1 - x y^2, y' = y (1 + e/2 + 3 e^2/8): five dependent
// operations after v_rsq_f64 instead of the eight of two Newton steps.  A dependent FP64 operation costs ~26 cycles on
// the 64-pivot chain of the diagonal block (tools/probes/ubench_diag.hip: 21.7 -> 19.5 us per 64 x 64 block together with the
// merged phases below); relative error ~ e0^3 (e0 ~ 2^-26) + one rounding, the same 2.2e-16 against LAPACK's factor.
__device__ __forceinline__ double rsqrt_nr(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double t = x * y;
  const double e = __builtin_fma(-t, y, 1.0);
  double p = __builtin_fma(0.375, e, 0.5);
  p = p * e;
  return __builtin_fma(y, p, y);
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
// the same two for pointers KNOWN to be global (address space 1): a pointer that comes out of memory (a BatchSlot's) is generic to the compiler,
// and generic loads are flat_load -- they also wait on the LDS counter and take no scalar base
typedef __attribute__((address_space(1))) double gdouble;
typedef double d2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) d2v gd2v;
__device__ __forceinline__ const gdouble* as_global(const double* p) { return (const gdouble*)p; }
__device__ __forceinline__ void stage_aside(double* lds, const gdouble* As, int lda, int tid) {
  const int srow = tid >> 5, scol = (tid & 31) * 2;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int kk = srow + 8 * p;
    const d2v v = *(const gd2v*)(As + (size_t)kk * lda + scol);
    *reinterpret_cast<d2v*>(&lds[kk * CPITCH + scol]) = v;
  }
}
__device__ __forceinline__ void load_bside(double (&bv)[16], const gdouble* Bs, int ldb, int w, int lane) {
  const int lk = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) bv[ks] = Bs[(size_t)(4 * ks + lk) * ldb + 16 * w + (lane & 15)];
}
// TRI = true: the staged side is a LOWER TRIANGULAR factor, tile[kk][c] = W(c, kk) = 0 for kk > c (the inverse of a diagonal block): the
// k-steps of column tile mi stop at 4 mi + 3 -- 40 of the 64 MFMAs, and the skipped ones added exact zeros (r05).
template <bool TRI = false>
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
    for (int mi = 0; mi < 4; ++mi)
      if (!TRI || ks < 4 * (mi + 1)) mfma16(trow[aoff + 16 * mi], bv[ks], c[mi]);
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
// Two barriers per 4 columns; the serial part is the 4x4 potf2 + inverse (done redundantly by the 16 threads of the block row).
// cs: 64 x 65 staging of the input block; sb: DIAG_SB doubles.  On return `lo` holds the L tile (tc <= tr) and z the
// W tile (tc <= tr).  Returns 0 or 1 + the first column with a non-positive pivot (LAPACK's info), workgroup-uniform.
constexpr int DIAG_SB = 16 + 4 * CB + 2;
// nlive: leading columns of the block that hold data -- the rest is the identity padding of the matrix, whose factor and
// inverse are the identity: the block steps past it are skipped (a block with 16 live columns takes 4 of the 16 steps).
__device__ __forceinline__ int diag_factor_invert(const double* cs, double* sb, double (&lo)[4][4], double (&z)[4][4], int tid,
                                                  int nlive = CB) {
  const int tr = tid >> 4, tc = tid & 15;
  double* dtile = sb;           // [4][4] the diagonal tile of the current step (lower part used)
  double* Y = sb + 16;          // [4][64]
  double* flag = sb + 16 + 256; // 1 + first bad column (as a double), 0 if none
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      z[i][c] = cs[(4 * tr + i) * (CB + 1) + 4 * tc + c];
      lo[i][c] = 0.0;
    }
  if (tid == 0) {
    flag[0] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) dtile[4 * i + c] = z[i][c];
  }
  __syncthreads();
  const int nsteps = min(16, (nlive + 3) >> 2);
  for (int jb = 0; jb < nsteps; ++jb) {
    // ---- A + B: the 16 threads of block row jb ALL factor and invert the 4x4 diagonal tile (published by its owner at the
    // end of the previous step) and go straight on to their own piece of the block row Y = M z(jb, :): one barrier and
    // one LDS round trip less per step than handing M from the diagonal thread to the others
    if (tr == jb) {
      double a[4][4], l[4][4], iv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c <= i; ++c) a[i][c] = dtile[4 * i + c];
      int bad = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double piv = a[j][j];
        const bool okp = piv > 0.0;
        bad = (!okp && bad == 0) ? 4 * jb + j + 1 : bad;
        piv = okp ? piv : 1.0;
        const double inv = rsqrt_nr(piv);
        iv[j] = inv;
#pragma unroll
        for (int i = j + 1; i < 4; ++i) l[i][j] = a[i][j] * inv;
#pragma unroll
        for (int i = j + 1; i < 4; ++i)
#pragma unroll
          for (int c = j + 1; c <= i; ++c) a[i][c] = __builtin_fma(-l[i][j], l[c][j], a[i][c]);
        double sq = piv * inv;  // sqrt(piv) to ~1 ulp, off the pivot chain
        sq = __builtin_fma(__builtin_fma(-sq, sq, piv), 0.5 * inv, sq);
        l[j][j] = sq;
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
      if (tc == jb) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            lo[i][c] = c <= i ? l[i][c] : 0.0;
            z[i][c] = mm[i][c];
            Y[i * CB + 4 * tc + c] = mm[i][c];
          }
        if (bad != 0 && flag[0] == 0.0) flag[0] = (double)bad;
      } else {
        double y[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k <= i; ++k) sacc = __builtin_fma(mm[i][k], z[k][c], sacc);
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
      if (tr == jb + 1 && tc == jb + 1) {  // the next diagonal tile is final: publish it for its block row
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) dtile[4 * i + c] = z[i][c];
      }
    }
    __syncthreads();
  }
  if (tr == tc && tr >= nsteps) {  // skipped identity blocks: z is still exactly the identity tile = its own factor and inverse
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) lo[i][c] = z[i][c];
  }
  return (int)flag[0];
}

// The 64 x 64 diagonal block by k_nll_small's pipelined scheme (r03; it replaced diag_factor_invert above in every kernel but the
// opt-in k_chol_chain): the block cut into
// 4 x 4 blocks, threads 0 .. 63 = the panel (one a ROW of a panel block: block row i = tid / 4), threads 64 .. 199 = one owner a
// block of the lower triangle, ONE barrier a 4-column step, the panel built one step ahead on copied-out blocks.  L and W = L^-1
// ARE the panels (L(i, kn) = P_kn[i] for i > kn; X = L^-T appears column by column: W(4 kn + c, 4 i + r) = X(i, kn)[r][c], i <= kn),
// so an owner's work ends at step bi (no -R^-1 phase) and both go straight to global memory: W (dense, zeros above the diagonal) into its
// 64 x 64 column-major buffer -- by the last wave, out of the panel in LDS, one step behind --, the lower triangle of L into the matrix at Ad
// by the panel threads (nullptr: not wanted).
// ~14 us a block (16 steps of ~2100 cycles) against diag_factor_invert's 19.5.  ED_LDS doubles of LDS scratch; returns LAPACK's info (valid in thread 0),
// *logsum (if given, thread 0) = sum(log diag L).
constexpr int ED_PITCH = 18;
constexpr int ED_LDS = 4 * 16 * ED_PITCH;
__device__ __forceinline__ int ed_pidx(int e, int i) { return (e >> 1) * (2 * ED_PITCH) + 2 * i + (e & 1); }
__device__ __forceinline__ void ed_tri_index(int q, int& bi, int& bj) {  // q = bi (bi + 1) / 2 + bj, bj <= bi
  bi = (int)((sqrtf(8.0f * q + 1.0f) - 1.0f) * 0.5f);
  while ((bi + 1) * (bi + 2) / 2 <= q) ++bi;
  while (bi * (bi + 1) / 2 > q) --bi;
  bj = q - bi * (bi + 1) / 2;
}
// what no panel thread writes: zeros above the diagonal of W, the identity of the padding (W and the lower triangle of L).  Independent of the
// factorisation: FILL = false leaves it to a caller whose other workgroups have done it (the elimination: k_elim_first fills every W_k of
// the evaluation), so that the 16 stores a lane are off the chain of diagonal blocks
__device__ __forceinline__ void diag_fill(double* __restrict__ Wn, double* __restrict__ Ad, int ld, int nlive, int tid) {
  const int nb = min(16, (nlive + 3) >> 2);
  for (int e = tid; e < CB * CB; e += 256) {
    const int r = e & 63, col = e >> 6;
    const bool pad = r >= 4 * nb || col >= 4 * nb;
    if (r < col || pad) Wn[col * CB + r] = r == col ? 1.0 : 0.0;
    if (Ad != nullptr && pad && r >= col) Ad[(size_t)col * ld + r] = r == col ? 1.0 : 0.0;
  }
}
template <bool FILL = true>
__device__ __forceinline__ int diag_pipe(const double* cs, double* scr, double* __restrict__ Wn, double* __restrict__ Ad, int ld, int nlive,
                                         int tid, double* logsum) {
  double* P = scr;                        // [2][16 * ED_PITCH], pair layout (ed_pidx)
  double* Raw = scr + 2 * 16 * ED_PITCH;  // [2][16 * ED_PITCH], element-major
  const int nb = min(16, (nlive + 3) >> 2);
  if (FILL) diag_fill(Wn, Ad, ld, nlive, tid);
  gd2v* Wg = (gd2v*)Wn;  // (known to be global memory: global_store instead of flat_store)
  const int ot = tid - 64;
  int bi = 0, bj = 0;
  if (ot >= 0) ed_tri_index(ot, bi, bj);
  const bool live = ot >= 0 && bi < nb;
  double T[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) T[r][c] = live ? cs[(4 * bi + r) * (CB + 1) + 4 * bj + c] : 0.0;
#define ED_PUBLISH(q_)                                                                                  \
  {                                                                                                     \
    const int q = (q_);                                                                                 \
    if (live && q < nb && (bj == q || bi == q)) {                                                       \
      double* rawb = Raw + (q & 1) * 16 * ED_PITCH;                                                     \
      if (bj == q) {                                                                                    \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                   \
          _Pragma("unroll") for (int c = 0; c < 4; ++c) rawb[(4 * r + c) * ED_PITCH + bi] = T[r][c];    \
      } else {                                                                                          \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                   \
          _Pragma("unroll") for (int c = 0; c < 4; ++c) rawb[(4 * r + c) * ED_PITCH + bj] = T[c][r];    \
      }                                                                                                 \
    }                                                                                                   \
  }
#define ED_RESTART(z_)                                                                                  \
  {                                                                                                     \
    const int z = (z_);                                                                                 \
    if (live && z < nb && bj == z) {                                                                    \
      _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                     \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) T[r][c] = 0.0;                                    \
    }                                                                                                   \
  }
  ED_PUBLISH(0)
  ED_RESTART(0)
  ED_PUBLISH(1)
  double pivm = 1.0;
  int pive = 0, bad_all = 0;
  __syncthreads();
  for (int kn = 0; kn <= nb; ++kn) {  // the panel threads: P_kn; the owners: update kn - 1
    if (tid < 64) {
      if (kn < nb) {
        const int i = min(tid >> 2, nb - 1), pr = tid & 3;
        double D[4][4], l[4][4], inv[4], Mr[4];
        const double* rawb = Raw + (kn & 1) * 16 * ED_PITCH;
#pragma unroll
        for (int c = 0; c < 4; ++c) Mr[c] = rawb[(4 * pr + c) * ED_PITCH + i];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c <= r; ++c) D[r][c] = rawb[(4 * r + c) * ED_PITCH + kn];
        if (kn > 0) {
          const double* q = P + ((kn - 1) & 1) * 16 * ED_PITCH;
          double Q[4][4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < 4; ++m) Q[r][m] = q[ed_pidx(4 * r + m, kn)];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) D[r][c] = ns_dot4_sub(D[r][c], Q[r][0], Q[c][0], Q[r][1], Q[c][1], Q[r][2], Q[c][2], Q[r][3], Q[c][3]);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            double sacc = Mr[c];
#pragma unroll
            for (int m = 0; m < 4; ++m) sacc = __builtin_fma(-T[0][m], Q[c][m], sacc);
            Mr[c] = sacc;
          }
        }
        double prod4 = 1.0;
        const int bad = ns_factor4_sub(D, l, inv, prod4);
        if (tid == 0) {
          if (bad && bad_all == 0) bad_all = 4 * kn + bad;
          int e2;
          pivm = frexp(pivm * prod4, &e2);
          pive += e2;
        }
        if (i == kn) {
#pragma unroll
          for (int c = 0; c < 4; ++c) Mr[c] = pr == c ? 1.0 : 0.0;
        }
        double* pdst = P + (kn & 1) * 16 * ED_PITCH;
        const bool mine = (tid >> 2) < nb;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double v = Mr[c];
#pragma unroll
          for (int m = 0; m < c; ++m) v = __builtin_fma(-T[0][m], l[c][m], v);
          v = v * inv[c];
          T[0][c] = v;
          if (mine) pdst[ed_pidx(4 * pr + c, i)] = v;
          if (Ad != nullptr && mine && i > kn) Ad[(size_t)(4 * kn + c) * ld + 4 * i + pr] = v;  // L(4 i + pr, 4 kn + c)
          if (Ad != nullptr && mine && i == kn && c <= pr) {  // row pr of the 4 x 4 factor (selected WITHOUT a run-time register index)
            const double lv = pr == 0 ? l[0][c] : (pr == 1 ? l[1][c < 2 ? c : 1] : (pr == 2 ? l[2][c < 3 ? c : 2] : l[3][c]));
            Ad[(size_t)(4 * kn + c) * ld + 4 * kn + pr] = lv;
          }
        }
      }
    } else if (kn > 0) {
      const int p = kn - 1;
      if (tid >= 192) {
        // W(4 p + c, 4 i + pr) = X(i, p)[pr][c], i <= p, out of the panel in LDS, one step behind the panel threads and by the wave with the
        // fewest owners: 64 lines a store instruction (W is column-major, a lane's four values are 32 contiguous bytes), which cost the
        // panel wave -- the one every step waits for -- 1.2 us a block when it issued them itself (r05, tools/probes/run_variants.sh)
        const int l = tid - 192, i = l >> 2, pr = l & 3;
        if (i < nb && i <= p) {
          const double* pp = P + (p & 1) * 16 * ED_PITCH;
          const d2v v01 = *reinterpret_cast<const d2v*>(pp + ed_pidx(4 * pr, i));
          const d2v v23 = *reinterpret_cast<const d2v*>(pp + ed_pidx(4 * pr + 2, i));
          gd2v* dst = Wg + ((4 * i + pr) * CB + 4 * p) / 2;
          dst[0] = v01;
          dst[1] = v23;
        }
      }
      if (live && p < bi) {  // R phase (p < bj) or X phase (bj <= p < bi); nothing after step bi
        const double* pp = P + (p & 1) * 16 * ED_PITCH;
        double pa[4][4], pb[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            pa[r][c] = pp[ed_pidx(4 * r + c, bi)];
            pb[r][c] = pp[ed_pidx(4 * r + c, bj)];
          }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            double sacc = T[r][c];
#pragma unroll
            for (int m = 0; m < 4; ++m) sacc = __builtin_fma(-pa[r][m], pb[c][m], sacc);
            T[r][c] = sacc;
          }
      }
      ED_RESTART(kn)
      ED_PUBLISH(kn + 1)
    }
    __syncthreads();
  }
#undef ED_PUBLISH
#undef ED_RESTART
  if (tid == 0 && logsum != nullptr) *logsum = log(pivm) + (double)pive * 0.6931471805599453;
  return bad_all;
}

// q -> (bi, bj), bj <= bi, row by row: only live tiles are launched (an m x m grid whose upper half exits at once costs
// dispatch time and skews the placement of the live workgroups over the CUs)
__device__ __forceinline__ void tri_index(int q, int& bi, int& bj) {
  bi = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5);
  while ((bi + 1) * (bi + 2) / 2 <= q) ++bi;
  while (bi * (bi + 1) / 2 > q) --bi;
  bj = q - bi * (bi + 1) / 2;
}

}  // namespace

}  // namespace bogp
