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
#include <atomic>
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

// mma_64 for two block rows against ONE staged tile, the accumulators held as MFMA operands throughout (k_elim_updateS_b keeps four tiles
// live and has no registers to spare for repacking): every A fragment read from LDS feeds both rows' MFMAs (half the LDS reads), the fragments of
// step ks + 1 are fetched while the eight MFMAs of step ks issue, and the scheduling fences keep the compiler from hoisting all 64
// fragment reads (128 VGPRs) above the chain.  Per output element the same sixteen accumulations in the same order as mma_64.
__device__ __forceinline__ void mma_64v2(const double* lds, const double (&bv0)[16], const double (&bv1)[16], d4 (&c0)[4], d4 (&c1)[4], int lane) {
  const int aoff = (lane >> 4) * CPITCH + (lane & 15);
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(c0[0]), "+v"(c0[1]), "+v"(c0[2]), "+v"(c0[3]), "+v"(c1[0]), "+v"(c1[1]), "+v"(c1[2]), "+v"(c1[3]));
  double an[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) an[mi] = lds[aoff + 16 * mi];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    double ac[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) ac[mi] = an[mi];
    if (ks + 1 < 16) {
      const double* trow = &lds[4 * (ks + 1) * CPITCH];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) an[mi] = trow[aoff + 16 * mi];
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      mfma16(ac[mi], bv0[ks], c0[mi]);
      mfma16(ac[mi], bv1[ks], c1[mi]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
               : "+v"(c0[0]), "+v"(c0[1]), "+v"(c0[2]), "+v"(c0[3]), "+v"(c1[0]), "+v"(c1[1]), "+v"(c1[2]), "+v"(c1[3]));
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

}  // namespace

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

// q -> (bi, bj), bj <= bi, row by row: only live tiles are launched (an m x m grid whose upper half exits at once costs
// dispatch time and skews the placement of the live workgroups over the CUs)
__device__ __forceinline__ void tri_index(int q, int& bi, int& bj) {
  bi = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5);
  while ((bi + 1) * (bi + 2) / 2 <= q) ++bi;
  while (bi * (bi + 1) / 2 > q) --bi;
  bj = q - bi * (bi + 1) / 2;
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
// Large matrices (ld >= 4096): the same products on 128 x 128 tiles.
//
// The 64 x 64 tile product above moves 64 KB of operands per 0.5 MFLOP (8 flop/B) and, at N = 8192, none of the 512 MB
// matrices stays in a cache: the inverse ran at 39 TF/s and the rank-64 trailing updates of the factorisation at
// 22 TF/s (profiles/r01_nll_n8192_kernel_stats.csv).  k_mm128 forms out(128 x 128) = alpha * Rside Cside^T (+ out):
//   * BOTH operands go through LDS (k-major tiles of 16 x 128, pitch 144 doubles: the four k-rows of a fragment read fall
//     into different 128-byte bank groups), register-staged global -> LDS one k-block ahead, two LDS buffers, one barrier
//     per k-block; 4 waves x (64 x 64) outputs = 16 d4 accumulators per wave, two workgroups per CU;
//   * the MFMA's A operand is the COLUMN side, so that D's lane index (lane & 15) runs along the rows of the column-major
//     output: every store instruction writes 128-byte row segments;
//   * workgroups are dealt to tiles XCD-aware (mm_tile_of): inside every 8 x 8 super tile each XCD owns a 2 x 4 block, so
//     its workgroups share operand panels through that XCD's L2 and all XCDs carry the same mix of long and short K;
//   * structure is exploited at tile level: tiles above the diagonal leave at once, K ranges follow the triangles.
// Two-level Cholesky: 256-wide panels are factored by the 64-block kernels above (updates confined to the panel), the
// trailing matrix then receives ONE rank-256 update from k_mm128 instead of four rank-64 ones.
// ---------------------------------------------------------------------------------------------------------------
namespace {
constexpr int MB = 128;        // tile edge
constexpr int MKB = 16;        // k rows per LDS stage
constexpr int MPT = MB + 16;   // LDS pitch in doubles
constexpr int BIG_LD = 6144;  // measured cross-over (tools/time_fit_big.py): 4096 5.1 vs 4.8 ms, 6144 9.9 vs 10.2, 8192 17.6 vs 20.2

struct MmTile {
  const double* Rs;  // row-side operand: element (row, k) at Rs[row + k * ldr]
  const double* Cs;  // column-side operand: element (col, k) at Cs[col + k * ldc]
  double* out;       // out(row, col) at out[row + col * ldo]
  int ldr, ldc, ldo;
  int k0, k1;        // K range, multiples of MKB
  double alpha;
  int beta;          // 1: out += alpha * product
};

__device__ __forceinline__ void mm128_tile(const MmTile& t, double* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w & 1, wn = w >> 1;  // this wave: rows 64 wm .., columns 64 wn ..
  const int lk = lane >> 4, li = lane & 15;
  d4 acc[4][4];  // [column fragment][row fragment]
  if (t.beta) {
    // out += alpha * product with alpha = +-1: start from alpha * out (exact) and scale by alpha at the end.  All 64 loads
    // of the tile are issued before anything waits on them (a read-modify-write in the epilogue serialises 64 round trips
    // per thread: measured 100 us per workgroup)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double* __restrict__ o = t.out + (size_t)(64 * wn + 16 * ci + 4 * r + lk) * t.ldo + 64 * wm + li;
#pragma unroll
        for (int rj = 0; rj < 4; ++rj) acc[ci][rj][r] = t.alpha * o[16 * rj];
      }
  } else {
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int rj = 0; rj < 4; ++rj) acc[ci][rj] = (d4){0.0, 0.0, 0.0, 0.0};
  }
  const int nkb = (t.k1 - t.k0) / MKB;
  if (nkb > 0) {
    const int srow = tid >> 6, scol = (tid & 63) * 2;
    const double* __restrict__ rbase = t.Rs + (size_t)t.k0 * t.ldr + scol;
    const double* __restrict__ cbase = t.Cs + (size_t)t.k0 * t.ldc + scol;
    double2 r0, r1, r2, r3, c0, c1, c2, c3;
#define BOGP_MM_LOAD(kb_)                                                                 \
  do {                                                                                    \
    const double* pr_ = rbase + (size_t)((kb_)*MKB + srow) * t.ldr;                       \
    const double* pc_ = cbase + (size_t)((kb_)*MKB + srow) * t.ldc;                       \
    r0 = *reinterpret_cast<const double2*>(pr_);                                          \
    r1 = *reinterpret_cast<const double2*>(pr_ + (size_t)4 * t.ldr);                      \
    r2 = *reinterpret_cast<const double2*>(pr_ + (size_t)8 * t.ldr);                      \
    r3 = *reinterpret_cast<const double2*>(pr_ + (size_t)12 * t.ldr);                     \
    c0 = *reinterpret_cast<const double2*>(pc_);                                          \
    c1 = *reinterpret_cast<const double2*>(pc_ + (size_t)4 * t.ldc);                      \
    c2 = *reinterpret_cast<const double2*>(pc_ + (size_t)8 * t.ldc);                      \
    c3 = *reinterpret_cast<const double2*>(pc_ + (size_t)12 * t.ldc);                     \
  } while (0)
#define BOGP_MM_STORE(buf_)                                                               \
  do {                                                                                    \
    double* qr_ = lds + (buf_)*2 * MKB * MPT + srow * MPT + scol;                         \
    double* qc_ = qr_ + MKB * MPT;                                                        \
    *reinterpret_cast<double2*>(qr_) = r0;                                                \
    *reinterpret_cast<double2*>(qr_ + 4 * MPT) = r1;                                      \
    *reinterpret_cast<double2*>(qr_ + 8 * MPT) = r2;                                      \
    *reinterpret_cast<double2*>(qr_ + 12 * MPT) = r3;                                     \
    *reinterpret_cast<double2*>(qc_) = c0;                                                \
    *reinterpret_cast<double2*>(qc_ + 4 * MPT) = c1;                                      \
    *reinterpret_cast<double2*>(qc_ + 8 * MPT) = c2;                                      \
    *reinterpret_cast<double2*>(qc_ + 12 * MPT) = c3;                                     \
  } while (0)
    const int roff = lk * MPT + 64 * wm + li, coff = MKB * MPT + lk * MPT + 64 * wn + li;
    BOGP_MM_LOAD(0);
    BOGP_MM_STORE(0);
    for (int kb = 0; kb < nkb; ++kb) {
      __syncthreads();  // block kb is in buffer kb & 1; every wave is done with the other buffer
      BOGP_MM_LOAD(min(kb + 1, nkb - 1));
      // keep the loads HERE: left alone, the scheduler sinks them below the MFMAs, next to the LDS stores that consume them
      // (fewer live registers), and every k-block then waits out the full global latency (measured: 20 TF/s)
      __builtin_amdgcn_sched_barrier(0);
      const double* tb = lds + (kb & 1) * 2 * MKB * MPT;
      double rf[2][4], cf[2][4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        rf[0][f] = tb[roff + 16 * f];
        cf[0][f] = tb[coff + 16 * f];
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < 3) {
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            rf[(ks + 1) & 1][f] = tb[roff + 4 * (ks + 1) * MPT + 16 * f];
            cf[(ks + 1) & 1][f] = tb[coff + 4 * (ks + 1) * MPT + 16 * f];
          }
          // (pinning these reads ahead of the MFMAs of k-step ks with sched_barrier / sched_group_barrier measured -1 %:
          // the second wave of the SIMD already covers the LDS round trips -- profiles/r03_mm128_order_ab.txt)
        }
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
          for (int rj = 0; rj < 4; ++rj) mfma16(cf[ks & 1][ci], rf[ks & 1][rj], acc[ci][rj]);
      }
      __builtin_amdgcn_sched_barrier(0);
      BOGP_MM_STORE((kb + 1) & 1);
    }
#undef BOGP_MM_LOAD
#undef BOGP_MM_STORE
    BOGP_CHOL_DRAIN();
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int rj = 0; rj < 4; ++rj) asm volatile("" : "+v"(acc[ci][rj]));  // the stores' reads of the accumulators stay behind the drain
  }
  // D[i][j]: i = column (MFMA A side), j = row; lane 16 (i % 4) + j, register i / 4
  // (out = alpha * acc: with beta the accumulators started from alpha * out, and alpha^2 = 1)
#pragma unroll
  for (int ci = 0; ci < 4; ++ci)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double* __restrict__ o = t.out + (size_t)(64 * wn + 16 * ci + 4 * r + lk) * t.ldo + 64 * wm + li;
#pragma unroll
      for (int rj = 0; rj < 4; ++rj) o[16 * rj] = t.alpha * acc[ci][rj][r];
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
  extern __shared__ __attribute__((aligned(16))) double mm_lds[];
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
    mm128_tile(g, mm_lds);
    return;
  }
  if (a.order && mode != MM_SYRK) {
    // Longest K first, only tiles that exist: the dispatcher hands out workgroups in index order as slots free up, and
    // consecutive indices go to different XCDs -- every XCD sees the same mix of long and short tiles.  (The products are not
    // bound by operand traffic -- identical times with every k-row at one address -- so the XCD-local super tiles of
    // mm_tile_of bought nothing and cost balance: profiles/r03_mm128_order_ab.txt.)  Empty workgroups in the queue would
    // steer the real ones onto a subset of the CUs (measured at N = 6144), hence the compaction over the partial last pair:
    // fp full pairs, and nl < nb2 live tile rows / columns of block 22 in one more pair.
    const int q = (int)blockIdx.x, nb2 = a.nb2, fp = a.fp, nl = a.nl;
    if (mode == MM_UUT) {  // row ti has ti + 1 tiles, K shrinks with ti
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
  mm128_tile(t, mm_lds);
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

// The two-level factorisation is kept behind BOGP_BIG_CHOL=1: its panel chain (a k_chol_panel + k_chol_update pair per block
// column, ~33 us each, bound by the 64-pivot diagonal block) runs beside the trailing update in the one-level variant
// (workgroup 0 of k_chol_update) but ahead of it here, and the look-ahead on a second stream does not hide it (the small
// chain kernels queue behind the update's workgroups): 9.5 ms against 8.9 ms at N = 8192.  The inverse and R^-1 do win.
static bool big_chol(int ld) {
  const char* e = getenv("BOGP_BIG_CHOL");
  return big_path(ld) && e && atoi(e) != 0;
}

// k_mm128's 73.7 KB of dynamic LDS must be granted once PER DEVICE (hipFuncSetAttribute acts on the current device's copy of the function),
// and bogp_nll runs on several host threads (the helper handles of bogp_nll_batch, the look-ahead of surrogate.py): one bit per device
// ordinal in an atomic word (r05 had a plain static bool: a second handle on another device would have launched without the grant).
static hipError_t mm128_grant_lds(int shm) {
  static std::atomic<unsigned long long> granted{0ull};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = (dev >= 0 && dev < 64) ? 1ull << dev : 0ull;  // (ordinals past 63: the attribute is set on every launch)
  if (bit && (granted.load(std::memory_order_acquire) & bit)) return hipSuccess;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mm128), hipFuncAttributeMaxDynamicSharedMemorySize, shm);
  if (e != hipSuccess) return e;
  if (bit) granted.fetch_or(bit, std::memory_order_release);
  return hipSuccess;
}

static hipError_t launch_mm128(MmArgs a, int mode, int TI, int TJ, int ny, int nz, hipStream_t st) {
  constexpr int shm = 2 * 2 * MKB * MPT * (int)sizeof(double);  // 73.7 KB: two stages of (row tile + column tile)
  {
    hipError_t e = mm128_grant_lds(shm);
    if (e != hipSuccess) return e;
  }
  a.TI = TI;
  a.TJ = TJ;
  static const int fixed = [] { const char* e = getenv("BOGP_MM128_FIXED"); return e ? atoi(e) : 0; }();
  a.fixed = fixed;
  static const int order = [] { const char* e = getenv("BOGP_MM128_ORDER"); return e ? atoi(e) : 1; }();
  a.order = order;
  if (order && mode != MM_SYRK && nz == 1) {
    unsigned count;
    if (mode == MM_UUT) {
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
    hipLaunchKernelGGL(k_mm128, dim3(count, 1, 1), 256, shm, st, a, mode);
    return hipGetLastError();
  }
  a.order = 0;
  const int nsuper = ((TI + 7) / 8) * ((TJ + 7) / 8);
  hipLaunchKernelGGL(k_mm128, dim3((unsigned)(nsuper * 64), ny, nz), 256, shm, st, a, mode);
  return hipGetLastError();
}

hipError_t launch_mm128_gen(const double* Rs, int ldr, const double* Cs, int ldc, double* out, int ldo, int TI, int TJ, int K,
                            hipStream_t st) {
  if (TI <= 0 || TJ <= 0) return hipSuccess;
  constexpr int shm = 2 * 2 * MKB * MPT * (int)sizeof(double);
  {
    hipError_t e = mm128_grant_lds(shm);
    if (e != hipSuccess) return e;
  }
  MmArgs a{};
  a.gR = Rs; a.gC = Cs; a.gO = out;
  a.gldr = ldr; a.gldc = ldc; a.gldo = ldo; a.gK = K;
  a.TI = TI; a.TJ = TJ;
  hipLaunchKernelGGL(k_mm128, dim3((unsigned)(TI * TJ), 1, 1), 256, shm, st, a, (int)MM_GEN);
  return hipGetLastError();
}

// two-level right-looking Cholesky: panels of 4 block columns (256); inside a panel the 64-block kernels with their updates
// confined to the panel; then one rank-256 update of the trailing matrix and the factorisation of its first block.
// With a second stream (st2 + two events) the update is split with LOOK-AHEAD: the two column tiles that form the NEXT
// panel are updated on the main stream, which then goes on factoring that panel (a serial chain of ~33 us per block
// column: 64 pivots each), while the rest of the trailing matrix is updated on st2 beside it.
static hipError_t launch_chol_lower_big(double* A, int ld, double* Winv, int* info, hipStream_t st, hipStream_t st2,
                                        hipEvent_t* ev) {
  const int nb = ld / CB;
  const bool ahead = st2 != nullptr && ev != nullptr;
  bool rest_pending = false;
  hipLaunchKernelGGL(k_chol_first, dim3(1), 256, 0, st, A, ld, Winv, info, 0, 1);
  static const int PWB = [] {  // block columns per panel (8 = 512 wide: measured 18.3 ms per llf+gradient at N = 8192 against 18.8 with 4 and 20.0 with 2)
    const char* e = getenv("BOGP_CHOL_PANEL");
    const int v = e ? atoi(e) : 8;
    return (v == 2 || v == 4 || v == 6 || v == 8) ? v : 4;
  }();
  for (int kbeg = 0; kbeg < nb; kbeg += PWB) {
    const int kend = min(nb, kbeg + PWB);
    for (int k = kbeg; k < kend; ++k) {
      const int m = nb - k - 1, k0 = k * CB;
      if (m == 0) break;
      hipLaunchKernelGGL(k_chol_panel, dim3(m), 256, 0, st, Winv + (size_t)k * CB * CB, A + (size_t)k0 * ld + k0 + CB, ld);
      const int nc = kend - 1 - k;  // block columns of this panel still to the right of k
      if (nc > 0)
        hipLaunchKernelGGL(k_chol_update, dim3(m * nc), 256, 0, st, A, ld, k0, 1, k0 + CB, m, nc,
                           Winv + (size_t)(k + 1) * CB * CB, info, (double*)nullptr, 0, CB);
    }
    if (kend < nb) {
      MmArgs a{};
      a.A = A; a.ld = ld; a.nt = ld / MB;
      a.t0 = kend / 2; a.kp0 = kbeg * CB; a.kp1 = kend * CB;
      const int TT = a.nt - a.t0;
      hipError_t e;
      if (!ahead || TT <= PWB / 2) {
        a.cj0 = 0; a.cj1 = TT;
        if ((e = launch_mm128(a, MM_SYRK, TT, TT, 1, 1, st)) != hipSuccess) return e;
      } else {
        // the previous panel's remainder update (st2) writes the same trailing columns: it has to be through first
        if (rest_pending && (e = hipStreamWaitEvent(st, ev[1], 0)) != hipSuccess) return e;
        const int nstrip = min(TT, PWB / 2);
        a.cj0 = 0; a.cj1 = nstrip;  // the next panel's column tiles
        if ((e = launch_mm128(a, MM_SYRK, TT, nstrip, 1, 1, st)) != hipSuccess) return e;
        if ((e = hipEventRecord(ev[0], st)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(st2, ev[0], 0)) != hipSuccess) return e;
        a.cj0 = nstrip; a.cj1 = TT;
        if ((e = launch_mm128(a, MM_SYRK, TT, TT, 1, 1, st2)) != hipSuccess) return e;
        if ((e = hipEventRecord(ev[1], st2)) != hipSuccess) return e;
        rest_pending = true;
      }
      hipLaunchKernelGGL(k_chol_first, dim3(1), 256, 0, st, A + (size_t)kend * CB * (ld + 1), ld, Winv + (size_t)kend * CB * CB, info,
                         kend * CB, 0);
    }
  }
  if (rest_pending) {
    hipError_t e = hipStreamWaitEvent(st, ev[1], 0);
    if (e != hipSuccess) return e;
  }
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
    static const int mm_min_level = [] {
      const char* e = getenv("BOGP_MM128_MIN_LEVEL");
      const int v = e ? atoi(e) : 4;
      return v < 1 ? 1 : v;
    }();
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
  if (big_chol(ld)) return launch_chol_lower_big(A, ld, Winv, info, st, st2, ev);
  hipLaunchKernelGGL(k_chol_first, dim3(1), 256, 0, st, A, ld, Winv, info, 0, 1, live(0));
  // BOGP_CHOL_GROUP=G > 1 (experiment, default 1): block columns in groups of G -- inside a group the update after panel k
  // touches only block column k + 1 (with all the group's panels so far), the group's LAST panel triggers ONE rank-64 G
  // update of everything to the right, so the trailing matrix is read and written nb / G times instead of nb times.
  // Measured 13.0 -> 13.0 / 13.2 / 13.4 ms per likelihood at N = 8192 for G = 2 / 3 / 4: the 64 x 64 tile kernel is bound
  // by its un-pipelined operand loads, not by that read-modify-write.  (Also measured, and removed again: the group's
  // trailing update as ONE k_mm128 SYRK with the diagonal duty in its tile (0, 0): 38-43 TF/s at K = 128 against 28-33,
  // but a longer chain per block column -- 13.5-13.8 ms for every threshold tried; tools/ab/ab_chol_group.sh.)
  static const int G = [] {
    const char* e = getenv("BOGP_CHOL_GROUP");
    const int v = e ? atoi(e) : 1;
    return (v >= 1 && v <= 8) ? v : 1;
  }();
  // Block columns whose trailing matrix has at most `fuse_max` block rows run as ONE launch (k_chol_step: panel solve
  // recomputed per workgroup); needs `scratch` (2 * ld * 64 doubles) for the copies of the unsolved panels.
  static const int fuse_max = [] {
    const char* e = getenv("BOGP_CHOL_FUSE_MAX");
    return e ? atoi(e) : 32;
  }();
  const bool can_fuse = G == 1 && scratch != nullptr && fuse_max > 0;
  auto panel_copy = [&](int k) { return scratch + (size_t)(k & 1) * ld * CB; };
  if (can_fuse && nb - 1 >= 1 && nb - 1 <= fuse_max) {  // step 0 is fused already: prime its panel copy from A
    hipError_t e = hipMemcpy2DAsync(panel_copy(0) + CB, (size_t)ld * sizeof(double), A + CB, (size_t)ld * sizeof(double),
                                    (size_t)(ld - CB) * sizeof(double), CB, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return e;
  }
  static const int tri = [] {  // 0: the r02 m x m grids whose upper half exits at once (A/B: tools/ab/ab_chol_tri.sh)
    const char* e = getenv("BOGP_CHOL_TRI_GRID");
    return e ? atoi(e) : 1;
  }();
  // BOGP_CHOL_CHAIN=1 (off by default; read per call so that the tests can run both): the fused block columns with their diagonal
  // chain in ONE resident workgroup (k_chol_chain) on the second stream instead of workgroup (0, 0) of every k_chol_step.
  // Measured (tools/ab/ab_chol_chain.sh, profiles/r03_chol_chain_ab.txt): with agent-scope release fences at the hand-overs every
  // block column got 1.3-2.4 us SLOWER (a fence is a write-back of the XCD's L2: what a kernel boundary costs anyway); with
  // write-through stores + relaxed flags the per-column time equals k_chol_step's (31 us: flag poll + acquire + two tile
  // loads + two 64^3 products + staging + stores are the same ~11 us around the 19.5-us diagonal routine whoever runs them)
  // and the two events + init kernel + second-stream launch add ~30 us per factorisation: N = 128 132 -> 165 us,
  // N = 2048 1.42 -> 1.44 ms, N = 4096 3.60 -> 3.69 ms per likelihood.  Same bits as the default path.
  const char* e_chain = getenv("BOGP_CHOL_CHAIN");
  const int chain_on = e_chain ? atoi(e_chain) : 0;
  const int kf = can_fuse ? max(0, nb - 1 - fuse_max) : nb;  // first fused block column
  const bool chain = chain_on && chain_flags != nullptr && st2 != nullptr && ev != nullptr && can_fuse && tri && kf + 1 < nb;
  unsigned int* flagW = chain ? chain_flags : nullptr;
  unsigned int* flagT = chain ? chain_flags + nb : nullptr;
  for (int kbeg = 0; kbeg + 1 < nb; kbeg += G) {
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
  static const int lo = [] { const char* e = getenv("BOGP_ELIM_STEP_PAIR_MIN"); return e ? atoi(e) : 110; }();
  static const int hi = [] { const char* e = getenv("BOGP_ELIM_STEP_PAIR_MAX"); return e ? atoi(e) : 500; }();
  return lo > 0 && grid >= lo && grid <= hi;
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
// fetches (profiles/r04_nll_batch_group.txt).  Here a workgroup owns the 2 x 2 blocks (2 BI + a, 2 BJ + b): per step it stages the TWO
// A-side tiles X_{2BJ}, X_{2BJ+1} and loads the TWO B-side fragments X_{2BI}, X_{2BI+1} for FOUR products -- half the fetches per
// flop.  Every block still sees the accumulations of k_elim_updateG_b on the same operands in the same order (its own first step ..
// klast; mma_64v2 issues mma_64's instructions for two block rows at once), then elim_store_block's stores: the same bits.  Blocks above
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
__device__ __forceinline__ void stage_aside_b(double* lds, __amdgpu_buffer_rsrc_t r, int tid) {  // stage_aside for a 64 x 64 tile, lda = 64
  const unsigned srow = tid >> 5, scol = (tid & 31) * 2;
  const unsigned voff = 8 * (srow * CB + scol);
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 8 * (p * 8 * CB), 0);
    *reinterpret_cast<v4u*>(&lds[(srow + 8 * p) * CPITCH + scol]) = v;
  }
}
__device__ __forceinline__ void load_bside_b(double (&bv)[16], __amdgpu_buffer_rsrc_t r, int w, int lane) {  // load_bside, ldb = 64
  const unsigned voff = (unsigned)(lane >> 4) * CB + 16 * w + (lane & 15);
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) bv[ks] = buf_load(r, voff, ks * 4 * CB);
}
__global__ __launch_bounds__(256, 2) void k_elim_updateS_b(const BatchSlot* __restrict__ slots, int k, int ng, int xcd, int G, int P) {
  __shared__ __attribute__((aligned(16))) double lds[2][CB * CPITCH];
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
  BI = __builtin_amdgcn_readfirstlane(BI);  // (tri_index goes through the vector ALU: tell the compiler the result is uniform, or every
  BJ = __builtin_amdgcn_readfirstlane(BJ);  //  buffer descriptor below gets a waterfall loop)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lk = lane >> 4;
  const unsigned lde = (unsigned)a.ld + CB;
  const size_t xsz = ((size_t)a.nb + 1) * CB * CB;
  const int klast = k + ng - 1;
  // per block: is it there, and the first step of the group it still needs (k_elim_updateG_b's rule).  The step loop itself has NO per-block
  // control flow -- all four products every step: a block that restarts at step f > k is simply zeroed when the loop gets there (what it
  // gathered before is discarded, as are the blocks above the diagonal / past the edge), so that the four accumulator tiles live in fixed
  // registers.  (With a branch per block the compiler keeps copies of the tiles across the variants and spills.)
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
  d4 acc[2][2][4];  // negated tiles, as MFMA accumulators: [mi] component t
#pragma unroll
  for (int ia = 0; ia < 2; ++ia)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      if (valid[ia][ib] && first[ia][ib] < 0) {
        int ldt;
        const __amdgpu_buffer_rsrc_t Tb = tile_rsrc(elim_tile(a, 2 * BI + ia, 2 * BJ + ib, ldt));
        const unsigned voff = (unsigned)lk * ldt + 16 * w + (lane & 15);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[ia][ib][mi][t] = -buf_load<BOGP_STATE_AUX>(Tb, voff, (unsigned)(16 * mi + 4 * t) * ldt);
      } else {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ia][ib][mi] = (d4){0.0, 0.0, 0.0, 0.0};
      }
    }
  const int tj1 = 2 * BJ + (2 * BJ + 1 <= a.nb ? 1 : 0), ti1 = 2 * BI + (2 * BI + 1 <= a.nb ? 1 : 0);  // (past the edge: the tile before, discarded)
  for (int sidx = k; sidx <= klast; ++sidx) {
    const double* xs = sl.xpanel + (size_t)(sidx & 3) * xsz;
    double bv0[16], bv1[16];
    if (sidx > k) __syncthreads();  // every wave is done with the previous step's tiles
    stage_aside_b(lds[0], tile_rsrc(xs + (size_t)(2 * BJ) * CB * CB), tid);
    stage_aside_b(lds[1], tile_rsrc(xs + (size_t)tj1 * CB * CB), tid);
    __builtin_amdgcn_sched_barrier(0);  // the staging registers are dead before the B-side fragments go live: 2 workgroups a CU (<= 256 VGPRs)
    load_bside_b(bv0, tile_rsrc(xs + (size_t)(2 * BI) * CB * CB), w, lane);
    load_bside_b(bv1, tile_rsrc(xs + (size_t)ti1 * CB * CB), w, lane);
#pragma unroll
    for (int ia = 0; ia < 2; ++ia)
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
        if (sidx == first[ia][ib]) {  // the restart: T <- 0 - X_i X_j^T from this step on
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) acc[ia][ib][mi] = (d4){0.0, 0.0, 0.0, 0.0};
        }
    __syncthreads();
    // step sidx:  -T <- -T + X_i X_j^T
    mma_64v2(lds[0], bv0, bv1, acc[0][0], acc[1][0], lane);
    mma_64v2(lds[1], bv0, bv1, acc[0][1], acc[1][1], lane);
  }
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
      const unsigned voff = (unsigned)lk * ldt + 16 * w + (lane & 15);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int t = 0; t < 4; ++t) buf_store<BOGP_STATE_AUX>(-acc[ia][ib][mi][t], Tb, voff, (unsigned)(16 * mi + 4 * t) * ldt);
      if (kn >= a.nb) continue;
      if (bj == kn && bi > kn) {
        const unsigned vo = (unsigned)lk * lde + 16 * w + (lane & 15);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int t = 0; t < 4; ++t) buf_store(-acc[ia][ib][mi][t], Pnext, vo, (unsigned)(16 * mi + 4 * t) * lde + (unsigned)(CB * bi));
      } else if (bi == kn && bj < kn) {
        const unsigned vo = (unsigned)(16 * w + (lane & 15)) * lde + lk;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int t = 0; t < 4; ++t) buf_store(-acc[ia][ib][mi][t], Pnext, vo, (unsigned)(CB * bj + 16 * mi + 4 * t));
      }
    }
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
  static const int group = [] {  // block columns per pass over the state: 4 (default), 2 (pair steps), 1; BOGP_ELIM_PAIRS=0 is the old name of 1
    const char* e = getenv("BOGP_ELIM_GROUP");
    const char* p2 = getenv("BOGP_ELIM_PAIRS");
    if (p2 && atoi(p2) == 0) return 1;
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
