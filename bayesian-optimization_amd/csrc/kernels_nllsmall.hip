// kernels_nllsmall.hip -- the WHOLE likelihood evaluation of a small training set (N <= 128, constant trend, one target) in ONE
// launch of ONE workgroup (gfx950).
//
// Why: at the sizes of an ordinary BO run an evaluation on the general path (bogp_api.hip: factorize + the gradient tail) is a
// chain of ~15 launches of 3-8 us each, whatever their arithmetic; a BO loop is 98 % such evaluations
// (profiles/r03_bo_loop.txt).  Here the correlation matrix never leaves the register file of one CU.
//
// What it computes (gpr.py:772-808 and :931-1038, the same quantities as the general path):
//   R = the per-mode normalised correlation matrix (k_build_R's expression), L = chol(R), Yt = L^-1 y, Ft = L^-1 1,
//   sum(log diag L), |Ft|, Ft.Yt, rho = Yt - Ft beta, rho.rho;  and for the gradient  R^-1 = L^-T L^-1,
//   gamma = L^-T (L^-1 y) - beta L^-T (L^-1 1),  the d + 1 contractions of k_grad_contract, trace(R^-1), gamma.gamma.
//
// How: the matrix is cut into 4 x 4 blocks, one THREAD per block of the lower triangle, block row nb (one extra) carries the
// right-hand sides [y; 1] -- their forward substitution is then nothing but the factorisation's own trsm + update.  The
// factorisation runs right-looking over the nb block columns; and it is continued IN PLACE into the inverse: bordering R with
// an identity, [[R, .], [I, 0]], the same elimination turns the identity into X = L^-T and the zero block into its Schur
// complement -X X^T = -R^-1.  X(j, i)^T, i > j, lives in the registers of thread (i, j) after its R block is finished (step j)
// and until step i; from step i on the same registers accumulate block (i, j) of -R^-1.  With P[i] = the block of panel k that
// block row i publishes (L(i, k) for i > k, X(i, k) for i <= k) EVERY thread does the same update at EVERY step,
//   T -= P[bi] P[bj]^T        (64 FMAs, operands from a 5-KB LDS panel),
// so the N^3 / 2 FMAs of factor + inverse + product are spread evenly over all threads and all steps, with two barriers per
// step.  Block row nb continued the same way ends as -(R^-1 y)^T, -(R^-1 1)^T: gamma, by the reference's cho_solve route.
#include "bogp_device.h"
#include "bogp_internal.h"

namespace bogp {

namespace {

constexpr int NS_BS = 4;           // block size
constexpr int NS_MAXNB = 39;       // N <= 156 (the 1024-thread instantiation; N <= 128 runs the 768-thread one)
constexpr int NS_PITCH = NS_MAXNB + 2;
constexpr int NS_THREADS = 1024;   // 4 (nb + 1) panel threads (3 waves) + (nb + 1)(nb + 2) / 2 - 1 owners: 132 + 560 at nb = 32, 160 + 819 at nb = 39
constexpr int NS_THREADS_128 = 768;  // N <= 128: 12 waves = 3 a SIMD = 168 registers a lane; 16 waves leave 128
constexpr int NS_WAVES = NS_THREADS / 64;

__device__ __forceinline__ double ns_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += shfl_xor_f64(v, o);
  return v;
}
// deterministic sum over the workgroup: butterfly inside a wave, the waves' sums added in wave order
__device__ __forceinline__ double ns_block_sum(double v, double* red /* [NS_WAVES] */, int nwaves) {
  v = ns_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < nwaves; ++w) s += red[w];
  return s;
}

// sqrt(p) and 1 / sqrt(p) of a pivot 0 < p (no range scaling: pivots of a correlation matrix lie in (1e-300, 4)): the hardware's
// reciprocal square root estimate + ONE third-order (Halley) step -- five dependent operations where sqrt() followed by a
// division is ~45; a dependent FP64 operation costs ~26 cycles in a lone wave and this chain is on the critical path of EVERY
// step (kernels_chol.hip: rsqrt_nr, same arithmetic)
__device__ __forceinline__ void ns_sqrt_rsqrt(double p, double& root, double& inv) {
  const double y = __builtin_amdgcn_rsq(p);
  const double t = p * y;
  const double e = __builtin_fma(-t, y, 1.0);
  double q = __builtin_fma(0.375, e, 0.5);
  q = q * e;
  inv = __builtin_fma(y, q, y);
  root = p * inv;
}

// 4 x 4 Cholesky of the lower triangle of a: l (strict lower part) and inv[c] = 1 / l_cc -- the panel's rows are then solved by
// substitution, o = M L^-T, column c of o as soon as pivot c is known: off the pivots' dependent chain except for one product
__device__ __forceinline__ int ns_factor4_sub(const double (&a)[4][4], double (&l)[4][4], double (&inv)[4], double& pivprod) {
  int bad = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double p = a[c][c];
#pragma unroll
    for (int m = 0; m < c; ++m) p = __builtin_fma(-l[c][m], l[c][m], p);
    if (!(p > 0.0) || !(p < 1e300)) {
      if (!bad) bad = c + 1;
      p = 1.0;
    }
    double lc;
    ns_sqrt_rsqrt(p, lc, inv[c]);
    l[c][c] = lc;
    pivprod *= lc;
#pragma unroll
    for (int r = c + 1; r < 4; ++r) {
      double v = a[r][c];
#pragma unroll
      for (int m = 0; m < c; ++m) v = __builtin_fma(-l[r][m], l[c][m], v);
      l[r][c] = v * inv[c];
    }
  }
  return bad;
}
// 4 x 4 Cholesky of the lower triangle of a, then W = L^-1 (lower); returns the 1-based index of the first non-positive
// pivot (0: none); pivprod *= l_00 l_11 l_22 l_33 (its logarithm is taken once, after the last step)
__device__ __forceinline__ int ns_factor4(const double (&a)[4][4], double (&w)[4][4], double& pivprod) {
  double l[4][4];
  double inv[4];
  int bad = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double p = a[c][c];
#pragma unroll
    for (int m = 0; m < c; ++m) p = __builtin_fma(-l[c][m], l[c][m], p);
    if (!(p > 0.0) || !(p < 1e300)) {
      if (!bad) bad = c + 1;
      p = 1.0;
    }
    double lc;
    ns_sqrt_rsqrt(p, lc, inv[c]);
    l[c][c] = lc;
    pivprod *= lc;
#pragma unroll
    for (int r = c + 1; r < 4; ++r) {
      double v = a[r][c];
#pragma unroll
      for (int m = 0; m < c; ++m) v = __builtin_fma(-l[r][m], l[c][m], v);
      l[r][c] = v * inv[c];
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    w[c][c] = inv[c];
#pragma unroll
    for (int r = 0; r < c; ++r) w[r][c] = 0.0;
  }
  // column by column of the inverse: w[r][c] = -(sum_{m = c}^{r - 1} l[r][m] w[m][c]) / l[r][r]
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = c + 1; r < 4; ++r) {
      double s = 0.0;
#pragma unroll
      for (int m = c; m < r; ++m) s = __builtin_fma(l[r][m], w[m][c], s);
      w[r][c] = -s * inv[r];
    }
  return bad;
}

// r0 = corr_profile(s2) and h = corr_dtheta_profile(s2, r0) with the square root and the exponential they share evaluated once
// (the same operations on the same values: bit-identical to the two calls)
template <int KERNEL>
__device__ __forceinline__ void ns_corr_pair(double s2, double& r0, double& h) {
  if (KERNEL == BOGP_KERNEL_SE || KERNEL == BOGP_KERNEL_ABSEXP) {
    r0 = exp(-s2);
    h = r0;
    return;
  }
  const double D = sqrt(s2);
  if (KERNEL == BOGP_KERNEL_MATERN12) {
    r0 = exp(-D);
    h = D > 0.0 ? 0.5 * r0 / D : 0.0;
  } else if (KERNEL == BOGP_KERNEL_MATERN32) {
    const double K = D * 1.7320508075688772;
    const double E = exp(-K);
    r0 = (1.0 + K) * E;
    h = 1.5 * E;
  } else {
    const double K = D * 2.23606797749979;
    const double E = exp(-K);
    r0 = (1.0 + K + (K * K) * 0.3333333333333333) * E;
    h = (5.0 / 6.0) * (1.0 + K) * E;
  }
}

}  // namespace

// out_scal: [0] sum(log diag L), [1] |Ft|, [2] Ft.Yt, [3] rho.rho, [62] the info word (int); out_S: the d + 1 contractions,
// [d + 1] trace(R^-1), [d + 2] gamma.gamma  (the layout the general path's read-back has, bogp_api.hip: fit_readback)
//
// Threads: the first 4 (nb + 1) are the PANEL threads, one a ROW of a block row's block of the current panel (1 - 3 waves); the
// others own the blocks.  Software pipeline, one barrier a step -- phase p:
//   owners: T -= P_p[bi] P_p[bj]^T; then those of column / row p + 1 restart from 0 (GRAD) and those of column / row p + 2 copy
//           their block (state after update p) to `Raw`;
//   panel threads, at the same time: take the copies of column / row p + 1 (state after update p - 1), apply update p themselves,
//           factor the diagonal block (every thread, the same instructions: no hand-over inside the panel), solve their row
//           against it and store P_{p+1}.
// The dependent chain of the 4 x 4 factorisation (~26 cycles an FP64 operation in a lone wave) sets the pace of a step,
// ~2000 cycles; the owners' 64 FMAs + 32 LDS reads a block hide behind it.
// The pair work before and after the loop (the correlation matrix; the gradient contractions) is spread over ALL threads in
// strips of 1 x 4 entries through an LDS image of the blocks, whatever the number of owners (15 at N = 16): it is bound by the
// exp / sqrt of each pair, ~100 instructions an entry.
template <int KERNEL, bool GRAD, int TMAX>
__global__ __launch_bounds__(TMAX) void k_nll_small(const NllSmallArgs a) {
  extern __shared__ double dyn[];      // Xs[N][dP] | Rst[16][nbR]: the blocks of R, later of R^-1, element-major
  __shared__ double P[2][16 * NS_PITCH];  // P[step & 1][e * NS_PITCH + i]: element e = 4 r + c of block row i's panel block
  __shared__ double Raw[2][16 * NS_PITCH];
  __shared__ double yt[NS_BS * NS_MAXNB], ft[NS_BS * NS_MAXNB], gam[NS_BS * NS_MAXNB];
  __shared__ double red[NS_WAVES];
  __shared__ double redk[NS_WAVES][65];
  __shared__ unsigned short blkmap[NS_MAXNB * (NS_MAXNB + 1) / 2];
  __shared__ double s_logdet;
  __shared__ int s_info;

#ifdef NS_PROFILE
  const long long tstart = clock64();
#endif
  const int N = a.N, d = a.d, nb = (N + NS_BS - 1) / NS_BS;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int nbR = nb * (nb + 1) / 2;              // blocks of R
  const int nown = (nb + 1) * (nb + 2) / 2 - 1;   // + block row nb; block (nb, nb) does not exist
  const int nwaves = nthr >> 6;
  const int dP = d | 1;
  double* Xs = dyn;
  double* Rst = dyn + (((size_t)N * dP + 1) & ~(size_t)1);
  // owner -> block (bi, bj), bi >= bj, row-major over the triangle
  const int npanel = ((4 * (nb + 1) + 63) / 64) * 64;  // panel threads: one a row of the (nb + 1) panel blocks, whole waves
  const int ot = tid - npanel;
  int bi = 0, bj = 0;
  if (ot >= 0) {
    bi = (int)((sqrt(8.0 * ot + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= ot) ++bi;
    while (bi * (bi + 1) / 2 > ot) --bi;
    bj = ot - bi * (bi + 1) / 2;
  }
  const bool live = ot >= 0 && ot < nown;
  const bool border = bi == nb;
  if (live && !border) blkmap[ot] = (unsigned short)((bi << 8) | bj);
  if (tid == 0) s_info = 0;
  for (int e = tid; e < N * d; e += nthr) Xs[(e / d) * dP + (e % d)] = a.X[e];
  __syncthreads();

  // ---- R, strip by strip, into the LDS image ---------------------------------------------------------------------
  for (int s = tid; s < 4 * nbR; s += nthr) {
    const int t = s >> 2, r = s & 3;
    const int sbi = blkmap[t] >> 8, sbj = blkmap[t] & 255;
    const int i = 4 * sbi + r;
    double s2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) s2[c] = dist_init<KERNEL>();
    if (i < N) {
      const double pexp = a.pexp;
      const double* xi = Xs + i * dP;
      const double* xj = Xs + min(4 * sbj, N - 1) * dP;  // (columns j <= i < N; the clamp only keeps padding reads in range)
      for (int k = 0; k < d; ++k) {
        const double th = a.theta[k], vi = xi[k];
#pragma unroll
        for (int c = 0; c < 4; ++c) s2[c] = dist_fold<KERNEL>(th, vi - xj[min(c, N - 1 - min(4 * sbj, N - 1)) * dP + k], s2[c], pexp);
      }
    }
    // (the four profiles evaluated unconditionally, side by side: a lone wave waits ~26 cycles for a dependent FP64 result, and a
    // branch per entry would string the four exp / sqrt chains one behind the other)
    double pv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) pv[c] = a.a * corr_profile<KERNEL>(s2[c]);
    if (a.div) {
#pragma unroll
      for (int c = 0; c < 4; ++c) pv[c] = pv[c] / a.b;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = 4 * sbj + c;
      double v = pv[c];
      if (i >= N || j >= N) v = i == j ? 1.0 : 0.0;  // identity padding
      else if (i == j) v = a.diag;
      Rst[(4 * r + c) * nbR + t] = v;
    }
  }
  __syncthreads();

  // ---- the owner's block of [R; y; 1] ----------------------------------------------------------------------------
  double T[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double v = 0.0;
      if (live && !border) v = Rst[(4 * r + c) * nbR + ot];
      else if (live && 4 * bj + c < N) v = r == 0 ? a.y[4 * bj + c] : (r == 1 ? 1.0 : 0.0);
      T[r][c] = v;
    }
  // Software pipeline, ONE barrier a step: in phase p the owners apply update p (panel P_p) while the panel wave already builds
  // P_{p+1}.  For that the blocks of column / row p + 1 were copied to `Raw` at the end of phase p - 1, in their state after update
  // p - 1, and the panel wave applies update p to its copy itself (lane i: M -= P_p[i] P_p[p + 1]^T, P_p[i] being its own output of
  // the phase before, kept in the T registers it has no other use for).  The owners' registers of those blocks restart from zero
  // at the END of phase p.  `P` and `Raw` are double-buffered by the parity of the step they belong to.
#define NS_PUBLISH(q_)                                                                        \
  {                                                                                           \
    const int q = (q_);                                                                       \
    if (live && q < nb && (bj == q || (GRAD && bi == q))) {                                   \
      double* rawb = Raw[q & 1];                                                              \
      if (bj == q) {                                                                          \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                         \
          _Pragma("unroll") for (int c = 0; c < 4; ++c) rawb[(4 * r + c) * NS_PITCH + bi] = T[r][c]; \
      } else {                                                                                \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                         \
          _Pragma("unroll") for (int c = 0; c < 4; ++c) rawb[(4 * r + c) * NS_PITCH + bj] = T[c][r]; \
      }                                                                                       \
    }                                                                                         \
  }
#define NS_RESTART(z_)                                                                        \
  {                                                                                           \
    const int z = (z_);                                                                       \
    if (GRAD && live && z < nb && (bj == z || bi == z)) {                                     \
      _Pragma("unroll") for (int r = 0; r < 4; ++r)                                           \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) T[r][c] = 0.0;                          \
    }                                                                                         \
  }
  // (the blocks of column q as they are, those of row q (bj < q) transposed; macros, not lambdas: a by-reference capture of T
  // left the block in scratch memory)
  NS_PUBLISH(0)
  NS_RESTART(0)
  NS_PUBLISH(1)
  double pivm = 1.0;  // lane 0 of the panel wave: prod(l_cc) = pivm 2^pive
  int pive = 0;
#ifdef NS_PROFILE
  long long tp0 = clock64(), tpA = 0, tpB = 0, tpC = 0, tpre = tp0;
  if (tid == 0) a.out_scal[20] = (double)(tp0 - tstart);
#endif
  __syncthreads();  // raw(0), raw(1) are published

  // ---- nb + 1 phases of factor / invert / multiply, in place ----------------------------------------------------
  for (int kn = 0; kn <= nb; ++kn) {  // the panel wave: P_kn; the owners: update kn - 1
    if (tid < npanel) {
      if (kn < nb) {
        // panel thread = ONE ROW of a block row's panel block: i = block row, pr = row in the block (T[0][.]: its output of the
        // phase before).  Every thread factors the diagonal block itself (same instructions in all lanes).
        const int i = min(tid >> 2, nb), pr = tid & 3;
        double D[4][4], l[4][4], inv[4], Mr[4];
        const double* rawb = Raw[kn & 1];
#pragma unroll
        for (int c = 0; c < 4; ++c) Mr[c] = rawb[(4 * pr + c) * NS_PITCH + i];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c <= r; ++c) D[r][c] = rawb[(4 * r + c) * NS_PITCH + kn];
        if (kn > 0) {
          // update kn - 1 of the copies: row pr of M -= P[i] P[kn]^T, and the diagonal block D -= P[kn] P[kn]^T from the panel in
          // LDS (the factorisation below then depends on 4 FMAs, not on another thread's row)
          const double* q = P[(kn - 1) & 1] + kn;
          double Q[4][4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < 4; ++m) Q[r][m] = q[(4 * r + m) * NS_PITCH];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) {
              double sacc = D[r][c];
#pragma unroll
              for (int m = 0; m < 4; ++m) sacc = __builtin_fma(-Q[r][m], Q[c][m], sacc);
              D[r][c] = sacc;
            }
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
          if (bad && s_info == 0) s_info = 4 * kn + bad;
          int e2;
          pivm = frexp(pivm * prod4, &e2);
          pive += e2;
        }
        if (i == kn) {  // X(kn, kn) = I L^-T
#pragma unroll
          for (int c = 0; c < 4; ++c) Mr[c] = pr == c ? 1.0 : 0.0;
        }
        // row pr of o = M L^-T:  L(i, kn) (i > kn);  X(i, kn) = Z^T L^-T (i < kn; its owner stored Z^T)
        const bool used = GRAD || i > kn;  // (rows above the diagonal carry nothing without the inverse: keep them finite)
        double* pdst = P[kn & 1];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double v = Mr[c];
#pragma unroll
          for (int m = 0; m < c; ++m) v = __builtin_fma(-T[0][m], l[c][m], v);
          v = used ? v * inv[c] : 0.0;
          T[0][c] = v;
          if ((tid >> 2) <= nb) pdst[(4 * pr + c) * NS_PITCH + i] = v;
        }
        if ((tid >> 2) == nb && pr < 2) {
#pragma unroll
          for (int c = 0; c < 4; ++c) (pr == 0 ? yt : ft)[4 * kn + c] = T[0][c];
        }
      }
#ifdef NS_PROFILE
      { const long long t = clock64(); tpB += t - tpre; tpre = t; }
#endif
    } else if (kn > 0) {
      const int p = kn - 1;
      if (live && (GRAD || bj > p)) {
        const double* pp = P[p & 1];
        double pa[4][4], pb[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            pa[r][c] = pp[(4 * r + c) * NS_PITCH + bi];
            pb[r][c] = pp[(4 * r + c) * NS_PITCH + bj];
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
      NS_RESTART(kn)
      NS_PUBLISH(kn + 1)
    }
    __syncthreads();
#ifdef NS_PROFILE
    { const long long t = clock64(); tpA += t - tpre; tpre = t; }
#endif
  }
#ifdef NS_PROFILE
  if (tid == 0) { a.out_scal[21] = (double)tpA; a.out_scal[22] = (double)tpB; a.out_scal[23] = (double)tpC; }
  const long long tloop_end = clock64();
#endif
  if (tid == 0) s_logdet = log(pivm) + (double)pive * 0.6931471805599453;
  __syncthreads();

  // ---- the likelihood's scalars (k_fit_rho's expressions) --------------------------------------------------------
  double sff = 0.0, sfy = 0.0;
  if (tid < N) {
    const double f = ft[tid];
    sff = f * f;
    sfy = f * yt[tid];
  }
  {  // both sums through one pair of barriers
    sff = ns_wave_sum(sff);
    sfy = ns_wave_sum(sfy);
    __syncthreads();
    if ((tid & 63) == 0) {
      redk[tid >> 6][0] = sff;
      redk[tid >> 6][1] = sfy;
    }
    __syncthreads();
    sff = sfy = 0.0;
    for (int w = 0; w < nwaves; ++w) {
      sff += redk[w][0];
      sfy += redk[w][1];
    }
  }
  const double nrm = sqrt(sff);
  double coef;
  if (a.estimate_trend) {
    const double G = -nrm, qty = sfy / G;
    coef = -(qty / G);
  } else {
    coef = -a.beta;
  }
  double srr = 0.0;
  if (tid < N) {
    const double rr = __builtin_fma(coef, ft[tid], yt[tid]);
    srr = rr * rr;
  }
  srr = ns_block_sum(srr, red, nwaves);
  if (tid == 0) {
    a.out_scal[0] = s_logdet;
    a.out_scal[1] = nrm;
    a.out_scal[2] = sfy;
    a.out_scal[3] = srr;
    double iw = 0.0;
    int info = s_info;
    memcpy(&iw, &info, sizeof(info));
    a.out_scal[62] = iw;
  }

  if (GRAD) {
    // gamma = R^-1 y - beta R^-1 1 from block row nb; R^-1 = -T into the LDS image
    if (live && border) {
#pragma unroll
      for (int c = 0; c < 4; ++c) gam[4 * bj + c] = -__builtin_fma(coef, T[1][c], T[0][c]);
    } else if (live) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) Rst[(4 * r + c) * nbR + ot] = -T[r][c];
    }
    __syncthreads();
    const double s2t = a.mode == BOGP_MODE_NOISY ? a.s2t_host : (a.mode == BOGP_MODE_NOISELESS ? srr / (N - (a.estimate_trend ? 1 : 0)) : srr / N);
    const double cw = 1.0 / s2t;
    // the pairs i > j of the strict lower triangle, <= 4 strips a thread: A = cw gamma_i gamma_j - Rinv_ij
    double B[4][4];
    int si[4], sj[4];
    double sd = 0.0, tr = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int s = tid + q * nthr;
      si[q] = -1;
      sj[q] = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) B[q][c] = 0.0;
      if (s < 4 * nbR) {
        const int t = s >> 2, r = s & 3;
        const int sbi = blkmap[t] >> 8, sbj = blkmap[t] & 255;
        const int i = 4 * sbi + r;
        if (i < N) {
          si[q] = i;
          sj[q] = 4 * sbj;
          double s2[4] = {0.0, 0.0, 0.0, 0.0};
          const double* xi = Xs + i * dP;
          for (int k = 0; k < d; ++k) {
            const double th = a.theta[k], vi = xi[k];
#pragma unroll
            for (int c = 0; c < 4; ++c) s2[c] += dist_term<KERNEL>(th, Xs[min(4 * sbj + c, N - 1) * dP + k] - vi);
          }
          const double gi = gam[i];
          double r0[4], hh[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) ns_corr_pair<KERNEL>(s2[c], r0[c], hh[c]);  // (side by side, see the prologue)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int j = 4 * sbj + c;
            const double rinv = Rst[(4 * r + c) * nbR + t];
            const double A = (gam[min(j, N - 1)] * gi) * cw - rinv;
            B[q][c] = j < i ? A * hh[c] : 0.0;
            sd += j < i ? A * r0[c] : 0.0;
            tr += j == i ? rinv : 0.0;
          }
        }
      }
    }
    double gg = tid < N ? gam[tid] * gam[tid] : 0.0;
    {  // the three sums through ONE pair of barriers, their butterflies side by side
      sd = ns_wave_sum(sd);
      tr = ns_wave_sum(tr);
      gg = ns_wave_sum(gg);
      __syncthreads();
      if ((tid & 63) == 0) {
        redk[tid >> 6][0] = sd;
        redk[tid >> 6][1] = tr;
        redk[tid >> 6][2] = gg;
      }
      __syncthreads();
      sd = tr = gg = 0.0;
      for (int w = 0; w < nwaves; ++w) {
        sd += redk[w][0];
        tr += redk[w][1];
        gg += redk[w][2];
      }
    }
    for (int k0 = 0; k0 < d; k0 += 64) {
      const int kn = min(64, d - k0);
      __syncthreads();
      for (int kk = 0; kk < kn; kk += 4) {  // four dimensions at a time: four independent butterflies fill each other's latency
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (si[q] >= 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int kq = min(k0 + kk + u, d - 1);  // (a dimension past d - 1 is computed and not stored)
              const double vi = Xs[si[q] * dP + kq];
#pragma unroll
              for (int c = 0; c < 4; ++c) acc[u] += B[q][c] * (-dtheta_weight<KERNEL>(Xs[min(sj[q] + c, N - 1) * dP + kq] - vi));
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = ns_wave_sum(acc[u]);
        if ((tid & 63) == 0) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (kk + u < kn) redk[tid >> 6][kk + u] = acc[u];
        }
      }
      __syncthreads();
      if (tid < kn) {
        double ssum = 0.0;
        for (int w = 0; w < nwaves; ++w) ssum += redk[w][tid];
        a.out_S[k0 + tid] = ssum;
      }
    }
    if (tid == 0) {
      a.out_S[d] = sd;
      a.out_S[d + 1] = tr;
      a.out_S[d + 2] = gg;
    }
  }
#ifdef NS_PROFILE
  if (tid == 0) a.out_scal[24] = (double)(clock64() - tloop_end);
#endif
  __threadfence_system();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// =====================================================================================================================
// 128 < N <= 252: the same in-place elimination on the matrix cores.  R comes from k_build_R (global memory), R^-1 and gamma
// go back to global memory for k_grad_contract: the pair work (~150 instructions an entry) belongs on all CUs, the
// factorisation chain on one.  16 x 16 tiles live in MFMA accumulators, <= 23 a wave, 6 owner waves; a step's update of a
// tile is ONE v_mfma_f64_16x16x4_f64, D -= P_I P_J^T, the two operands being 64 consecutive doubles of the panel each.
// D[i][j] sits in lane 16 (i % 4) + j, component i / 4: component v of lane l is element (l / 16, l % 4) of the 4 x 4
// sub-block (4 I + v, 4 J + (l % 16) / 4) -- the publish / reset events of the 4 x 4 scheme above are per-lane predicates.
// Wave 0: the panel wave (lane i = block row i, lane nb = the [y; 1] rows); wave 1: block row nb, one 4 x 4 block a lane.
// =====================================================================================================================
namespace {
constexpr int MD_PITCH = 17;    // doubles per panel block in LDS (16 + 1: two-way conflicts at worst for the per-block writers)
constexpr int MD_SLOTS = 23;    // tiles per owner wave: 16 * 17 / 2 = 136 tiles over 6 waves; 2 waves a SIMD = 256 registers a lane
constexpr int MD_OWNERS = 6;
constexpr int MD_WAVES = 8;       // wave w runs on SIMD w % 4: 0 = panel, 4 = block row nb (SIMD 0 to themselves); the other six own tiles
constexpr int MD_THREADS = 64 * MD_WAVES;
constexpr int MD_MAXNB = 63;    // N <= 252: block rows 0 .. nb fit the 64 lanes of the panel wave

typedef double md4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void md_mfma(double a, double b, md4& c) {
  asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
}  // namespace

// The three roles run their own copy of the phase loop (the same number of barriers in each): the 184 accumulator registers of
// an owner, the ~120 of the panel wave's factor + trsm and the block of wave 1 are then never live together.
//
// Software pipeline, ONE barrier a step: in phase p the owners apply update p (panel P_p) while the panel wave already builds
// P_{p+1}.  For that the blocks of column / row p + 1 were copied to `Raw` at the end of phase p - 1, in their state after update
// p - 1, and the panel wave applies update p to its copy itself (lane i: M -= P_p[i] P_p[p + 1]^T, P_p[i] being its own output
// of the phase before).  The owners' registers of those blocks restart from zero at the END of phase p (the update the MFMA
// just added to them belongs to the old role).  `P2` and `Raw` are double-buffered by the parity of the step they belong to.
template <bool GRAD>
__global__ __launch_bounds__(MD_THREADS) void k_spd_mid(const SpdMidArgs a) {
  __shared__ double P2[2][(MD_MAXNB + 1) * MD_PITCH];
  __shared__ double Raw[2][(MD_MAXNB + 1) * MD_PITCH];
  __shared__ double yt[4 * (MD_MAXNB + 1)], ft[4 * (MD_MAXNB + 1)], gy[4 * (MD_MAXNB + 1)], g1[4 * (MD_MAXNB + 1)];
  __shared__ double red[MD_THREADS / 64];
  __shared__ int tileTab[MD_OWNERS][MD_SLOTS];   // (I << 8) | J of an owner's slot, -1: none
  __shared__ int maskTab[MD_OWNERS][MD_SLOTS + 2];  // slots of an owner holding a tile of tile-column or tile-row K
  __shared__ double s_logdet;
  __shared__ int s_info;

  const int N = a.N, nb = (N + 3) / 4, NB16 = (N + 15) / 16, ntiles = NB16 * (NB16 + 1) / 2;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = tid & 63;
  const int nwaves = MD_THREADS / 64;
  for (int e = tid; e < 2 * (MD_MAXNB + 1) * MD_PITCH; e += MD_THREADS) {
    (&P2[0][0])[e] = 0.0;
    (&Raw[0][0])[e] = 0.0;
  }
  if (tid == 0) s_info = 0;
  if (tid < MD_THREADS / 64) red[tid] = 0.0;
  const bool is_owner = (wave & 3) != 0;
  const int ow = wave - 1 - (wave > 4 ? 1 : 0);  // 0 .. 5
  auto tile_of = [&](int t) {  // tile t, row-major over the lower triangle -> (I << 8) | J
    int I = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
    while ((I + 1) * (I + 2) / 2 <= t) ++I;
    while (I * (I + 1) / 2 > t) --I;
    return (I << 8) | (t - I * (I + 1) / 2);
  };
  if (is_owner) {
    if (lane < MD_SLOTS) {
      const int t = ow + MD_OWNERS * lane;
      tileTab[ow][lane] = t < ntiles ? tile_of(t) : -1;
    }
    if (lane < MD_SLOTS + 2) {  // K = lane
      int m = 0;
      for (int sl = 0; sl < MD_SLOTS; ++sl) {
        const int t = ow + MD_OWNERS * sl;
        if (t < ntiles) {
          const int ij = tile_of(t);
          if ((ij >> 8) == lane || (ij & 255) == lane) m |= 1 << sl;
        }
      }
      maskTab[ow][lane] = m;
    }
  }
  __syncthreads();  // (the zeroed LDS, the tables)
  // (SIMD 0 is left to the panel wave and the light wave 4: the factor chain sets the pace of a step)

  if (wave == 0) {
    // ================= the panel wave: lane i = block row i of the panel (lane nb: the [y; 1] rows) =================
    __builtin_amdgcn_s_setprio(3);
    double pivm = 1.0;  // prod(l_cc) = pivm 2^pive
    int pive = 0;
    const int i = min(lane, nb);
    double o[4][4];
#ifdef NS_PROFILE
    long long tq = clock64(), tpan = 0, twait = 0;
#endif
    __syncthreads();  // raw(0), raw(1) are published
    for (int kn = 0; kn < nb; ++kn) {  // P_kn: before the loop's first barrier for kn = 0, in phase kn - 1 otherwise
      double D[4][4], w[4][4], M[4][4];
      const double* rawb = Raw[kn & 1];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) M[r][c] = rawb[i * MD_PITCH + 4 * r + c];
      if (kn > 0) {  // update kn - 1 of the copy: M -= P[i] P[kn]^T
        const double* q = P2[(kn - 1) & 1] + kn * MD_PITCH;
        double Q[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) Q[r][c] = q[4 * r + c];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            double sacc = M[r][c];
#pragma unroll
            for (int m = 0; m < 4; ++m) sacc = __builtin_fma(-o[r][m], Q[c][m], sacc);
            M[r][c] = sacc;
          }
      }
      // the diagonal block is lane kn's
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) {
          const int lo = __builtin_amdgcn_readlane(__double2loint(M[r][c]), kn);
          const int hi = __builtin_amdgcn_readlane(__double2hiint(M[r][c]), kn);
          D[r][c] = __hiloint2double(hi, lo);
        }
      double prod4 = 1.0;
      const int bad = ns_factor4(D, w, prod4);
      if (lane == 0) {
        if (bad && s_info == 0) s_info = 4 * kn + bad;
        int e2;
        pivm = frexp(pivm * prod4, &e2);
        pive += e2;
      }
      if (i == kn) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) M[r][c] = r == c ? 1.0 : 0.0;
      }
      const bool used = GRAD || i > kn;  // (rows above the diagonal carry nothing without the inverse: keep them finite)
      double* pdst = P2[kn & 1] + lane * MD_PITCH;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double sacc = 0.0;
#pragma unroll
          for (int m = 0; m <= c; ++m) sacc = __builtin_fma(M[r][m], w[c][m], sacc);
          sacc = used ? sacc : 0.0;
          o[r][c] = sacc;
          if (lane <= nb) pdst[4 * r + c] = sacc;
          if (lane == nb && r == 0) yt[4 * kn + c] = sacc;
          if (lane == nb && r == 1) ft[4 * kn + c] = sacc;
        }
#ifdef NS_PROFILE
      { const long long t = clock64(); tpan += t - tq; tq = t; }
#endif
      __syncthreads();  // P_kn is published (and the raw blocks of step kn + 1)
#ifdef NS_PROFILE
      { const long long t = clock64(); twait += t - tq; tq = t; }
#endif
    }
    __syncthreads();  // (phase nb - 1: the owners' last update)
#ifdef NS_PROFILE
    if (lane == 0) { a.scal[21] = (double)twait; a.scal[22] = (double)tpan; }
#endif
    if (lane == 0) s_logdet = log(pivm) + (double)pive * 0.6931471805599453;
  } else if (wave == 4) {
    // ================= block row nb: block (nb, lane) of [y; 1], 4 x 4 on the vector ALU ==========================
    double T[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double v = 0.0;
        if (lane < nb && 4 * lane + c < N) v = r == 0 ? a.y[4 * lane + c] : (r == 1 ? 1.0 : 0.0);
        T[r][c] = v;
      }
    auto events = [&](int z, int q) {  // restart the block of column z; copy out the block of column q
      if (GRAD && lane == z) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) T[r][c] = 0.0;
      }
      if (lane == q && q < nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) Raw[q & 1][nb * MD_PITCH + 4 * r + c] = T[r][c];
      }
    };
    events(-1, 0);
    events(0, 1);
    __syncthreads();
    __syncthreads();
    for (int p = 0; p < nb; ++p) {
      if (lane < nb && (GRAD || lane > p)) {
        const double* pp = P2[p & 1];
        double pa[4][4], pb[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            pa[r][c] = pp[nb * MD_PITCH + 4 * r + c];
            pb[r][c] = pp[lane * MD_PITCH + 4 * r + c];
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
      events(p + 1, p + 2);
      __syncthreads();
    }
    if (GRAD && lane < nb) {  // -(R^-1 y), -(R^-1 1)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        gy[4 * lane + c] = T[0][c];
        g1[4 * lane + c] = T[1][c];
      }
    }
  } else {
    // ================= owners: tiles t = ow + 6 s -> (I, J), row-major over the lower triangle ======================
    const int lr = lane >> 4, lcb = (lane & 15) >> 2, lc = lane & 3;  // the lane's element (lr, lc) of sub-block column lcb
    int tIJ[MD_SLOTS];  // (I << 8) | J, -1: no tile
    md4 acc[MD_SLOTS];
#pragma unroll
    for (int s = 0; s < MD_SLOTS; ++s) {
      tIJ[s] = __builtin_amdgcn_readfirstlane(tileTab[ow][s]);
      acc[s] = (md4){0.0, 0.0, 0.0, 0.0};
      if (tIJ[s] >= 0) {
        const int tI = tIJ[s] >> 8, tJ = tIJ[s] & 255;
        const int col = 16 * tJ + (lane & 15);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * tI + 4 * v + lr;
          double x;
          if (row >= N || col >= N) x = row == col ? 1.0 : 0.0;  // identity padding
          else x = row >= col ? a.R[(size_t)col * a.ldr + row] : a.R[(size_t)row * a.ldr + col];
          acc[s][v] = -x;  // the accumulators hold the NEGATED state: acc += P_I P_J^T needs no operand negation
        }
      }
    }
    // Restart (GRAD) the blocks of column z and row z; copy out those of column q (as they are) and row q (transposed).  Only the
    // few slots of this wave with a tile in tile-column / tile-row z / 4 or q / 4 do anything (maskTab: one bit test a slot).
    auto events = [&](int z, int q) {
      const int Kz = max(z, 0) >> 2;
      const int hit = __builtin_amdgcn_readfirstlane(maskTab[ow][Kz] | maskTab[ow][Kz + 1]);
      if (hit == 0) return;
      double* rawb = Raw[q & 1];
      const int zK = z >> 2, zv = z & 3, qK = q >> 2, qv = q & 3;
      const bool lane_zc = lcb == zv, lane_qc = lcb == qv;
#pragma unroll
      for (int s = 0; s < MD_SLOTS; ++s) {
        if (!(hit & (1 << s))) continue;
        const int tI = tIJ[s] >> 8, tJ = tIJ[s] & 255;
        if (GRAD && z >= 0) {
          if (tJ == zK) {
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[s][v] = (lane_zc && 4 * tI + v >= z) ? 0.0 : acc[s][v];
          }
          if (tI == zK) {
            const bool zr = 4 * tJ + lcb < z;
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[s][v] = (zr && v == zv) ? 0.0 : acc[s][v];
          }
        }
        if (q < nb) {
          if (tJ == qK && lane_qc) {  // column q: sub-blocks (4 I + v, q), 4 I + v >= q
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int bi = 4 * tI + v;
              if (bi >= q && bi < nb) rawb[bi * MD_PITCH + 4 * lr + lc] = -acc[s][v];
            }
          }
          if (GRAD && tI == qK) {  // row q: sub-blocks (q, bj), bj < q, transposed
            const int bj = 4 * tJ + lcb;
            if (bj < q) {
              const double x = qv == 0 ? acc[s][0] : (qv == 1 ? acc[s][1] : (qv == 2 ? acc[s][2] : acc[s][3]));
              rawb[bj * MD_PITCH + 4 * lc + lr] = -x;
            }
          }
        }
      }
    };
    events(-1, 0);
    events(0, 1);
    __syncthreads();
    __syncthreads();
    const int offA = ((lane & 15) >> 2) * MD_PITCH + 4 * (lane & 3) + (lane >> 4);
    static_assert(MD_SLOTS == 23, "the hazard guards below name 23 accumulators");
#define MD_ALL_ACC "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),       \
                   "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]), \
                   "+v"(acc[16]), "+v"(acc[17]), "+v"(acc[18]), "+v"(acc[19]), "+v"(acc[20]), "+v"(acc[21]), "+v"(acc[22])
    for (int p = 0; p < nb; ++p) {
      const double* pp = P2[p & 1] + offA;
      // (VALU writes to the accumulators -- the restarts of events() -- must have retired before an MFMA reads them)
      asm volatile("s_nop 7\n\ts_nop 7" : MD_ALL_ACC);
      // operands one slot ahead of the MFMA that consumes them
      double pa[2], pb[2];
      pa[0] = pp[4 * max(tIJ[0] >> 8, 0) * MD_PITCH];
      pb[0] = pp[4 * (max(tIJ[0], 0) & 255) * MD_PITCH];
#pragma unroll
      for (int s = 0; s < MD_SLOTS; ++s) {
        if (s + 1 < MD_SLOTS) {
          pa[(s + 1) & 1] = pp[4 * max(tIJ[s + 1] >> 8, 0) * MD_PITCH];
          pb[(s + 1) & 1] = pp[4 * (max(tIJ[s + 1], 0) & 255) * MD_PITCH];
        }
        if (tIJ[s] >= 0 && (GRAD || 4 * (tIJ[s] & 255) + 3 > p)) md_mfma(pa[s & 1], pb[s & 1], acc[s]);
      }
      // the drain names the accumulators as in/out operands so that no read of them can be scheduled above it
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : MD_ALL_ACC);
      events(p + 1, p + 2);
      __syncthreads();
    }
#undef MD_ALL_ACC
    if (GRAD) {  // R^-1 = -S, lower triangle, column-major
#pragma unroll
      for (int s = 0; s < MD_SLOTS; ++s) {
        if (tIJ[s] < 0) continue;
        const int tI = tIJ[s] >> 8, tJ = tIJ[s] & 255;
        const int col = 16 * tJ + (lane & 15);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * tI + 4 * v + lr;
          if (row < N && col <= row) a.Rinv[(size_t)col * a.ldi + row] = acc[s][v];  // (negated state: -S = R^-1)
        }
      }
    }
  }
  __syncthreads();

  // ---- the likelihood's scalars (k_fit_rho's expressions) --------------------------------------------------------
  double sff = 0.0, sfy = 0.0;
  if (tid < N) {
    const double f = ft[tid];
    sff = f * f;
    sfy = f * yt[tid];
  }
  sff = ns_block_sum(sff, red, nwaves);
  sfy = ns_block_sum(sfy, red, nwaves);
  const double nrm = sqrt(sff);
  double coef;
  if (a.estimate_trend) {
    const double G = -nrm, qty = sfy / G;
    coef = -(qty / G);
  } else {
    coef = -a.beta;
  }
  double srr = 0.0;
  if (tid < N) {
    const double rr = __builtin_fma(coef, ft[tid], yt[tid]);
    srr = rr * rr;
  }
  srr = ns_block_sum(srr, red, nwaves);
  if (tid == 0) {
    a.scal[0] = s_logdet;
    a.scal[1] = nrm;
    a.scal[2] = sfy;
    a.scal[3] = srr;
    double iw = 0.0;
    int info = s_info;
    memcpy(&iw, &info, sizeof(info));
    a.scal[62] = iw;
    if (GRAD) {  // k_grad_coef's weights for one target
      const double s2t = a.mode == BOGP_MODE_NOISY ? a.s2t_host : (a.mode == BOGP_MODE_NOISELESS ? srr / (N - (a.estimate_trend ? 1 : 0)) : srr / N);
      a.coef[0] = 1.0 / s2t;
      a.coef[8] = 1.0 / s2t;
    }
  }
  if (GRAD && tid < N) a.gamma[tid] = -__builtin_fma(coef, g1[tid], gy[tid]);  // gamma = R^-1 y - beta R^-1 1
}

int spd_mid_max_n() { return 4 * MD_MAXNB; }
hipError_t launch_spd_mid(bool grad, const SpdMidArgs& a, hipStream_t st) {
  if (grad) hipLaunchKernelGGL((k_spd_mid<true>), dim3(1), MD_THREADS, 0, st, a);
  else hipLaunchKernelGGL((k_spd_mid<false>), dim3(1), MD_THREADS, 0, st, a);
  return hipGetLastError();
}

size_t nll_small_lds_bytes(int N, int d);
int nll_small_max_n() { return NS_BS * NS_MAXNB; }
// one workgroup: <= 1024 threads, <= 160 KB of LDS (~36 KB static + X and the image of the blocks)
bool nll_small_fits(int N, int d) {
  if (N > NS_BS * NS_MAXNB || d > 64) return false;
  const int nb = (N + NS_BS - 1) / NS_BS;
  const int block = ((4 * (nb + 1) + 63) / 64) * 64 + (((nb + 1) * (nb + 2) / 2 - 1 + 63) / 64) * 64;
  return block <= NS_THREADS && nll_small_lds_bytes(N, d) + 36 * 1024 <= 160 * 1024;
}

size_t nll_small_lds_bytes(int N, int d) {
  const int nb = (N + NS_BS - 1) / NS_BS;
  return ((((size_t)N * (d | 1) + 1) & ~(size_t)1) + (size_t)16 * (nb * (nb + 1) / 2)) * sizeof(double);
}

template <typename K>
static hipError_t ns_launch(K kern, int block, size_t lds, const NllSmallArgs& a, hipStream_t st) {
  if (lds > 24576) {  // (static + dynamic above the default 64 KB: raise it for this kernel; once would do, the call is cheap)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(1), block, lds, st, a);
  return hipGetLastError();
}

hipError_t launch_nll_small(int kernel, bool grad, const NllSmallArgs& a, hipStream_t st) {
  const int nb = (a.N + NS_BS - 1) / NS_BS;
  const int nown = (nb + 1) * (nb + 2) / 2 - 1;
  const int block = ((4 * (nb + 1) + 63) / 64) * 64 + ((nown + 63) / 64) * 64;  // the panel waves + the owners
  const size_t lds = nll_small_lds_bytes(a.N, a.d);
#define NS_GO(K, G) (block <= NS_THREADS_128 ? ns_launch(k_nll_small<K, G, NS_THREADS_128>, block, lds, a, st) : ns_launch(k_nll_small<K, G, NS_THREADS>, block, lds, a, st))
  if (block > NS_THREADS || lds + 36 * 1024 > 160 * 1024) return hipErrorInvalidValue;  // (nll_small_fits() said otherwise)
  if (grad) {
    switch (kernel) {
      case BOGP_KERNEL_SE: return NS_GO(BOGP_KERNEL_SE, true);
      case BOGP_KERNEL_MATERN12: return NS_GO(BOGP_KERNEL_MATERN12, true);
      case BOGP_KERNEL_MATERN32: return NS_GO(BOGP_KERNEL_MATERN32, true);
      case BOGP_KERNEL_ABSEXP: return NS_GO(BOGP_KERNEL_ABSEXP, true);
      case BOGP_KERNEL_MATERN52: return NS_GO(BOGP_KERNEL_MATERN52, true);
      default: return hipErrorInvalidValue;  // cubic / generalized_exponential have no theta-derivative
    }
  } else {
    switch (kernel) {
      case BOGP_KERNEL_SE: return NS_GO(BOGP_KERNEL_SE, false);
      case BOGP_KERNEL_MATERN12: return NS_GO(BOGP_KERNEL_MATERN12, false);
      case BOGP_KERNEL_MATERN32: return NS_GO(BOGP_KERNEL_MATERN32, false);
      case BOGP_KERNEL_ABSEXP: return NS_GO(BOGP_KERNEL_ABSEXP, false);
      case BOGP_KERNEL_MATERN52: return NS_GO(BOGP_KERNEL_MATERN52, false);
      case BOGP_KERNEL_CUBIC: return NS_GO(BOGP_KERNEL_CUBIC, false);
      case BOGP_KERNEL_GENEXP: return NS_GO(BOGP_KERNEL_GENEXP, false);
      default: return hipErrorInvalidValue;
    }
  }
  return hipGetLastError();
}
#undef NS_GO

}  // namespace bogp
