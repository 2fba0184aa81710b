// kernels_nllsmall.hip -- the WHOLE likelihood evaluation of a small training set (N <= 156, constant trend, one target) in ONE
// launch of ONE workgroup (gfx950).  (Above that, up to N = 2048, the same elimination runs at 64-block granularity with one
// workgroup a block: kernels_chol.hip, k_elim_*.)
//
// Why: at the sizes of an ordinary BO run an evaluation on the general path (bogp_api.hip: factorize + the gradient tail) is a
// chain of ~18 launches, whatever their arithmetic; a BO loop is 98 % such evaluations
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
// so the N^3 / 2 FMAs of factor + inverse + product are spread evenly over all threads and all steps (one barrier a step: the
// panel is built by separate threads one step ahead, see k_nll_small).  Block row nb continued the same way ends as
// -(R^-1 y)^T, -(R^-1 1)^T: gamma, by the reference's cho_solve route.
#include <atomic>

#include "bogp_device.h"
#include "bogp_internal.h"

namespace bogp {

namespace {

constexpr int NS_BS = 4;           // block size
constexpr int NS_MAXNB = 39;       // N <= 156 (the 1024-thread instantiation; N <= 128 runs the 768-thread one)
constexpr int NS_PITCH = NS_MAXNB + 2;
__device__ __forceinline__ int ns_pidx(int e, int i) { return (e >> 1) * (2 * NS_PITCH) + 2 * i + (e & 1); }
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
//
// BATCH (bogp_nll_batch): grid = P workgroups, one parameter vector each.  Slot s = blockIdx.x takes what differs between the
// evaluations -- theta, the exponent, k_build_R's (a, b, diag, div) and the noisy mode's total variance -- from the block
// a.bpar + s * NS_BPAR (device-mapped pinned memory the host filled), leaves its 64 scalars and d + 3 sums in a.bout + s * a.bout_stride,
// and the LAST workgroup to finish (a ticket) publishes the sequence word.  The arithmetic is the one-evaluation kernel's own
// (same expressions on the same values): slot s returns the bits of the s-th sequential call.
template <int KERNEL, bool GRAD, int TMAX, bool BATCH>
__global__ __launch_bounds__(TMAX) void k_nll_small(const NllSmallArgs a) {
  extern __shared__ double dyn[];      // Xs[N][dP] | Rst[16][nbR]: the blocks of R, later of R^-1, element-major
  // P[step & 1][ns_pidx(e, i)]: element e = 4 r + c of block row i's panel block, elements 2 q and 2 q + 1 next to each other so
  // that an owner fetches its operands with 16 ds_read_b128 instead of 32 ds_read_b64 (consecutive owners: consecutive 16-byte words)
  __shared__ __attribute__((aligned(16))) double P[2][16 * NS_PITCH];
  __shared__ double Raw[2][16 * NS_PITCH];
  __shared__ double yt[NS_BS * NS_MAXNB], ft[NS_BS * NS_MAXNB], gam[NS_BS * NS_MAXNB];
  __shared__ double red[NS_WAVES];
  __shared__ double redk[NS_WAVES][65];
  __shared__ unsigned short blkmap[NS_MAXNB * (NS_MAXNB + 1) / 2];
  __shared__ double s_logdet;
  __shared__ int s_info;
  __shared__ double s_par[BATCH ? NS_BPAR : 1];

#ifdef NS_PROFILE
  const long long tstart = clock64();
#endif
  const int N = a.N, d = a.d, nb = (N + NS_BS - 1) / NS_BS;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int slot = BATCH ? (int)blockIdx.x : 0;
  if (BATCH && tid < NS_BPAR) s_par[tid] = a.bpar[(size_t)slot * NS_BPAR + tid];
  double* const out_scal = BATCH ? a.bout + (size_t)slot * a.bout_stride : a.out_scal;
  double* const out_S = BATCH ? out_scal + 64 : a.out_S;
#define NS_TH(k_) (BATCH ? s_par[k_] : a.theta[k_])
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
  const double par_pexp = BATCH ? s_par[64] : a.pexp, par_a = BATCH ? s_par[65] : a.a, par_b = BATCH ? s_par[66] : a.b;
  const double par_diag = BATCH ? s_par[67] : a.diag, par_s2t = BATCH ? s_par[68] : a.s2t_host;
  const int par_div = BATCH ? (s_par[69] != 0.0 ? 1 : 0) : a.div;

  // ---- R, strip by strip, into the LDS image ---------------------------------------------------------------------
  for (int s = tid; s < 4 * nbR; s += nthr) {
    const int t = s >> 2, r = s & 3;
    const int sbi = blkmap[t] >> 8, sbj = blkmap[t] & 255;
    const int i = 4 * sbi + r;
    double s2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) s2[c] = dist_init<KERNEL>();
    if (i < N) {
      const double pexp = par_pexp;
      const double* xi = Xs + i * dP;
      const double* xj = Xs + min(4 * sbj, N - 1) * dP;  // (columns j <= i < N; the clamp only keeps padding reads in range)
      for (int k = 0; k < d; ++k) {
        const double th = NS_TH(k), vi = xi[k];
#pragma unroll
        for (int c = 0; c < 4; ++c) s2[c] = dist_fold<KERNEL>(th, vi - xj[min(c, N - 1 - min(4 * sbj, N - 1)) * dP + k], s2[c], pexp);
      }
    }
    // (the four profiles evaluated unconditionally, side by side: a lone wave waits ~26 cycles for a dependent FP64 result, and a
    // branch per entry would string the four exp / sqrt chains one behind the other)
    double pv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) pv[c] = par_a * corr_profile<KERNEL>(s2[c], par_pexp);
    if (par_div) {
#pragma unroll
      for (int c = 0; c < 4; ++c) pv[c] = pv[c] / par_b;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = 4 * sbj + c;
      double v = pv[c];
      if (i >= N || j >= N) v = i == j ? 1.0 : 0.0;  // identity padding
      else if (i == j) v = par_diag;
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
  if (tid == 0) out_scal[20] = (double)(tp0 - tstart);
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
          const double* q = P[(kn - 1) & 1];
          double Q[4][4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < 4; m += 2) {
              const double2 v = *reinterpret_cast<const double2*>(q + ns_pidx(4 * r + m, kn));
              Q[r][m] = v.x;
              Q[r][m + 1] = v.y;
            }
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
          if ((tid >> 2) <= nb) pdst[ns_pidx(4 * pr + c, i)] = v;
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
          for (int c = 0; c < 4; c += 2) {
            const double2 va = *reinterpret_cast<const double2*>(pp + ns_pidx(4 * r + c, bi));
            const double2 vb = *reinterpret_cast<const double2*>(pp + ns_pidx(4 * r + c, bj));
            pa[r][c] = va.x;
            pa[r][c + 1] = va.y;
            pb[r][c] = vb.x;
            pb[r][c + 1] = vb.y;
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
  if (tid == 0) { out_scal[21] = (double)tpA; out_scal[22] = (double)tpB; out_scal[23] = (double)tpC; }
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
    out_scal[0] = s_logdet;
    out_scal[1] = nrm;
    out_scal[2] = sfy;
    out_scal[3] = srr;
    double iw = 0.0;
    int info = s_info;
    memcpy(&iw, &info, sizeof(info));
    out_scal[62] = iw;
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
    const double s2t = a.mode == BOGP_MODE_NOISY ? par_s2t : (a.mode == BOGP_MODE_NOISELESS ? srr / (N - (a.estimate_trend ? 1 : 0)) : srr / N);
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
            const double th = NS_TH(k), vi = xi[k];
#pragma unroll
            for (int c = 0; c < 4; ++c) s2[c] += dist_term<KERNEL>(th, Xs[min(4 * sbj + c, N - 1) * dP + k] - vi);
          }
          const double gi = gam[i];
          double r0[4], hh[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) corr_pair<KERNEL>(s2[c], r0[c], hh[c]);  // (side by side, see the prologue)
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
        out_S[k0 + tid] = ssum;
      }
    }
    if (tid == 0) {
      out_S[d] = sd;
      out_S[d + 1] = tr;
      out_S[d + 2] = gg;
    }
  }
#ifdef NS_PROFILE
  if (tid == 0) out_scal[24] = (double)(clock64() - tloop_end);
#endif
#undef NS_TH
  __threadfence_system();
  __syncthreads();
  if (tid == 0) {
    bool publish = true;
    if (BATCH) {  // (every slot's record was fenced to the system before its ticket: the last one may publish for all)
      publish = atomicAdd(a.bticket, 1u) == (unsigned int)(a.P - 1);
      if (publish) {
        *a.bticket = 0u;
        __threadfence_system();
      }
    }
    if (publish) __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
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

template <auto KERN>
static hipError_t ns_launch(int block, size_t lds, const NllSmallArgs& a, hipStream_t st) {
  // static + dynamic LDS above the default 64 KB: the limit of this instantiation is raised, once per size reached (the largest
  // request so far is remembered per instantiation and per device)
  static std::atomic<size_t> raised[16];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<size_t>& r = raised[dev & 15];
  if (lds > 24576 && lds > r.load(std::memory_order_relaxed)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    r.store(lds, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(KERN, dim3(a.bpar ? a.P : 1), block, lds, st, a);
  return hipGetLastError();
}

hipError_t launch_nll_small(int kernel, bool grad, const NllSmallArgs& a, hipStream_t st) {
  const int nb = (a.N + NS_BS - 1) / NS_BS;
  const int nown = (nb + 1) * (nb + 2) / 2 - 1;
  const int block = ((4 * (nb + 1) + 63) / 64) * 64 + ((nown + 63) / 64) * 64;  // the panel waves + the owners
  const size_t lds = nll_small_lds_bytes(a.N, a.d);
#define NS_GO(K, G)                                                                                                               \
  (a.bpar ? (block <= NS_THREADS_128 ? ns_launch<k_nll_small<K, G, NS_THREADS_128, true>>(block, lds, a, st)                        \
                                     : ns_launch<k_nll_small<K, G, NS_THREADS, true>>(block, lds, a, st))                           \
          : (block <= NS_THREADS_128 ? ns_launch<k_nll_small<K, G, NS_THREADS_128, false>>(block, lds, a, st)                       \
                                     : ns_launch<k_nll_small<K, G, NS_THREADS, false>>(block, lds, a, st)))
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
      case BOGP_KERNEL_MATERN_NU: return NS_GO(BOGP_KERNEL_MATERN_NU, false);
      default: return hipErrorInvalidValue;
    }
  }
  return hipGetLastError();
}
#undef NS_GO

}  // namespace bogp
