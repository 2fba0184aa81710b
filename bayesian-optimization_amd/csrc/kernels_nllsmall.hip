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
constexpr int NS_MAXNB = 32;       // N <= 128
constexpr int NS_PITCH = NS_MAXNB + 2;
constexpr int NS_THREADS = 576;    // (nb + 1)(nb + 2) / 2 = 561 at nb = 32
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
// reciprocal square root estimate + two coupled Goldschmidt steps -- a dependent chain of 8 operations where sqrt() followed by a
// division is 45, and that chain is on the critical path of EVERY step
__device__ __forceinline__ void ns_sqrt_rsqrt(double p, double& root, double& inv) {
  const double y = __builtin_amdgcn_rsq(p);
  double g = p * y, hh = 0.5 * y;
  double r = __builtin_fma(-g, hh, 0.5);
  g = __builtin_fma(g, r, g);
  hh = __builtin_fma(hh, r, hh);
  r = __builtin_fma(-g, hh, 0.5);
  g = __builtin_fma(g, r, g);
  hh = __builtin_fma(hh, r, hh);
  // one correction of the root against p itself
  const double e = __builtin_fma(-g, g, p);
  root = __builtin_fma(e, hh, g);
  inv = hh + hh;
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

}  // namespace

// out_scal: [0] sum(log diag L), [1] |Ft|, [2] Ft.Yt, [3] rho.rho, [62] the info word (int); out_S: the d + 1 contractions,
// [d + 1] trace(R^-1), [d + 2] gamma.gamma  (the layout the general path's read-back has, bogp_api.hip: fit_readback)
template <int KERNEL, bool GRAD>
__global__ __launch_bounds__(NS_THREADS) void k_nll_small(const NllSmallArgs a) {
  __shared__ double P[16 * NS_PITCH];  // P[e * NS_PITCH + i]: element e = 4 r + c of block row i's panel block
  __shared__ double Wb[2][16];
  __shared__ double yt[NS_BS * NS_MAXNB], ft[NS_BS * NS_MAXNB], gam[NS_BS * NS_MAXNB];
  __shared__ double red[NS_WAVES];
  __shared__ double logpart[NS_MAXNB];
  __shared__ double redk[NS_WAVES][65];
  __shared__ int s_info;

  const int N = a.N, d = a.d, nb = (N + NS_BS - 1) / NS_BS;
  const int tid = threadIdx.x;
  const int nthreads = (nb + 1) * (nb + 2) / 2 - 1;  // block (nb, nb) does not exist
  const int nwaves = blockDim.x >> 6;
  // thread -> block (bi, bj), bi >= bj, row-major over the triangle
  int bi = (int)((sqrt(8.0 * tid + 1.0) - 1.0) * 0.5);
  while ((bi + 1) * (bi + 2) / 2 <= tid) ++bi;
  while (bi * (bi + 1) / 2 > tid) --bi;
  const int bj = tid - bi * (bi + 1) / 2;
  const bool live = tid < nthreads;
  const bool border = bi == nb;
  if (tid == 0) s_info = 0;

  // ---- the thread's block of [R; y; 1] ---------------------------------------------------------------------------
  double T[4][4];
  {
    double s2[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) s2[r][c] = dist_init<KERNEL>();
    if (live && !border) {
      const double pexp = a.pexp;
      for (int k = 0; k < d; ++k) {
        const double th = a.theta[k];
        double vi[4], vj[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) vi[r] = (4 * bi + r < N) ? a.X[(size_t)(4 * bi + r) * d + k] : 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) vj[c] = (4 * bj + c < N) ? a.X[(size_t)(4 * bj + c) * d + k] : 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) s2[r][c] = dist_fold<KERNEL>(th, vi[r] - vj[c], s2[r][c], pexp);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = 4 * bi + r, j = 4 * bj + c;
        double v = 0.0;
        if (!live) {
          v = 0.0;
        } else if (border) {
          if (j < N) v = r == 0 ? a.y[j] : (r == 1 ? 1.0 : 0.0);
        } else if (i >= N || j >= N) {
          v = i == j ? 1.0 : 0.0;  // identity padding
        } else if (i == j) {
          v = a.diag;
        } else if (a.div) {
          v = (a.a * corr_profile<KERNEL>(s2[r][c])) / a.b;
        } else {
          v = a.a * corr_profile<KERNEL>(s2[r][c]);
        }
        T[r][c] = v;
      }
  }
  double pivprod = 1.0;
  if (tid == 0) {  // block (0, 0): the first diagonal factor
    double w[4][4];
    const int bad = ns_factor4(T, w, pivprod);
    if (bad) s_info = bad;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) Wb[0][4 * r + c] = w[r][c];
  }

  // ---- nb steps of factor / invert / multiply, in place ----------------------------------------------------------
  for (int k = 0; k < nb; ++k) {
    __syncthreads();  // W of step k is published; every read of the previous panel is done
    if (live) {
      const bool is_col = bj == k && bi > k, is_row = bi == k && bj < k, is_diag = bi == k && bj == k;
      if (is_col || (GRAD && (is_row || is_diag))) {
        double w[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) w[r][c] = Wb[k & 1][4 * r + c];
        // one expression for the three publishers, o = M W^T:  L(bi, k) = T W^T;  X(bj, k) = (W Z)^T = Z^T W^T with Z = this
        // thread's X(bj, k)^T;  X(k, k) = I W^T
        double M[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) M[r][c] = is_col ? T[r][c] : (is_row ? T[c][r] : (r == c ? 1.0 : 0.0));
        double o[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            double s = 0.0;
#pragma unroll
            for (int m = 0; m <= c; ++m) s = __builtin_fma(M[r][m], w[c][m], s);
            o[r][c] = s;
          }
        if (is_col && border) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            yt[4 * k + c] = o[0][c];
            ft[4 * k + c] = o[1][c];
          }
        }
        const int slot = is_col ? bi : bj;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) P[(4 * r + c) * NS_PITCH + slot] = o[r][c];
        if (GRAD) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) T[r][c] = 0.0;
        }
      }
    }
    __syncthreads();  // the panel of step k is published
    if (live && (GRAD || bj > k)) {
      double pa[4][4], pb[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          pa[r][c] = P[(4 * r + c) * NS_PITCH + bi];
          pb[r][c] = P[(4 * r + c) * NS_PITCH + bj];
        }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double s = T[r][c];
#pragma unroll
          for (int m = 0; m < 4; ++m) s = __builtin_fma(-pa[r][m], pb[c][m], s);
          T[r][c] = s;
        }
      if (bi == k + 1 && bj == k + 1 && k + 1 < nb) {  // look-ahead: the next diagonal factor, behind this thread's update
        double w[4][4];
        const int bad = ns_factor4(T, w, pivprod);
        if (bad && s_info == 0) s_info = 4 * (k + 1) + bad;  // (only diagonal threads write, in step order)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) Wb[(k + 1) & 1][4 * r + c] = w[r][c];
      }
    }
  }
  if (live && bi == bj && bi < nb) logpart[bi] = log(pivprod);
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
    const double r = __builtin_fma(coef, ft[tid], yt[tid]);
    srr = r * r;
  }
  srr = ns_block_sum(srr, red, nwaves);
  if (tid == 0) {
    double ld = 0.0;
    for (int b = 0; b < nb; ++b) ld += logpart[b];
    a.out_scal[0] = ld;
    a.out_scal[1] = nrm;
    a.out_scal[2] = sfy;
    a.out_scal[3] = srr;
    double iw = 0.0;
    int info = s_info;
    memcpy(&iw, &info, sizeof(info));
    a.out_scal[62] = iw;
  }

  if (GRAD) {
    // gamma = R^-1 y - beta R^-1 1 from block row nb
    if (live && border) {
#pragma unroll
      for (int c = 0; c < 4; ++c) gam[4 * bj + c] = -__builtin_fma(coef, T[1][c], T[0][c]);
    }
    __syncthreads();
    const double s2t = a.mode == BOGP_MODE_NOISY ? a.s2t_host : (a.mode == BOGP_MODE_NOISELESS ? srr / (N - (a.estimate_trend ? 1 : 0)) : srr / N);
    const double cw = 1.0 / s2t;
    // the thread's pairs i > j (row 4 bi + r, column 4 bj + c) of the strict lower triangle: A = cw gamma_i gamma_j - Rinv_ij
    double B[4][4];
    double sd = 0.0, tr = 0.0;
    const bool pairs = live && !border;
    {
      double s2[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) s2[r][c] = 0.0;
      if (pairs) {
        for (int k = 0; k < d; ++k) {
          const double th = a.theta[k];
          double vi[4], vj[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) vi[r] = (4 * bi + r < N) ? a.X[(size_t)(4 * bi + r) * d + k] : 0.0;
#pragma unroll
          for (int c = 0; c < 4; ++c) vj[c] = (4 * bj + c < N) ? a.X[(size_t)(4 * bj + c) * d + k] : 0.0;
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) s2[r][c] += dist_term<KERNEL>(th, vj[c] - vi[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int i = 4 * bi + r, j = 4 * bj + c;
          double bb = 0.0;
          if (pairs && i > j && i < N) {
            const double r0 = corr_profile<KERNEL>(s2[r][c]);
            const double h = corr_dtheta_profile<KERNEL>(s2[r][c], r0);
            const double rinv = -T[r][c];
            const double A = __builtin_fma(gam[j] * gam[i], cw, 0.0) - rinv;
            bb = A * h;
            sd += A * r0;
          }
          if (pairs && i == j && i < N) tr += -T[r][c];
          B[r][c] = bb;
        }
    }
    sd = ns_block_sum(sd, red, nwaves);
    tr = ns_block_sum(tr, red, nwaves);
    double gg = tid < N ? gam[tid] * gam[tid] : 0.0;
    gg = ns_block_sum(gg, red, nwaves);
    for (int k0 = 0; k0 < d; k0 += 64) {
      const int kn = min(64, d - k0);
      __syncthreads();
      for (int kk = 0; kk < kn; ++kk) {
        double acc = 0.0;
        if (pairs) {
          double vi[4], vj[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) vi[r] = (4 * bi + r < N) ? a.X[(size_t)(4 * bi + r) * d + k0 + kk] : 0.0;
#pragma unroll
          for (int c = 0; c < 4; ++c) vj[c] = (4 * bj + c < N) ? a.X[(size_t)(4 * bj + c) * d + k0 + kk] : 0.0;
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc += B[r][c] * (-dtheta_weight<KERNEL>(vj[c] - vi[r]));
        }
        acc = ns_wave_sum(acc);
        if ((tid & 63) == 0) redk[tid >> 6][kk] = acc;
      }
      __syncthreads();
      if (tid < kn) {
        double s = 0.0;
        for (int w = 0; w < nwaves; ++w) s += redk[w][tid];
        a.out_S[k0 + tid] = s;
      }
    }
    if (tid == 0) {
      a.out_S[d] = sd;
      a.out_S[d + 1] = tr;
      a.out_S[d + 2] = gg;
    }
  }
  __threadfence_system();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int nll_small_max_n() { return NS_BS * NS_MAXNB; }

hipError_t launch_nll_small(int kernel, bool grad, const NllSmallArgs& a, hipStream_t st) {
  const int nb = (a.N + NS_BS - 1) / NS_BS;
  const int nthreads = (nb + 1) * (nb + 2) / 2 - 1;
  const int block = ((nthreads + 63) / 64) * 64;
  if (grad) {
    switch (kernel) {
      case BOGP_KERNEL_SE: hipLaunchKernelGGL((k_nll_small<BOGP_KERNEL_SE, true>), dim3(1), block, 0, st, a); break;
      case BOGP_KERNEL_MATERN12: hipLaunchKernelGGL((k_nll_small<BOGP_KERNEL_MATERN12, true>), dim3(1), block, 0, st, a); break;
      case BOGP_KERNEL_MATERN32: hipLaunchKernelGGL((k_nll_small<BOGP_KERNEL_MATERN32, true>), dim3(1), block, 0, st, a); break;
      case BOGP_KERNEL_ABSEXP: hipLaunchKernelGGL((k_nll_small<BOGP_KERNEL_ABSEXP, true>), dim3(1), block, 0, st, a); break;
      case BOGP_KERNEL_MATERN52: hipLaunchKernelGGL((k_nll_small<BOGP_KERNEL_MATERN52, true>), dim3(1), block, 0, st, a); break;
      default: return hipErrorInvalidValue;  // cubic / generalized_exponential have no theta-derivative
    }
  } else {
    switch (kernel) {
      case BOGP_KERNEL_SE: hipLaunchKernelGGL((k_nll_small<BOGP_KERNEL_SE, false>), dim3(1), block, 0, st, a); break;
      case BOGP_KERNEL_MATERN12: hipLaunchKernelGGL((k_nll_small<BOGP_KERNEL_MATERN12, false>), dim3(1), block, 0, st, a); break;
      case BOGP_KERNEL_MATERN32: hipLaunchKernelGGL((k_nll_small<BOGP_KERNEL_MATERN32, false>), dim3(1), block, 0, st, a); break;
      case BOGP_KERNEL_ABSEXP: hipLaunchKernelGGL((k_nll_small<BOGP_KERNEL_ABSEXP, false>), dim3(1), block, 0, st, a); break;
      case BOGP_KERNEL_MATERN52: hipLaunchKernelGGL((k_nll_small<BOGP_KERNEL_MATERN52, false>), dim3(1), block, 0, st, a); break;
      case BOGP_KERNEL_CUBIC: hipLaunchKernelGGL((k_nll_small<BOGP_KERNEL_CUBIC, false>), dim3(1), block, 0, st, a); break;
      case BOGP_KERNEL_GENEXP: hipLaunchKernelGGL((k_nll_small<BOGP_KERNEL_GENEXP, false>), dim3(1), block, 0, st, a); break;
      default: return hipErrorInvalidValue;
    }
  }
  return hipGetLastError();
}

}  // namespace bogp
