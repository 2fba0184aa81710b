// kernels_pairs.hip -- the two N^2 d pair kernels of the likelihood path of libbogp (gfx950):
//
//   k_build_R        correlation_matrix (gpr.py:772-782) + the per-mode normalisation (:949-969), without the
//                    N(N-1)/2 x d pair list of l1_cross_distances (:48-61)
//   k_grad_contract  the d + 1 trace-contractions of the likelihood gradient (gpr.py:994-1038) formed on the fly from
//                    X:  sum_{i<j} A_ij dR0_ij/dtheta_k  and  sum_{i<j} A_ij R0_ij  with
//                    A_ij = gamma_i gamma_j c1 + q_i q_j c2 - Rinv_ij  -- no (N, N, d) tensor (:736-770)
//
// Both walk 64 x 64 tiles of point pairs with 256 threads (4 x 4 pairs each) and stage the coordinates of the two point
// blocks through LDS in chunks of 16 dimensions (k-major, pitch 65: conflict-free stores and reads).  The first
// versions gave one THREAD one pair and re-read 2 d coordinates from global memory per pair; at N = 8192, d = 50 they
// took 2.5 ms and 5.2 ms of a 29-ms likelihood + gradient evaluation (profiles/r01_nll_n8192_kernel_stats.csv).
#include <cstdlib>

#include "bogp_device.h"
#include "bogp_internal.h"

namespace bogp {

namespace {

constexpr int PT = 64;        // points per tile side
constexpr int KC = 16;        // dimensions per staged chunk
constexpr int PP = PT + 1;    // LDS pitch (doubles)

// coordinates of points p0 .. p0 + 63, dimensions kc .. kc + 15 -> dst[kk][p]; zero beyond N / d
__device__ __forceinline__ void stage_points(double* dst, const double* __restrict__ X, int N, int d, int p0, int kc, int tid) {
#pragma unroll
  for (int it = 0; it < (PT * KC) / 256; ++it) {
    const int idx = tid + 256 * it;
    const int p = idx / KC, kk = idx % KC;
    const int gp = p0 + p, gk = kc + kk;
    dst[kk * PP + p] = (gp < N && gk < d) ? X[(size_t)gp * d + gk] : 0.0;
  }
}

// the same for a tile side of 16 Q points (Q = 4: stage_points)
// PERM: point p of the tile goes to slot (p % Q) 16 + p / Q, so that the 16 lanes tx of a row of threads, which own the points Q tx + c, read
// consecutive words for a given c (in point order their reads are Q doubles apart: 4-way bank conflicts at Q = 4, ~7 % of k_grad_contract's time,
// profiles/r04_elim_batch_pmc.txt)
template <int Q, bool PERM = false>
__device__ __forceinline__ void stage_points_q(double* dst, const double* __restrict__ X, int N, int d, int p0, int kc, int tid) {
  constexpr int PTQ = 16 * Q, PPQ = PTQ + 1;
#pragma unroll
  for (int it = 0; it < (PTQ * KC) / 256; ++it) {
    const int idx = tid + 256 * it;
    const int p = idx / KC, kk = idx % KC;
    const int gp = p0 + p, gk = kc + kk;
    dst[kk * PPQ + (PERM ? (p % Q) * 16 + p / Q : p)] = (gp < N && gk < d) ? X[(size_t)gp * d + gk] : 0.0;
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += shfl_xor_f64(v, o);
  return v;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// R (lower 64 x 64 tiles, diagonal tiles complete and exactly symmetric) into a column-major matrix.
//   DIV = false:  off-diagonal = a * corr          (NOISELESS a = 1; NOISE_ESTIM a = alpha, gpr.py:951)
//   DIV = true:   off-diagonal = (a * corr) / b    (NOISY: C = sigma2 R0 + tau2 I; R = C / sigma2_total, :966-967)
// ---------------------------------------------------------------------------------------------------------------
template <int KERNEL, bool DIV>
__device__ __forceinline__ void build_R_tile(const double* __restrict__ X, int N, int d, const double* __restrict__ theta,
                                             double a, double b, double diag, double* __restrict__ R, int ld) {
  __shared__ double xi[KC * PP], xj[KC * PP];
  const int bi = blockIdx.y, bj = blockIdx.x;  // row tile, column tile
  if (bj > bi) return;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int i0 = bi * PT, j0 = bj * PT;
  double s2[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) s2[r][c] = dist_init<KERNEL>();
  const double pexp = kernel_exponent<KERNEL>(theta, d);
  for (int kc = 0; kc < d; kc += KC) {
    __syncthreads();
    stage_points(xi, X, N, d, i0, kc, tid);
    stage_points(xj, X, N, d, j0, kc, tid);
    __syncthreads();
    const int kn = min(KC, d - kc);
    for (int kk = 0; kk < kn; ++kk) {
      const double th = theta[kc + kk];
      double vi[4], vj[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) vi[r] = xi[kk * PP + 4 * ty + r];
#pragma unroll
      for (int c = 0; c < 4; ++c) vj[c] = xj[kk * PP + 4 * tx + c];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) s2[r][c] = dist_fold<KERNEL>(th, vi[r] - vj[c], s2[r][c], pexp);
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * ty + r, j = j0 + 4 * tx + c;
      if (i >= N || j >= N) continue;
      double v;
      if (i == j)
        v = diag;
      else if (DIV)
        v = (a * corr_profile<KERNEL>(s2[r][c], pexp)) / b;
      else
        v = a * corr_profile<KERNEL>(s2[r][c], pexp);
      R[(size_t)j * ld + i] = v;
    }
}
template <int KERNEL, bool DIV>
__global__ __launch_bounds__(256) void k_build_R(const double* __restrict__ X, int N, int d, const double* __restrict__ theta,
                                                 double a, double b, double diag, double* __restrict__ R, int ld) {
  build_R_tile<KERNEL, DIV>(X, N, d, theta, a, b, diag, R, ld);
}
// bogp_nll_batch: blockIdx.z = the parameter vector; the same tile routine on that slot's theta / normalisation / matrix
template <int KERNEL, bool DIV>
__global__ __launch_bounds__(256) void k_build_R_b(const double* __restrict__ X, int N, int d, const BatchSlot* __restrict__ slots, int ld) {
  const BatchSlot& sl = slots[blockIdx.z];
  build_R_tile<KERNEL, DIV>(X, N, d, sl.theta, sl.par[0], sl.par[1], sl.par[2], sl.ea.E, ld);
}

// ---------------------------------------------------------------------------------------------------------------
// min over all pairs i < j of |x_i - x_j|^2 for M points (row-major, d columns): what pyDOE's `maximin` criterion computes
// with scipy's pdist for each of its candidate Latin hypercubes (the reference's DoE call, search_space.py:751:
// lhs(dim, samples=N, criterion="maximin")).  Same 64 x 64 pair tiles as k_build_R.  The sum over the dimensions is
// sequential and NOT contracted into FMAs (s = s + diff * diff, the arithmetic of pdist's C loop and of oracle/philox.py),
// and a minimum does not depend on the order of its operands: the result is bit-reproducible.  out: bit pattern of the
// minimum as an unsigned integer (non-negative doubles order like their bit patterns), initialised to all ones.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_min_pdist2(const double* __restrict__ X, int M, int d, unsigned long long* __restrict__ out) {
#pragma clang fp contract(off)
  __shared__ double xi[KC * PP], xj[KC * PP];
  __shared__ unsigned long long wmin[4];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int i0 = bi * PT, j0 = bj * PT;
  double s2[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) s2[r][c] = 0.0;
  for (int kc = 0; kc < d; kc += KC) {
    __syncthreads();
    stage_points(xi, X, M, d, i0, kc, tid);
    stage_points(xj, X, M, d, j0, kc, tid);
    __syncthreads();
    const int kn = min(KC, d - kc);
    for (int kk = 0; kk < kn; ++kk) {
      double vi[4], vj[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) vi[r] = xi[kk * PP + 4 * ty + r];
#pragma unroll
      for (int c = 0; c < 4; ++c) vj[c] = xj[kk * PP + 4 * tx + c];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double diff = vi[r] - vj[c];
          const double sq = diff * diff;
          s2[r][c] = s2[r][c] + sq;
        }
    }
  }
  unsigned long long best = ~0ull;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = i0 + 4 * ty + r, j = j0 + 4 * tx + c;
      if (i < M && j < i) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(s2[r][c]);
        best = b < best ? b : best;
      }
    }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const unsigned long long ob = (unsigned long long)shfl_xor_i64((int64_t)best, o);
    best = ob < best ? ob : best;
  }
  if ((tid & 63) == 0) wmin[tid >> 6] = best;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) best = wmin[w] < best ? wmin[w] : best;
    if (best != ~0ull) atomicMin(out, best);
  }
}

hipError_t launch_min_pdist2(const double* X, int M, int d, unsigned long long* out, hipStream_t st) {
  hipError_t e = hipMemsetAsync(out, 0xFF, sizeof(unsigned long long), st);
  if (e != hipSuccess) return e;
  const unsigned nt = (unsigned)((M + PT - 1) / PT);
  hipLaunchKernelGGL(k_min_pdist2, dim3(nt, nt), 256, 0, st, X, M, d, out);
  return hipGetLastError();
}

#define BOGP_FOR_KERNEL(kernel, CALL)                      \
  switch (kernel) {                                        \
    case BOGP_KERNEL_SE: { CALL(BOGP_KERNEL_SE); } break;             \
    case BOGP_KERNEL_MATERN12: { CALL(BOGP_KERNEL_MATERN12); } break; \
    case BOGP_KERNEL_MATERN32: { CALL(BOGP_KERNEL_MATERN32); } break; \
    case BOGP_KERNEL_ABSEXP: { CALL(BOGP_KERNEL_ABSEXP); } break;     \
    default: { CALL(BOGP_KERNEL_MATERN52); } break;                   \
  }
// the correlation matrix itself also exists for cubic (no theta-derivative: k_grad_contract is not instantiated for it)
#define BOGP_FOR_KERNEL_R(kernel, CALL)                    \
  switch (kernel) {                                        \
    case BOGP_KERNEL_CUBIC: { CALL(BOGP_KERNEL_CUBIC); } break; \
    case BOGP_KERNEL_GENEXP: { CALL(BOGP_KERNEL_GENEXP); } break; \
    case BOGP_KERNEL_MATERN_NU: { CALL(BOGP_KERNEL_MATERN_NU); } break; \
    default: BOGP_FOR_KERNEL(kernel, CALL)                 \
  }

hipError_t launch_build_R(int kernel, const double* X, int N, int d, const double* theta, double off_scale, double diag,
                          double* R, int ld, hipStream_t st) {
  const int nt = (N + PT - 1) / PT;
  const dim3 grid(nt, nt);
#define CALL(K) hipLaunchKernelGGL((k_build_R<K, false>), grid, 256, 0, st, X, N, d, theta, off_scale, 1.0, diag, R, ld)
  BOGP_FOR_KERNEL_R(kernel, CALL)
#undef CALL
  return hipGetLastError();
}

hipError_t launch_build_R_batch(int kernel, bool div, const double* X, int N, int d, const BatchSlot* slots, int P, int ld, hipStream_t st) {
  const int nt = (N + PT - 1) / PT;
  const dim3 grid(nt, nt, P);
#define CALL(K)                                                                                  \
  if (div) hipLaunchKernelGGL((k_build_R_b<K, true>), grid, 256, 0, st, X, N, d, slots, ld);    \
  else hipLaunchKernelGGL((k_build_R_b<K, false>), grid, 256, 0, st, X, N, d, slots, ld)
  BOGP_FOR_KERNEL_R(kernel, CALL)
#undef CALL
  return hipGetLastError();
}

hipError_t launch_build_R_div(int kernel, const double* X, int N, int d, const double* theta, double mul, double div,
                              double diag, double* R, int ld, hipStream_t st) {
  const int nt = (N + PT - 1) / PT;
  const dim3 grid(nt, nt);
#define CALL(K) hipLaunchKernelGGL((k_build_R<K, true>), grid, 256, 0, st, X, N, d, theta, mul, div, diag, R, ld)
  BOGP_FOR_KERNEL_R(kernel, CALL)
#undef CALL
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// res_i = b_i - sum_j R_ij gamma_j (b = y - F beta, formed by the caller) with R recomputed from X exactly as k_build_R formed it (same expressions, so
// the residual is taken against the matrix that was factorised): the right-hand side of one step of iterative refinement
// of gamma = R^-1 (y - beta 1) at commit (bogp_api.hip: refine_gamma).  One workgroup per 64 rows walks all column tiles;
// the sum over j runs in a fixed order (tile by tile, then the 16 thread columns): deterministic.
// ---------------------------------------------------------------------------------------------------------------
template <int KERNEL, bool DIV>
__global__ __launch_bounds__(256) void k_resid_gamma(const double* __restrict__ X, int N, int d, const double* __restrict__ theta,
                                                     double a, double b, double diag, const double* __restrict__ bvec,
                                                     const double* __restrict__ gamma, double* __restrict__ res) {
  __shared__ double xi[KC * PP], xj[KC * PP];
  __shared__ double red[16][PT + 1];
  const int bi = blockIdx.x;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int i0 = bi * PT, nt = (N + PT - 1) / PT;
  const double pexp = kernel_exponent<KERNEL>(theta, d);
  double rowacc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int bj = 0; bj < nt; ++bj) {
    const int j0 = bj * PT;
    double s2[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) s2[r][c] = dist_init<KERNEL>();
    for (int kc = 0; kc < d; kc += KC) {
      __syncthreads();
      stage_points(xi, X, N, d, i0, kc, tid);
      stage_points(xj, X, N, d, j0, kc, tid);
      __syncthreads();
      const int kn = min(KC, d - kc);
      for (int kk = 0; kk < kn; ++kk) {
        const double th = theta[kc + kk];
        double vi[4], vj[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) vi[r] = xi[kk * PP + 4 * ty + r];
#pragma unroll
        for (int c = 0; c < 4; ++c) vj[c] = xj[kk * PP + 4 * tx + c];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) s2[r][c] = dist_fold<KERNEL>(th, vi[r] - vj[c], s2[r][c], pexp);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = j0 + 4 * tx + c;
      const double gj = j < N ? gamma[j] : 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 4 * ty + r;
        if (i >= N || j >= N) continue;
        double v;
        if (i == j)
          v = diag;
        else if (DIV)
          v = (a * corr_profile<KERNEL>(s2[r][c], pexp)) / b;
        else
          v = a * corr_profile<KERNEL>(s2[r][c], pexp);
        rowacc[r] = __builtin_fma(v, gj, rowacc[r]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[tx][4 * ty + r] = rowacc[r];
  __syncthreads();
  if (tid < PT) {
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += red[t][tid];
    const int i = i0 + tid;
    if (i < N) res[i] = bvec[i] - s;
  }
}

hipError_t launch_resid_gamma(int kernel, bool div, const double* X, int N, int d, const double* theta, double a, double b,
                              double diag, const double* bvec, const double* gamma, double* res, hipStream_t st) {
  const int nt = (N + PT - 1) / PT;
#define CALL(K)                                                                                                          \
  if (div) hipLaunchKernelGGL((k_resid_gamma<K, true>), dim3(nt), 256, 0, st, X, N, d, theta, a, b, diag, bvec, gamma, res); \
  else hipLaunchKernelGGL((k_resid_gamma<K, false>), dim3(nt), 256, 0, st, X, N, d, theta, a, b, diag, bvec, gamma, res)
  BOGP_FOR_KERNEL_R(kernel, CALL)
#undef CALL
  return hipGetLastError();
}

__global__ void k_sub_const(const double* __restrict__ y, double c, double* __restrict__ out, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) out[i] = y[i] - c;
}
hipError_t launch_sub_const(const double* y, double c, double* out, int N, hipStream_t st) {
  hipLaunchKernelGGL(k_sub_const, dim3((N + 255) / 256), 256, 0, st, y, c, out, N);
  return hipGetLastError();
}
__global__ void k_add_vec(double* __restrict__ y, const double* __restrict__ x, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) y[i] += x[i];
}
hipError_t launch_add_vec(double* y, const double* x, int N, hipStream_t st) {
  hipLaunchKernelGGL(k_add_vec, dim3((N + 255) / 256), 256, 0, st, y, x, N);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// partial[blk][k] = sum over the tile's pairs i < j of A_ij * dR0_ij/dtheta_k  (k < d),  partial[blk][d] = sum A_ij R0_ij
// Rinv arrives as `nparts` K-slices (lower triangles, part_stride doubles apart) that are added here.
// Tile (bi <= bj): i in tile bi (thread columns), j in tile bj (thread rows); blk = bj (bj + 1) / 2 + bi.
// ---------------------------------------------------------------------------------------------------------------
template <int KERNEL, int Q>
__device__ __forceinline__ void grad_contract_tile(const double* __restrict__ X, int N, int d, const double* __restrict__ theta,
                                                   const GradVecs& gv,
                                                   const double* __restrict__ qv, double c2,
                                                   const double* __restrict__ Rinv, int ld, int nparts, size_t part_stride,
                                                   double* __restrict__ partial) {
  constexpr int PTQ = 16 * Q, PPQ = PTQ + 1;  // points per tile side (64 or 32), LDS pitch
  __shared__ double xi[KC * PPQ], xj[KC * PPQ];
  __shared__ double red[4][KC + 1];
  const int bj = blockIdx.y, bi = blockIdx.x;  // j (rows, the larger index) tile, i (columns) tile
  if (bi > bj) return;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15, lane = tid & 63, wv = tid >> 6;
  const int i0 = bi * PTQ, j0 = bj * PTQ;
  double* out = partial + ((size_t)bj * (bj + 1) / 2 + bi) * (d + 1);

  // ---- pass 1: weighted distances of the 16 pairs -> r0, h, A ------------------------------------------
  double s2[Q][Q];
#pragma unroll
  for (int r = 0; r < Q; ++r)
#pragma unroll
    for (int c = 0; c < Q; ++c) s2[r][c] = 0.0;
  for (int kc = 0; kc < d; kc += KC) {
    __syncthreads();
    stage_points_q<Q, true>(xi, X, N, d, i0, kc, tid);
    stage_points_q<Q>(xj, X, N, d, j0, kc, tid);
    __syncthreads();
    const int kn = min(KC, d - kc);
    for (int kk = 0; kk < kn; ++kk) {
      const double th = theta[kc + kk];
      double vj[Q], vi[Q];
#pragma unroll
      for (int r = 0; r < Q; ++r) vj[r] = xj[kk * PPQ + Q * ty + r];
#pragma unroll
      for (int c = 0; c < Q; ++c) vi[c] = xi[kk * PPQ + 16 * c + tx];
#pragma unroll
      for (int r = 0; r < Q; ++r)
#pragma unroll
        for (int c = 0; c < Q; ++c) s2[r][c] += dist_term<KERNEL>(th, vi[c] - vj[r]);
    }
  }
  double B[Q][Q];  // A_ij * h_ij (zero for pairs outside i < j < N)
  double sd = 0.0;
#pragma unroll
  for (int r = 0; r < Q; ++r)
#pragma unroll
    for (int c = 0; c < Q; ++c) {
      const int i = i0 + Q * tx + c, j = j0 + Q * ty + r;
      double bb = 0.0;
      if (i < j && j < N) {
        double r0, h;  // corr_profile and corr_dtheta_profile with the square root and the exponential they share evaluated once
        corr_pair<KERNEL>(s2[r][c], r0, h);
        double rinv = 0.0;  // element (j, i) of the lower triangle, column-major
        for (int q = 0; q < nparts; ++q) rinv += Rinv[q * part_stride + (size_t)i * ld + j];
        // A = sum_t c_t gamma_t gamma_t^T - c0 R^-1: one vector per target (gpr.py:996-1037 with n_targets columns);
        // the theta contractions (cA) and the R0 contraction (cB) weigh the targets differently in noise_estim mode
        double gA = 0.0, gB = 0.0;
        for (int t = 0; t < gv.n; ++t) {
          const double gg = gv.v[t * gv.stride + i] * gv.v[t * gv.stride + j];
          gA = __builtin_fma(gg, gv.dcoef ? gv.dcoef[t] : gv.cA[t], gA);
          gB = __builtin_fma(gg, gv.dcoef ? gv.dcoef[8 + t] : gv.cB[t], gB);
        }
        double A = gA - gv.c0 * rinv, A2 = gB - gv.c0 * rinv;
        if (qv) {  // REML: the (L^-T Q)(L^-T Q)^T term of gpr.py:876-878, 896-898
          const double qq = qv[i] * qv[j] * c2;
          A += qq;
          A2 += qq;
        }
        bb = A * h;
        sd += A2 * r0;
      }
      B[r][c] = bb;
    }
  sd = wave_sum(sd);
  if (lane == 0) red[wv][KC] = sd;

  // ---- pass 2: the d contractions, 16 dimensions at a time ---------------------------------------------
  for (int kc = 0; kc < d; kc += KC) {
    __syncthreads();
    stage_points_q<Q, true>(xi, X, N, d, i0, kc, tid);
    stage_points_q<Q>(xj, X, N, d, j0, kc, tid);
    __syncthreads();
    const int kn = min(KC, d - kc);
    for (int kk = 0; kk < kn; ++kk) {
      double vj[Q], vi[Q];
#pragma unroll
      for (int r = 0; r < Q; ++r) vj[r] = xj[kk * PPQ + Q * ty + r];
#pragma unroll
      for (int c = 0; c < Q; ++c) vi[c] = xi[kk * PPQ + 16 * c + tx];
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < Q; ++r)
#pragma unroll
        for (int c = 0; c < Q; ++c) acc += B[r][c] * (-dtheta_weight<KERNEL>(vi[c] - vj[r]));
      acc = wave_sum(acc);
      if (lane == 0) red[wv][kk] = acc;
    }
    __syncthreads();
    if (tid < kn) out[kc + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
  }
  __syncthreads();
  if (tid == 0) out[d] = ((red[0][KC] + red[1][KC]) + red[2][KC]) + red[3][KC];
}
template <int KERNEL, int Q>
__global__ __launch_bounds__(256) void k_grad_contract(const double* __restrict__ X, int N, int d, const double* __restrict__ theta,
                                                       const GradVecs gv,
                                                       const double* __restrict__ qv, double c2,
                                                       const double* __restrict__ Rinv, int ld, int nparts, size_t part_stride,
                                                       double* __restrict__ partial) {
  grad_contract_tile<KERNEL, Q>(X, N, d, theta, gv, qv, c2, Rinv, ld, nparts, part_stride, partial);
}
// bogp_nll_batch: blockIdx.z = the parameter vector (one target, weights from that slot's k_elim_finish)
template <int KERNEL, int Q>
__global__ __launch_bounds__(256) void k_grad_contract_b(const double* __restrict__ X, int N, int d, const BatchSlot* __restrict__ slots,
                                                         int Np, int ld) {
  const BatchSlot& sl = slots[blockIdx.z];
  // the weights' descriptor in LDS, one per workgroup: built in registers it went to SCRATCH (the tile routine indexes its arrays at run time) --
  // 176 B a lane, 470 MB of writes a launch at N = 2048, P = 16 (WRITE_SIZE of profiles/r04_elim_batch_pmc.txt; the one-evaluation kernel
  // takes it as a kernel argument)
  __shared__ __attribute__((aligned(16))) unsigned char gv_raw[sizeof(GradVecs)];  // (raw: the struct has default member initialisers)
  GradVecs& gv = *reinterpret_cast<GradVecs*>(gv_raw);
  if (threadIdx.x == 0) {
    gv.v = sl.gamma; gv.stride = (size_t)Np; gv.n = 1; gv.c0 = 1.0;
    for (int t = 0; t < BOGP_MAX_TARGETS; ++t) gv.cA[t] = gv.cB[t] = 0.0;
    gv.dcoef = sl.scal + 4 * BOGP_MAX_TARGETS;
  }
  __syncthreads();
  grad_contract_tile<KERNEL, Q>(X, N, d, sl.theta, gv, nullptr, 0.0, sl.Rinv, ld, 1, (size_t)ld * ld, sl.partial);
}

// Tile side: 64 x 64 pairs (16 a thread) from N = 1025 on; 32 x 32 (4 a thread) below, 16 x 16 (one a thread) up to N = 256: a likelihood gradient of a few hundred
// points is ten-odd workgroups either way, and a thread's 16 exp / sqrt chains were 25 us of a 175-us evaluation at N = 200
static int grad_contract_q(int N) {
  constexpr int q1max = 256;
  return N <= q1max ? 1 : (N <= 1024 ? 2 : 4);
}
int grad_contract_blocks(int N) {
  const int pt = 16 * grad_contract_q(N);
  const int nt = (N + pt - 1) / pt;
  return nt * (nt + 1) / 2;
}
hipError_t launch_grad_contract_batch(int kernel, const double* X, int N, int d, const BatchSlot* slots, int P, int Np, int ld, hipStream_t st) {
  const int q = grad_contract_q(N), pt = 16 * q;
  const int nt = (N + pt - 1) / pt;
  const dim3 grid(nt, nt, P);
#define CALL(K)                                                                                           \
  if (q == 1) hipLaunchKernelGGL((k_grad_contract_b<K, 1>), grid, 256, 0, st, X, N, d, slots, Np, ld);     \
  else if (q == 2) hipLaunchKernelGGL((k_grad_contract_b<K, 2>), grid, 256, 0, st, X, N, d, slots, Np, ld); \
  else hipLaunchKernelGGL((k_grad_contract_b<K, 4>), grid, 256, 0, st, X, N, d, slots, Np, ld)
  BOGP_FOR_KERNEL(kernel, CALL)
#undef CALL
  return hipGetLastError();
}
hipError_t launch_grad_contract(int kernel, const double* X, int N, int d, const double* theta, const GradVecs& gv,
                                const double* qv, double c2, const double* Rinv, int ld, int nparts,
                                size_t part_stride, double* partial, int nblk, hipStream_t st) {
  const int q = grad_contract_q(N), pt = 16 * q;
  const int nt = (N + pt - 1) / pt;
  (void)nblk;
  const dim3 grid(nt, nt);
#define CALL(K)                                                                                                                     \
  if (q == 1) hipLaunchKernelGGL((k_grad_contract<K, 1>), grid, 256, 0, st, X, N, d, theta, gv, qv, c2, Rinv, ld, nparts, part_stride, partial); \
  else if (q == 2) hipLaunchKernelGGL((k_grad_contract<K, 2>), grid, 256, 0, st, X, N, d, theta, gv, qv, c2, Rinv, ld, nparts, part_stride, partial); \
  else hipLaunchKernelGGL((k_grad_contract<K, 4>), grid, 256, 0, st, X, N, d, theta, gv, qv, c2, Rinv, ld, nparts, part_stride, partial)
  BOGP_FOR_KERNEL(kernel, CALL)
#undef CALL
  return hipGetLastError();
}

}  // namespace bogp
