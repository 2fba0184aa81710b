// kernels_point.hip -- the ONE-POINT (and B-point) consumption path of the reference's inner optimisers.
//
// The reference maximises its acquisition function with L-BFGS-B on `criterion(x, return_dx=True)` one point per call
// (acquisition/optim/__init__.py:74-153): predict (gpr.py:486-510) + gradient (:537-576, corr_dx :600-661) + the closed
// form and its chain rule (acquisition_fun.py:139-146, 181-188, 220-227, 292-309).  r02 served that call with
// k_point_corr + two rocBLAS dtrmv + three dgemv + k_acquisition (177 us at N = 2048, the data movement is ~10 us); the
// library is gone altogether since r03 (kernels_gemm.hip serves what is left of that route: the linear trend basis).
//
// Here it is three launches for B >= 1 points, no library call:
//
//   k_point_rhs<KERNEL, NC>   one thread per training row n: r_n = corr(x, X_n) and dr_n/dx_k, written as the n-major
//                             right-hand sides  rhs[b][pass][n][NC] = [ r_n | dr_n/dx_k for the pass's <= NC-1 dimensions ]
//   k_point_tri<NC>           C = V rhs with V = L^-1 (column-major, lower): ONE pass over V serves all d + 1 columns,
//                             because  (V^T V r) . dr/dx_k = (V r) . (V dr/dx_k)  -- no second triangular product, no
//                             grid-wide dependency.  A workgroup owns 16 rows of V; lane = (row pair, 1 of 8 n-phases),
//                             4 waves = 32 columns of V per step, 16-B loads (8 x 128-B segments per wave-load); FP64 VALU
//                             FMAs against the rhs row of the lane's n (16-B loads, L1/L2 resident: 8 N NC bytes).  The
//                             n-phases are folded by 3 shuffle steps, the waves through LDS in fixed order; the block
//                             then forms  sum_rows C_0 C_c  (c = 0: |V r|^2; c = 1 + k: z . dr/dx_k).  One extra
//                             workgroup treats (gamma, w) as two more rows: gamma . rhs_c and w . rhs_c.
//   k_point_finish<NC>        one workgroup per point adds the block records in fixed order and finishes: mu, MSE
//                             (gpr.py:490, 496-510), dmu, dMSE (:561-576), the q criteria (acq_value) and their
//                             input-gradients -- one record per point.
//   k_polish_step             lock-step projected L-BFGS over B starts: consumes the records of the
//                             current trial points, accepts / backtracks per start, writes the next trial points.
//
// Work per point: N^2 (d + 1) / 2 FMAs + N (2 d + profile) for the rhs -- 4.4e7 at C3, FP64-VALU; V is read once
// (4 N^2 bytes, L2 / MALL resident between calls): the call is bound by launch + synchronisation latency, not by either.
// Everything is deterministic (no floating-point atomics): identical inputs give identical bits.
#include <algorithm>
#include <cstdlib>

#include "bogp_device.h"
#include "bogp_internal.h"

namespace bogp {

namespace {

template <int KERNEL>
__device__ __forceinline__ double point_dx_entry(double rv, double D, double e5, double theta_k, double diff) {
  // d r / d x_k (corr_dx, gpr.py:600-661); the same expressions as k_point_corr (kernels_fit.hip)
  if (KERNEL == BOGP_KERNEL_ABSEXP) return -1.0 * rv * theta_k * (diff > 0.0 ? 1.0 : (diff < 0.0 ? -1.0 : 0.0));
  if (KERNEL == BOGP_KERNEL_SE) return -2 * rv * (theta_k * diff);
  if (KERNEL == BOGP_KERNEL_MATERN32) return D > 0.0 ? (diff * theta_k / D) * (-3.0 * D * e5) : 0.0;
  if (KERNEL == BOGP_KERNEL_MATERN52) return (-(5.0 / 3.0) * (1.0 + 2.23606797749979 * D) * e5) * (theta_k * diff);
  return D > 0.0 ? -diff * theta_k / D * rv : 0.0;
}

// 64 training rows x 4 column slices per workgroup: every thread forms the weighted distance of its row (d terms, redundant
// across the 4 slices -- cheaper than a shuffle), then its quarter of the row's NC entries for every pass.
template <int KERNEL, int NC>
__global__ __launch_bounds__(256) void k_point_rhs(PointRhsArgs a) {
  constexpr int CPS = ((NC / 2 + 3) / 4) * 2;  // columns per slice (even: 16-B stores)
  __shared__ double xs[BOGP_POINT_MAX_D], ths[BOGP_POINT_MAX_D];
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const int slice = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int d = a.d;
  // the point (from the kernel arguments when it is the only one: no upload) and theta, once per workgroup: the distance
  // loop below then has ONE global load per term, and unrolled they are all in flight together (the first version
  // selected between the argument segment and memory per term: 20 dependent round trips, 8 us)
  for (int k = threadIdx.x; k < d; k += 256) {
    xs[k] = a.Xb ? a.Xb[(size_t)b * d + k] : a.x[k];
    ths[k] = a.theta[k];
  }
  __syncthreads();
  if (n >= a.Npp) return;
  double rv = 0.0, D = 0.0, e5 = 0.0;
  const bool live = n < a.N;
  const double* __restrict__ xr = a.X + (size_t)(live ? n : 0) * d;
  if (live) {
    double s2 = 0.0;
#pragma unroll 8
    for (int k = 0; k < d; ++k) s2 += dist_term<KERNEL>(ths[k], fabs(xs[k] - xr[k]));
    rv = corr_profile<KERNEL>(s2);
    D = sqrt(s2);
    if (KERNEL == BOGP_KERNEL_MATERN32) e5 = exp(-1.7320508075688772 * D);
    if (KERNEL == BOGP_KERNEL_MATERN52) e5 = exp(-2.23606797749979 * D);
  }
  const int c0 = slice * CPS, c1 = min(NC, c0 + CPS);
  for (int g = 0; g < a.npass; ++g) {
    double* row = a.rhs + (((size_t)b * a.npass + g) * a.Npp + n) * NC;
    for (int c = c0; c < c1; c += 2) {
      double v[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int cc = c + i;
        const int k = g * (NC - 1) + (cc - 1);
        v[i] = cc == 0 ? rv : ((live && k < d) ? point_dx_entry<KERNEL>(rv, D, e5, ths[k], xs[k] - xr[k]) : 0.0);
      }
      *(double2*)(row + c) = make_double2(v[0], v[1]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// MFMA flavour of the B-point path (r03, SURVEY.md 8 f2: "another MFMA GEMM L^-1 r_dx").  With enough right-hand sides --
// B (d + 1) columns -- C = V rhs is the SAME triangular product the candidate sweep runs, so it goes through k_contract16
// (kernels_posterior.hip, its cross-product epilogue <4, NCP>) against the packed V of the commit instead of k_point_tri's
// FP64-VALU loop:
//   k_point_rhs_T<KERNEL, NCP>  the right-hand sides in the contraction's layout rT[n][m], m = b NCP + c, c = 0: r_n,
//                               c = 1 + k: dr_n / dx_k, zero beyond d + 1 (NCP = 16 / 32 / 64 columns per point, ONE pass)
//   k_point_gw<NCP>             gamma . rhs_c and w . rhs_c (what k_point_tri's extra workgroup does) -> record 0 of the point
//   k_contract16<4, NCP>        sum_j C_0[j] C_c[j] per 256-column group of V                        -> records 1 .. nJ
//   k_point_finish<NCP>         unchanged (npass = 1, nRB = nJ)
// ---------------------------------------------------------------------------------------------------------------------
template <int KERNEL, int NCP>
__global__ __launch_bounds__(256) void k_point_rhs_T(PointRhsArgs a, double* __restrict__ rT, long long Mc, int Np, int B) {
  constexpr int CPS = NCP / 4;  // columns per slice
  __shared__ double xs[BOGP_POINT_MAX_D], ths[BOGP_POINT_MAX_D];
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const int slice = threadIdx.x >> 6;
  const int b = blockIdx.y;  // b >= B: the padding columns up to Mc (zeros)
  const int d = a.d;
  if (b < B) {
    for (int k = threadIdx.x; k < d; k += 256) {
      xs[k] = a.Xb[(size_t)b * d + k];
      ths[k] = a.theta[k];
    }
  }
  __syncthreads();
  if (n >= Np) return;
  const bool live = n < a.N && b < B;
  double rv = 0.0, D = 0.0, e5 = 0.0;
  const double* __restrict__ xr = a.X + (size_t)(live ? n : 0) * d;
  if (live) {
    double s2 = 0.0;
#pragma unroll 8
    for (int k = 0; k < d; ++k) s2 += dist_term<KERNEL>(ths[k], fabs(xs[k] - xr[k]));
    rv = corr_profile<KERNEL>(s2);
    D = sqrt(s2);
    if (KERNEL == BOGP_KERNEL_MATERN32) e5 = exp(-1.7320508075688772 * D);
    if (KERNEL == BOGP_KERNEL_MATERN52) e5 = exp(-2.23606797749979 * D);
  }
  double* row = rT + (size_t)n * Mc + (size_t)b * NCP;
  const int c0 = slice * CPS;
  if ((long long)b * NCP + c0 >= Mc) return;
#pragma unroll
  for (int c = c0; c < c0 + CPS; c += 2) {
    double v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int cc = c + i, k = cc - 1;
      v[i] = cc == 0 ? rv : ((live && k < d) ? point_dx_entry<KERNEL>(rv, D, e5, ths[k], xs[k] - xr[k]) : 0.0);
    }
    *(double2*)(row + c) = make_double2(v[0], v[1]);
  }
}

// one thread per right-hand-side column m: g = gamma . rhs_m, w = wvec . rhs_m over the Nr32 live rows (four interleaved
// partial sums added in a fixed order); record 0 of point b = m / NCP
template <int NCP>
__global__ __launch_bounds__(256) void k_point_gw(const double* __restrict__ rT, long long Mc, const double* __restrict__ gamma,
                                                  const double* __restrict__ wvec, int Nr32, int B, int nJ, double* __restrict__ part) {
  const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
  if (m >= (long long)B * NCP) return;
  double g[4] = {0.0, 0.0, 0.0, 0.0}, w[4] = {0.0, 0.0, 0.0, 0.0};
  const double* __restrict__ col = rT + m;
  for (int n = 0; n < Nr32; n += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double v = col[(size_t)(n + u) * Mc];
      g[u] = __builtin_fma(gamma[n + u], v, g[u]);
      w[u] = __builtin_fma(wvec[n + u], v, w[u]);
    }
  }
  const long long b = m / NCP;
  const int c = (int)(m - b * NCP);
  double* rec = part + (size_t)b * (nJ + 1) * (2 * NCP);
  rec[c] = (g[0] + g[1]) + (g[2] + g[3]);
  rec[NCP + c] = (w[0] + w[1]) + (w[2] + w[3]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Linear trend basis f(x) = [1, x] (trend.py:94-116) in the one-point path (r05; the reference's `gradient` serves it, gpr.py:556-575;
// the quadratic basis has no Jacobian there either, trend.py:138-139).  Two small launches between k_point_tri and k_point_finish:
//   k_point_wt<NC>     Tw[b][g][j][c] = sum_n W[n][j] rhs[b][g][n][c],  W = L^-T Ft (N x p): (Ft^T L^-1) [r | dr/dx] (gpr.py:570-571);
//                      a workgroup = 8 basis columns j x 32 lanes over the pass's NC right-hand-side columns, every thread walks n
//   k_point_trend_fin  per point: mu_t = f . beta, dmu_t = beta_{1+k}; under universal kriging c = Tw[:, 0] - f, Sc = (Ft^T Ft)^-1 c,
//                      uu = c . Sc, duu_k = 2 Sc . (Tw[:, 1+k] - e_{1+k})  ->  trec[b] = [mu_t, uu, dmu_t (d), duu (d)]
// k_point_finish then adds them: mu = mu_t + gamma . r, MSE = (1 - |V r|^2 + uu) sigma2, dMSE_k = 2 sigma2 (-z . dr_k + duu_k / 2).
// ---------------------------------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(256) void k_point_wt(const double* __restrict__ rhs, const double* __restrict__ W, int ldW, int N, int Npp, int npass,
                                                  int p, double* __restrict__ Tw) {
  const int b = blockIdx.z, g = blockIdx.y;
  const int j = blockIdx.x * 8 + (threadIdx.x >> 5), c = threadIdx.x & 31;
  if (j >= p || c >= NC) return;
  const double* __restrict__ col = rhs + (((size_t)b * npass + g) * Npp) * NC + c;
  const double* __restrict__ w = W + (size_t)j * ldW;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  int n = 0;
  for (; n + 4 <= N; n += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = __builtin_fma(w[n + u], col[(size_t)(n + u) * NC], s[u]);
  }
  for (; n < N; ++n) s[0] = __builtin_fma(w[n], col[(size_t)n * NC], s[0]);
  Tw[(((size_t)b * npass + g) * p + j) * NC + c] = (s[0] + s[1]) + (s[2] + s[3]);
}

template <int NC>
__global__ __launch_bounds__(256) void k_point_trend_fin(PointRhsArgs pa, const double* __restrict__ Tw, int p, const double* __restrict__ betav,
                                                         const double* __restrict__ Sinv, int estimate_trend, double* __restrict__ trec) {
  extern __shared__ double sm[];
  double* f = sm;          // [p]
  double* cv = sm + p;     // [p]
  double* Sc = sm + 2 * p; // [p]
  const int b = blockIdx.x, tid = threadIdx.x, d = pa.d, npass = pa.npass;
  for (int j = tid; j < p; j += 256) f[j] = j == 0 ? 1.0 : (pa.Xb ? pa.Xb[(size_t)b * d + (j - 1)] : pa.x[j - 1]);
  __syncthreads();
  const double* T0 = Tw + ((size_t)b * npass) * p * NC;  // pass 0: column 0 is r
  double* out = trec + (size_t)b * (2 + 2 * d);
  if (estimate_trend) {
    for (int j = tid; j < p; j += 256) cv[j] = T0[(size_t)j * NC] - f[j];
    __syncthreads();
    for (int j = tid; j < p; j += 256) {
      double a = 0.0;
      for (int i = 0; i < p; ++i) a = __builtin_fma(Sinv[(size_t)i * p + j], cv[i], a);
      Sc[j] = a;
    }
    __syncthreads();
  }
  if (tid == 0) {
    double m = 0.0, uu = 0.0;
    for (int j = 0; j < p; ++j) m = __builtin_fma(f[j], betav[j], m);  // the order of k_trend_terms: the sweep's f . beta bit for bit
    if (estimate_trend)
      for (int j = 0; j < p; ++j) uu = __builtin_fma(cv[j], Sc[j], uu);
    out[0] = m;
    out[1] = uu;
  }
  for (int k = tid; k < d; k += 256) {
    out[2 + k] = betav[1 + k];  // beta^T f_dx: the Jacobian of [1, x] is [0; I] (trend.py:109-112)
    double du = 0.0;
    if (estimate_trend) {
      const int g = k / (NC - 1), cc = 1 + (k - g * (NC - 1));
      const double* Tk = Tw + (((size_t)b * npass + g) * p) * NC + cc;
      double a = 0.0;
      for (int j = 0; j < p; ++j) a = __builtin_fma(Sc[j], Tk[(size_t)j * NC], a);
      du = 2.0 * (a - Sc[1 + k]);
    }
    out[2 + d + k] = du;
  }
}

// d acq / d x_k = a_dy * dy_k + a_dsd * dsd_k   (acquisition_fun.py:139-146, 181-188, 220-227, 292-309); the guards of
// the reference return a zero gradient, its FloatingPointError path (np.errstate(all="raise")) likewise
__device__ __forceinline__ void acq_grad_coef(int id, double par, double y, double sd, double plugin, double sigma2, double& a_dy,
                                              double& a_dsd) {
  a_dy = 0.0;
  a_dsd = 0.0;
  switch (id) {
    case BOGP_ACQ_UCB:
      a_dy = 1.0;
      a_dsd = par;
      return;
    case BOGP_ACQ_EI: {
      if (sd / sqrt(sigma2) < 1e-6) return;
      const double z = (plugin - y) / sd;
      a_dsd = norm_pdf(z);
      a_dy = -ndtr(z);
      return;
    }
    case BOGP_ACQ_EPSILON_PI: {
      const double shrink = y > 0 ? 1 - par : 1 + par;
      const double z = (plugin - shrink * y) / sd;
      const double f = norm_pdf(z) / sd;
      a_dy = -shrink * f;
      a_dsd = -z * f;
      return;
    }
    default: {  // MGFI
      if (fabs(sd) <= 1e-8) return;
      const double t = fmin(par, 22.36);
      const double var = sd * sd;
      const double z = (plugin - (y - t * var)) / sd;
      const double scale = exp(t * (plugin + t * var / 2 - y - 1));
      const double pdf = norm_pdf(z), cdf = ndtr(z);
      const double c_dy = scale * (-pdf / sd - cdf * t);
      const double c_dsd = scale * (pdf * (2.0 * t * sd - z) / sd + cdf * (t * t) * sd);
      if (isfinite(c_dy) && isfinite(c_dsd) && isfinite(scale)) {
        a_dy = c_dy;
        a_dsd = c_dsd;
      }
      return;
    }
  }
}

// NC columns per pass.  A workgroup owns RB = 64 rows of V -- one row per lane, so that every lane of a wave works on the
// SAME column n and the right-hand-side row of that column is a wave-uniform (scalar, SMEM) load: rhs values sit in SGPRs
// and feed the FMAs as scalar operands, V arrives as one coalesced 512-B load per wave and column.  (The first version
// gave each lane its own n and loaded the rhs row per lane: 8 lanes per address, so the L1 -> VGPR path moved every rhs
// byte 8 times and bounded the kernel; profiles/r03_point_tri_ab.txt.)  The 4 waves take every 4th column; `nsplit` > 1
// deals the columns to that many workgroups per row block as well (latency of ONE point: chains nsplit times shorter,
// more waves per CU); their partial C tiles meet in a scratch slab and the last arriver of the row block (a per-row-block
// ticket) adds them in fixed order -- still deterministic.  Few VGPRs (the accumulators: 2 NC) => 8 waves per SIMD.
// ROWS = rows of V per lane (1 or 2): with two, a scalar rhs row feeds 44 FMAs instead of 22.  Built to test whether the scalar
// loads bound the batched rate: they do not (neutral, profiles/r03_point_tri_ab.txt); the default is ROWS = 1.
template <int NC, int ROWS>
__global__ __launch_bounds__(256) void k_point_tri(const double* __restrict__ V, const double* __restrict__ gamma,
                                                   const double* __restrict__ wvec, const double* __restrict__ rhs_all,
                                                   PointTriArgs a) {
  constexpr int RB = 64 * ROWS;
  __shared__ double Cs[RB][NC];
  __shared__ bool s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = a.nsplit;
  const int wg = blockIdx.x / S, sp = blockIdx.x - wg * S;
  const int g = blockIdx.y, b = blockIdx.z;
  const bool special = wg == 0;            // (gamma, w) as two extra rows; scheduled first, it is as long as the longest
  const int j0 = special ? 0 : (a.nRB - wg) * RB;  // wg = 1 owns the LAST (longest) rows of V
  const int nend = special ? a.Nr32 : min(j0 + RB, a.ld);  // columns [0, nend): V is exactly zero above its diagonal
  const double* __restrict__ rhs = rhs_all + ((size_t)b * a.npass + g) * (size_t)a.Npp * NC;

  double acc[ROWS][NC];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[r][c] = 0.0;
  const double* __restrict__ vcol = V + j0 + lane;
  const bool second = ROWS > 1 && j0 + 64 + lane < a.ld;  // (ld is a multiple of 64, not of 128: the last block may be half)
  if (special) {
#pragma unroll 2
    for (int n = 4 * sp + wv; n < nend; n += 4 * S) {
      const double v = lane == 0 ? gamma[n] : (lane == 1 ? wvec[n] : 0.0);
      const double* __restrict__ rr = rhs + (size_t)n * NC;
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[0][c] = __builtin_fma(v, rr[c], acc[0][c]);
    }
  } else {
#pragma unroll 2
    for (int n = 4 * sp + wv; n < nend; n += 4 * S) {
      double v[ROWS];
      v[0] = vcol[(size_t)n * a.ld];
      if (ROWS > 1) v[ROWS - 1] = second ? vcol[(size_t)n * a.ld + 64] : 0.0;
      const double* __restrict__ rr = rhs + (size_t)n * NC;  // wave-uniform: scalar loads, SGPR operands
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const double rc = rr[c];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc[r][c] = __builtin_fma(v[r], rc, acc[r][c]);
      }
    }
  }
  // the 4 waves add their tiles in fixed order ((w0 + w1) + w2) + w3 in LDS
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (wv == w) {
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) Cs[64 * r + lane][c] = w == 0 ? acc[r][c] : Cs[64 * r + lane][c] + acc[r][c];
    }
    __syncthreads();
  }
  const size_t rb_slot = ((size_t)b * a.npass + g) * (a.nRB + 1) + wg;  // this row block of this pass of this point
  if (S > 1) {
    // the split's partial tile -> scratch; the last of the S splits adds them (s = 0 .. S-1) and carries on alone
    double* mine = a.split_scratch + (rb_slot * S + sp) * (size_t)(RB * NC);
    // write-through (sc1) stores drained by every storing wave, then ONE relaxed agent-scope ticket: no release fence
    // (a `buffer_wbl2` per workgroup serialises on the XCD's L2: 1000 of them cost 130 us, profiles/r03_point_tri_ab.txt)
    for (int t = tid; t < RB * NC; t += 256) __hip_atomic_store(mine + t, (&Cs[0][0])[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned int ticket = __hip_atomic_fetch_add(a.split_counter + rb_slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = ticket == (unsigned int)S - 1;
      if (s_last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(a.split_counter + rb_slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
      }
    }
    __syncthreads();
    if (!s_last) return;
    const double* all = a.split_scratch + rb_slot * S * (size_t)(RB * NC);
    for (int t = tid; t < RB * NC; t += 256) {
      double s = 0.0;
#pragma unroll 4
      for (int k = 0; k < S; ++k) s += __builtin_nontemporal_load(all + (size_t)k * (RB * NC) + t);
      (&Cs[0][0])[t] = s;
    }
  }
  __syncthreads();
  // the block's record: [0][c] = sum_rows C_0 C_c (special: gamma . rhs_c), [1][c] = 0 (special: w . rhs_c)
  double* rec = a.part + rb_slot * (2 * NC);
  if (tid < NC) {
    double r0, r1 = 0.0;
    if (special) {
      r0 = Cs[0][tid];
      r1 = Cs[1][tid];
    } else {
      r0 = 0.0;
#pragma unroll 8
      for (int row = 0; row < RB; ++row) r0 = __builtin_fma(Cs[row][0], Cs[row][tid], r0);
    }
    rec[tid] = r0;
    rec[NC + tid] = r1;
  }
}

// One workgroup per point, after k_point_tri (stream order: no ticket, no fence): adds the row-block records in fixed order
// and finishes the point -- mu, MSE (gpr.py:490, 496-510), dmu, dMSE (:561-576), the q criteria (acq_value) and their
// input-gradients.  Its own kernel so that the erf / exp code of the criteria does not set the register budget of
// k_point_tri's loop (132 -> 58 VGPRs: 8 waves per SIMD instead of 3).
template <int NC>
__global__ __launch_bounds__(256) void k_point_finish(PointTriArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.x;
  __shared__ double fin[4][NC];
  __shared__ double zdr[BOGP_POINT_MAX_D], gdr[BOGP_POINT_MAX_D], wdr[BOGP_POINT_MAX_D];
  __shared__ double sc[4];  // |V r|^2, gamma . r, w . r
  const int d = a.d;
  for (int gg = 0; gg < a.npass; ++gg) {
    const double* base = a.part + ((size_t)b * a.npass + gg) * (a.nRB + 1) * (2 * NC);
    if (lane < NC) {
      double s = 0.0;
#pragma unroll 8
      for (int w2 = 1 + wv; w2 <= a.nRB; w2 += 4) s += base[(size_t)w2 * (2 * NC) + lane];
      fin[wv][lane] = s;
    }
    __syncthreads();
    if (tid < NC) {
      const double tot = ((fin[0][tid] + fin[1][tid]) + fin[2][tid]) + fin[3][tid];
      const double gv = base[tid], wv2 = base[NC + tid];
      if (tid == 0) {
        if (gg == 0) {
          sc[0] = tot;
          sc[1] = gv;
          sc[2] = wv2;
        }
      } else {
        const int k = gg * (NC - 1) + tid - 1;
        if (k < d) {
          zdr[k] = tot;
          gdr[k] = gv;
          wdr[k] = wv2;
        }
      }
    }
    __syncthreads();
  }
  // posterior of the point and its input-gradients (constant trend basis; a linear one through the record of k_point_trend_fin)
  const double* tr = a.trend_rec ? a.trend_rec + (size_t)b * (2 + 2 * d) : nullptr;
  double mu, mse;
  if (tr) {
    mu = tr[0] + sc[1];
    mse = (1.0 - sc[0] + (a.estimate_trend ? tr[1] : 0.0)) * a.sigma2;
    if (mse < 0.0) mse = 0.0;
  } else {
    posterior_of_sums(sc[1], sc[2], sc[0], a.beta, a.G, a.estimate_trend, a.sigma2, mu, mse);
  }
  const double sign = a.minimize ? 1.0 : -1.0;
  const double y = sign * mu, sd = sqrt(mse);
  double* out = a.out + (size_t)b * a.rec_stride;  // [mu, mse, acq (q), dmu (d), dmse (d), dacq (q x d)]
  const int q = a.q;
  if (tid == 0) {
    out[0] = mu;
    out[1] = mse;
  }
  if (tid < q) out[2 + tid] = acq_value(a.acq_id[tid], a.acq_par[tid], y, sd, a.plugin, a.sigma2);
  for (int k = tid; k < d; k += 256) {
    const double dmu = gdr[k] + (tr ? tr[2 + k] : 0.0);  // beta^T f_dx = 0 for the constant basis (gpr.py:561)
    double m = -1.0 * zdr[k];
    if (tr) {
      if (a.estimate_trend) m += 0.5 * tr[2 + d + k];
    } else if (a.estimate_trend) {
      m += (sc[2] - 1.0) * (1.0 / a.ftft) * wdr[k];
    }
    const double dmse = 2.0 * a.sigma2 * m;  // gpr.py:573-576
    out[2 + q + k] = dmu;
    out[2 + q + d + k] = dmse;
    if (a.want_dacq) {
      const double dy = sign * dmu, dsd = dmse / (2.0 * sd);
      for (int i = 0; i < q; ++i) {
        double c_dy, c_dsd;
        acq_grad_coef(a.acq_id[i], a.acq_par[i], y, sd, a.plugin, a.sigma2, c_dy, c_dsd);
        // a zero coefficient is the reference's guard path (a zero gradient), whatever dsd is (0 / 0 at sd = 0)
        const double t1 = c_dy != 0.0 ? c_dy * dy : 0.0, t2 = c_dsd != 0.0 ? c_dsd * dsd : 0.0;
        out[2 + q + 2 * d + (size_t)i * d + k] = t1 + t2;
      }
    }
  }
  // one-point calls: the record went straight into pinned host memory; a sequence word behind it (written after every
  // storing thread's system-scope fence) lets the host see completion by polling that word instead of waiting for the stream
  if (a.done_flag) {
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(a.done_flag, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Lock-step projected L-BFGS (maximisation inside a box), one 64-thread block per start.  Every call consumes the record of
// the start's current TRIAL point (criterion 0: value and input-gradient) and writes the next trial point:
//   * Armijo test  f_t >= f + 1e-4 g . (x_t - x): accept (x, f, g <- trial; curvature pair into the history when
//     s . y > 0; step length back to 1) or halve the step from the same x;
//   * stop rules of scipy's L-BFGS-B as the reference configures it (pgtol = 1e-8 on the projected gradient, factr = 1e6
//     on the relative improvement; optim/__init__.py:94-101), or 30 consecutive rejected trials, or max evaluations;
//   * direction: two-loop recursion over the last POLISH_M pairs applied to the gradient with the coordinates that sit on
//     a bound and point outward removed; not an ascent direction -> steepest ascent and the history is dropped.
// A finished start keeps proposing its own x (the evaluation is wasted, the state never moves).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmax(v, shfl_xor_f64(v, m));
  return v;
}

// CPL = coordinates per lane: lane l owns coordinates l, l + 64, ... (d <= 64 CPL; r05: CPL up to 5 = BOGP_MAX_DIM, where r03-r04 refused
// d > 64).  CPL = 1 is the r03 kernel operation for operation.  The state's vectors have a pitch of DP = 64 CPL doubles.
template <int CPL>
__global__ __launch_bounds__(64) void k_polish_step(PolishArgs a) {
  constexpr int DP = 64 * CPL;
  const int b = blockIdx.x, lane = threadIdx.x;
  const int d = a.d;
  bool on[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) on[c] = lane + 64 * c < d;
  double* st = a.state + (size_t)b * a.state_stride;
  // state layout: [f, alpha, nhist, head, nfail, done, nevals, first] (8) | x (DP) | g (DP) | S (M x DP) | Y (M x DP) | rho (M)
  double* sx = st + 8;
  double* sg = sx + DP;
  double* sS = sg + DP;
  double* sY = sS + POLISH_M * DP;
  double* srho = sY + POLISH_M * DP;
  const double* rec = a.rec + (size_t)b * a.rec_stride;
  const double ft = rec[2];  // criterion 0 at the trial point
  double gt[CPL], xt[CPL], lo[CPL], hi[CPL], x[CPL], gcur[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int k = lane + 64 * c;
    gt[c] = on[c] ? rec[2 + a.q + 2 * d + k] : 0.0;  // its gradient
    xt[c] = on[c] ? a.Xt[(size_t)b * d + k] : 0.0;
    lo[c] = on[c] ? a.lo[k] : 0.0;
    hi[c] = on[c] ? a.hi[k] : 0.0;
    x[c] = on[c] ? sx[k] : 0.0;
    gcur[c] = on[c] ? sg[k] : 0.0;
  }
  // sum over the lane's coordinates, then over the wave (CPL = 1: exactly the r03 order)
  auto dot = [&](auto f) {
    double v = f(0);
#pragma unroll
    for (int c = 1; c < CPL; ++c) v += f(c);
    return wave_sum(v);
  };
  auto amax = [&](auto f) {
    double v = f(0);
#pragma unroll
    for (int c = 1; c < CPL; ++c) v = fmax(v, f(c));
    return wave_max(v);
  };
  double f = st[0], alpha = st[1];
  int nhist = (int)st[2], head = (int)st[3], nfail = (int)st[4], done = (int)st[5], nevals = (int)st[6];
  const int first = (int)st[7];
  if (done) return;  // the trial buffer already holds x
  nevals += 1;
  bool accepted = false;
  if (first) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) { x[c] = xt[c]; gcur[c] = gt[c]; }
    f = ft;
    accepted = true;
    if (!(ft == ft) || !isfinite(ft)) done = 1;  // nothing to climb from
  } else {
    const double slope = dot([&](int c) { return gcur[c] * (xt[c] - x[c]); });
    const bool finite = (ft == ft) && isfinite(ft) && isfinite(dot([&](int c) { return fabs(gt[c]); }));
    if (finite && ft >= f + 1e-4 * slope && ft > f) {
      double sv[CPL], yv[CPL];  // pair of the equivalent minimisation of -f
#pragma unroll
      for (int c = 0; c < CPL; ++c) { sv[c] = xt[c] - x[c]; yv[c] = gcur[c] - gt[c]; }
      const double sy = dot([&](int c) { return sv[c] * yv[c]; }), yy = dot([&](int c) { return yv[c] * yv[c]; });
      if (sy > 2.2e-16 * yy) {
#pragma unroll
        for (int c = 0; c < CPL; ++c)
          if (on[c]) {
            sS[head * DP + lane + 64 * c] = sv[c];
            sY[head * DP + lane + 64 * c] = yv[c];
          }
        if (lane == 0) srho[head] = 1.0 / sy;
        head = (head + 1) % POLISH_M;
        nhist = nhist < POLISH_M ? nhist + 1 : POLISH_M;
      }
      const double gain = ft - f;
      const double scale_f = fmax(fmax(fabs(f), fabs(ft)), 1.0);
#pragma unroll
      for (int c = 0; c < CPL; ++c) { x[c] = xt[c]; gcur[c] = gt[c]; }
      f = ft;
      alpha = 1.0;
      nfail = 0;
      accepted = true;
      if (gain <= a.factr_eps * scale_f) done = 1;  // scipy: (f_k - f_{k+1}) / max(|f_k|, |f_{k+1}|, 1) <= factr * eps
    } else {
      alpha *= 0.5;
      nfail += 1;
      if (nfail >= 30) done = 1;
    }
  }
  if (accepted && !done) {
    // projected gradient of the ascent
    if (amax([&](int c) { return on[c] ? fabs(fmin(fmax(x[c] + gcur[c], lo[c]), hi[c]) - x[c]) : 0.0; }) < a.pgtol) done = 1;
  }
  if (nevals >= a.max_evals) done = 1;
  // next trial point
  double xn[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) xn[c] = x[c];
  if (!done) {
    __syncthreads();  // history written above is read below (one wave: a compiler barrier is all this is)
    bool blocked[CPL];
    double gf[CPL], p[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      blocked[c] = on[c] && ((x[c] <= lo[c] && gcur[c] < 0.0) || (x[c] >= hi[c] && gcur[c] > 0.0));
      gf[c] = (on[c] && !blocked[c]) ? gcur[c] : 0.0;
      p[c] = gf[c];
    }
    const double gnorm = sqrt(dot([&](int c) { return gf[c] * gf[c]; }));
    if (nhist > 0) {
      double qv[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) qv[c] = gf[c];
      double al[POLISH_M];
#pragma unroll
      for (int i = 0; i < POLISH_M; ++i) {
        al[i] = 0.0;
        if (i < nhist) {
          const int h = (head - 1 - i + 2 * POLISH_M) % POLISH_M;
          al[i] = srho[h] * dot([&](int c) { return (on[c] ? sS[h * DP + lane + 64 * c] : 0.0) * qv[c]; });
#pragma unroll
          for (int c = 0; c < CPL; ++c) qv[c] -= al[i] * (on[c] ? sY[h * DP + lane + 64 * c] : 0.0);
        }
      }
      const int hl = (head - 1 + POLISH_M) % POLISH_M;
      const double gam = 1.0 / (srho[hl] * dot([&](int c) { const double yl = on[c] ? sY[hl * DP + lane + 64 * c] : 0.0; return yl * yl; }));
#pragma unroll
      for (int c = 0; c < CPL; ++c) qv[c] *= gam;
#pragma unroll
      for (int i = POLISH_M - 1; i >= 0; --i) {
        if (i < nhist) {
          const int h = (head - 1 - i + 2 * POLISH_M) % POLISH_M;
          const double be = srho[h] * dot([&](int c) { return (on[c] ? sY[h * DP + lane + 64 * c] : 0.0) * qv[c]; });
#pragma unroll
          for (int c = 0; c < CPL; ++c) qv[c] += (al[i] - be) * (on[c] ? sS[h * DP + lane + 64 * c] : 0.0);
        }
      }
#pragma unroll
      for (int c = 0; c < CPL; ++c) p[c] = blocked[c] ? 0.0 : qv[c];
      const double asc = dot([&](int c) { return p[c] * gf[c]; });
      if (!(asc > 0.0) || !isfinite(asc)) {  // not an ascent direction: steepest ascent, history dropped
#pragma unroll
        for (int c = 0; c < CPL; ++c) p[c] = gf[c];
        nhist = 0;
        head = 0;
      }
    }
    if (nhist == 0 && gnorm > 0.0) {  // first move: a fixed length along the gradient
#pragma unroll
      for (int c = 0; c < CPL; ++c) p[c] = gf[c] * (a.first_step / gnorm);
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) xn[c] = on[c] ? fmin(fmax(x[c] + alpha * p[c], lo[c]), hi[c]) : 0.0;
    if (amax([&](int c) { return fabs(xn[c] - x[c]); }) == 0.0) done = 1;  // the step no longer moves any coordinate
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    if (done) xn[c] = x[c];
    if (on[c]) {
      sx[lane + 64 * c] = x[c];
      sg[lane + 64 * c] = gcur[c];
      a.Xt[(size_t)b * d + lane + 64 * c] = xn[c];
    }
  }
  if (lane == 0) {
    st[0] = f;
    st[1] = alpha;
    st[2] = (double)nhist;
    st[3] = (double)head;
    st[4] = (double)nfail;
    st[5] = (double)done;
    st[6] = (double)nevals;
    st[7] = 0.0;
    if (done) atomicAdd(a.n_done, 1u);
  }
}

}  // namespace

int point_columns_per_pass(int d) {
  return d + 1 <= 12 ? 12 : 22;  // (12 / 22 forced either way: profiles/r03_point_tri_ab.txt)
}
int point_passes(int d) {
  const int nc = point_columns_per_pass(d);
  return (d + nc - 2) / (nc - 1);
}

hipError_t launch_point_rhs(int kernel, const PointRhsArgs& a, int B, hipStream_t st) {
  dim3 grid((a.Npp + 63) / 64, B);
  const int nc = point_columns_per_pass(a.d);
#define CALL(K)                                                                  \
  if (nc == 12) hipLaunchKernelGGL((k_point_rhs<K, 12>), grid, 256, 0, st, a);    \
  else hipLaunchKernelGGL((k_point_rhs<K, 22>), grid, 256, 0, st, a)
  switch (kernel) {
    case BOGP_KERNEL_SE: CALL(BOGP_KERNEL_SE); break;
    case BOGP_KERNEL_MATERN12: CALL(BOGP_KERNEL_MATERN12); break;
    case BOGP_KERNEL_MATERN32: CALL(BOGP_KERNEL_MATERN32); break;
    case BOGP_KERNEL_ABSEXP: CALL(BOGP_KERNEL_ABSEXP); break;
    default: CALL(BOGP_KERNEL_MATERN52); break;
  }
#undef CALL
  return hipGetLastError();
}

// splits of a 64-row block for B points.  Measured at C3 size (profiles/r03_point_tri_ab.txt): the loop is a chain of
// dependent (scalar load -> FMA) steps, so shorter chains win well beyond the point where the row blocks alone fill the
// chip -- 16 splits for one point (47 vs 168 us unsplit), 8 for 8 points, 4 from 32 points up (179 vs 283 us at B = 32).
void point_tri_geometry(int N, int d, int B, int* rb, int* nsplit) {
  const int npass = point_passes(d);
  // rows of V per lane: 2 halves the scalar rhs traffic per FMA -- measured NEUTRAL (524 vs 525 us at B = 128): what bounds the
  // batched rate is the V stream from L2 (8 bytes per 22 FMAs, ~4 TB/s at B = 128), not the rhs (profiles/r03_point_tri_ab.txt)
  constexpr int rows = 1;
  const int r = 64 * rows;
  const long long wgs = (long long)((N + r - 1) / r + 1) * npass * B;
  int s = wgs < 128 ? 16 : (wgs < 512 ? 8 : 4);
  s = std::max(1, std::min(s, (N + 15) / 16));  // at least ~4 columns per wave and split
  *rb = r;
  *nsplit = s;
}

hipError_t launch_point_tri(const PointTriArgs& a, int B, hipStream_t st) {
  dim3 grid((a.nRB + 1) * a.nsplit, a.npass, B);
  if (point_columns_per_pass(a.d) == 12) {
    if (a.rb == 128) hipLaunchKernelGGL((k_point_tri<12, 2>), grid, 256, 0, st, a.V, a.gamma, a.wvec, a.rhs, a);
    else hipLaunchKernelGGL((k_point_tri<12, 1>), grid, 256, 0, st, a.V, a.gamma, a.wvec, a.rhs, a);
    hipLaunchKernelGGL(k_point_finish<12>, dim3(B), 256, 0, st, a);
  } else {
    if (a.rb == 128) hipLaunchKernelGGL((k_point_tri<22, 2>), grid, 256, 0, st, a.V, a.gamma, a.wvec, a.rhs, a);
    else hipLaunchKernelGGL((k_point_tri<22, 1>), grid, 256, 0, st, a.V, a.gamma, a.wvec, a.rhs, a);
    hipLaunchKernelGGL(k_point_finish<22>, dim3(B), 256, 0, st, a);
  }
  return hipGetLastError();
}

hipError_t launch_point_trend(const PointRhsArgs& ra, const double* W, int ldW, int p, const double* betav, const double* Sinv,
                              int estimate_trend, double* Tw, double* trec, int B, hipStream_t st) {
  const size_t shm = (size_t)3 * p * sizeof(double);
  if (point_columns_per_pass(ra.d) == 12) {
    if (estimate_trend) hipLaunchKernelGGL(k_point_wt<12>, dim3((p + 7) / 8, ra.npass, B), 256, 0, st, ra.rhs, W, ldW, ra.N, ra.Npp, ra.npass, p, Tw);
    hipLaunchKernelGGL(k_point_trend_fin<12>, dim3(B), 256, shm, st, ra, Tw, p, betav, Sinv, estimate_trend, trec);
  } else {
    if (estimate_trend) hipLaunchKernelGGL(k_point_wt<22>, dim3((p + 7) / 8, ra.npass, B), 256, 0, st, ra.rhs, W, ldW, ra.N, ra.Npp, ra.npass, p, Tw);
    hipLaunchKernelGGL(k_point_trend_fin<22>, dim3(B), 256, shm, st, ra, Tw, p, betav, Sinv, estimate_trend, trec);
  }
  return hipGetLastError();
}

int point_mfma_columns(int d) { return d + 1 <= 16 ? 16 : (d + 1 <= 32 ? 32 : (d + 1 <= 64 ? 64 : 0)); }

hipError_t launch_point_rhs_T(int kernel, const PointRhsArgs& a, int ncp, double* rT, long long Mc, int Np, int B, hipStream_t st) {
  dim3 grid((Np + 63) / 64, (unsigned)(Mc / ncp));
#define CALLN(K, N_) hipLaunchKernelGGL((k_point_rhs_T<K, N_>), grid, 256, 0, st, a, rT, Mc, Np, B)
#define CALL(K)                                  \
  if (ncp == 16) CALLN(K, 16);                   \
  else if (ncp == 32) CALLN(K, 32);              \
  else CALLN(K, 64)
  switch (kernel) {
    case BOGP_KERNEL_SE: CALL(BOGP_KERNEL_SE); break;
    case BOGP_KERNEL_MATERN12: CALL(BOGP_KERNEL_MATERN12); break;
    case BOGP_KERNEL_MATERN32: CALL(BOGP_KERNEL_MATERN32); break;
    case BOGP_KERNEL_ABSEXP: CALL(BOGP_KERNEL_ABSEXP); break;
    default: CALL(BOGP_KERNEL_MATERN52); break;
  }
#undef CALL
#undef CALLN
  return hipGetLastError();
}

hipError_t launch_point_gw(int ncp, const double* rT, long long Mc, const double* gamma, const double* wvec, int Nr32, int B, int nJ,
                           double* part, hipStream_t st) {
  const dim3 grid((unsigned)(((long long)B * ncp + 255) / 256));
  if (ncp == 16) hipLaunchKernelGGL(k_point_gw<16>, grid, 256, 0, st, rT, Mc, gamma, wvec, Nr32, B, nJ, part);
  else if (ncp == 32) hipLaunchKernelGGL(k_point_gw<32>, grid, 256, 0, st, rT, Mc, gamma, wvec, Nr32, B, nJ, part);
  else hipLaunchKernelGGL(k_point_gw<64>, grid, 256, 0, st, rT, Mc, gamma, wvec, Nr32, B, nJ, part);
  return hipGetLastError();
}

// a.npass = 1, a.nRB = column groups, a.part = the records of k_point_gw + k_contract16<4, NCP>
hipError_t launch_point_finish_mfma(const PointTriArgs& a, int ncp, int B, hipStream_t st) {
  if (ncp == 16) hipLaunchKernelGGL(k_point_finish<16>, dim3(B), 256, 0, st, a);
  else if (ncp == 32) hipLaunchKernelGGL(k_point_finish<32>, dim3(B), 256, 0, st, a);
  else hipLaunchKernelGGL(k_point_finish<64>, dim3(B), 256, 0, st, a);
  return hipGetLastError();
}

size_t polish_state_doubles(int d) {
  const size_t DP = (size_t)((d + 63) / 64) * 64;
  return 8 + 2 * DP + 2 * (size_t)POLISH_M * DP + POLISH_M;
}

hipError_t launch_polish_step(const PolishArgs& a, int B, hipStream_t st) {
  switch ((a.d + 63) / 64) {
    case 1: hipLaunchKernelGGL(k_polish_step<1>, dim3(B), 64, 0, st, a); break;
    case 2: hipLaunchKernelGGL(k_polish_step<2>, dim3(B), 64, 0, st, a); break;
    case 3: hipLaunchKernelGGL(k_polish_step<3>, dim3(B), 64, 0, st, a); break;
    case 4: hipLaunchKernelGGL(k_polish_step<4>, dim3(B), 64, 0, st, a); break;
    case 5: hipLaunchKernelGGL(k_polish_step<5>, dim3(B), 64, 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace bogp
