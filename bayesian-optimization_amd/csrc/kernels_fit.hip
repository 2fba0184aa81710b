// kernels_fit.hip -- fit-side kernels of libbogp (gfx950): everything around the factorisation (kernels_chol.hip).
//
//   (k_build_R and k_grad_contract, the two N^2 d pair kernels, live in kernels_pairs.hip)
//   k_scale_transpose theta-scaled, transposed copy of X that the sweep's producer reads with scalar loads
//   k_pack_V         L^-1 -> MFMA B-fragment order for k_contract
//   k_logdet         sum(log(diag(L)))  (gpr.py:943-945)
//   k_gemv2, k_fit_rho, k_trace_gg, k_sumsq   products with V = L^-1 / U = L^-T and the scalars of the likelihood
//   k_trend_*        polynomial trend bases with p > 1 columns (trend.py:94-142)
//   k_point_corr     r and dr/dx at ONE point for GaussianProcess.gradient (gpr.py:537-576, corr_dx :600-661)
#include "bogp_device.h"
#include "bogp_internal.h"

namespace bogp {

__global__ void k_scale_transpose(const double* __restrict__ X, int N, int d, int Np, const double* __restrict__ sth,
                                  double* __restrict__ XthT, double* __restrict__ xnorm) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= Np) return;
  double s = 0.0;  // |sqrt(theta) x_n|^2, summed in dimension order: the training side of k_corr_mfma's cross-term distance
  for (int k = 0; k < d; ++k) {
    const double v = n < N ? X[(size_t)n * d + k] * sth[k] : 0.0;
    XthT[(size_t)k * Np + n] = v;
    s = __builtin_fma(v, v, s);
  }
  if (xnorm) xnorm[n] = s;
}
hipError_t launch_scale_transpose(const double* X, int N, int d, int Np, const double* sqrt_theta, double* XthT, double* xnorm,
                                  hipStream_t st) {
  hipLaunchKernelGGL(k_scale_transpose, dim3((Np + 255) / 256), 256, 0, st, X, N, d, Np, sqrt_theta, XthT, xnorm);
  return hipGetLastError();
}

// Vp[(jt*NKP + kp)*64 + lane] = ( V[j][8kp + k], V[j][8kp + 4 + k] ),  j = 16 jt + (lane & 15), k = lane >> 4,
// zero above the diagonal and in the padding.  Vcm is column-major: V(j, n) = Vcm[j + n*ld].
__global__ __launch_bounds__(64) void k_pack_V(const double* __restrict__ Vcm, int N, int ld, int NKP, double2* __restrict__ Vp) {
  const int jt = blockIdx.y, kp = blockIdx.x, lane = threadIdx.x;
  const int j = 16 * jt + (lane & 15);
  const int n0 = 8 * kp + (lane >> 4), n1 = n0 + 4;
  double2 v;
  v.x = (j < N && n0 <= j) ? Vcm[(size_t)n0 * ld + j] : 0.0;
  v.y = (j < N && n1 <= j) ? Vcm[(size_t)n1 * ld + j] : 0.0;
  Vp[((size_t)jt * NKP + kp) * 64 + lane] = v;
}
hipError_t launch_pack_V(const double* Vcm, int N, int ld, int Np, double2* Vp, hipStream_t st) {
  hipLaunchKernelGGL(k_pack_V, dim3(Np / 8, Np / 16), 64, 0, st, Vcm, N, ld, Np / 8, Vp);
  return hipGetLastError();
}

// r05, polynomial trends under universal kriging with p > 32 columns: the u term of the variance (gpr.py:496-498,
//   u = G^-T (Ft^T L^-1 r - f(x*)),  MSE = (1 - |L^-1 r|^2 + |u|^2) sigma2)
// becomes p EXTRA ROWS of the triangular factor.  With W = L^-T Ft (N x p) and B = G^-T (p x p, LOWER triangular because the G of the QR is
// upper triangular):  u = (B W^T) r + B (-f).  So the matrix
//        [ V        0   0 ]   rows 0 .. Np - 1           (V = L^-1, zero padded)
//   Vx = [ 0        0   0 ]   rows Np .. Ne - 1          (Ne = Np rounded up to a whole column group of k_contract16)
//        [ B W^T    0   B ]   rows Ne .. Ne + p - 1
// is lower triangular, and against the extended chunk column  rx = [r ; 0 ; -f(x*)]  the contraction kernel -- unchanged -- returns
// |V r|^2 in its first Ne / 256 column groups and |u|^2 in the groups behind them; k_acquisition subtracts the former and adds the latter.
// The two per-chunk tile products of the r02-r04 path (T = W^T r: 2 N p flop a candidate at k_mm128's 60 %, then T S^-1) and their
// second pass over the 1-GiB chunk are gone: the same 2 N p flop run inside k_contract16 at its 0.87, on the chunk it reads anyway.
// Same packing as k_pack_V: Vp[(jt*NKP + kp)*64 + lane] = ( Vx[j][8kp + k], Vx[j][8kp + 4 + k] ), j = 16 jt + (lane & 15), k = lane >> 4.
// At = W G^-1 (N x p, column-major, ld = ldA): At(n, i) = (B W^T)(i, n);  Ginv column-major p x p: B(i, k) = Ginv(k, i).
__global__ __launch_bounds__(64) void k_pack_Vx(const double* __restrict__ Vcm, int N, int ld, const double* __restrict__ At, int ldA,
                                                const double* __restrict__ Ginv, int p, int Ne, int NKP, double2* __restrict__ Vp) {
  const int jt = blockIdx.y, kp = blockIdx.x, lane = threadIdx.x;
  const int j = 16 * jt + (lane & 15);
  const int n0 = 8 * kp + (lane >> 4), n1 = n0 + 4;
  auto at = [&](int n) -> double {
    if (j < Ne) return (j < N && n <= j) ? Vcm[(size_t)n * ld + j] : 0.0;
    const int i = j - Ne;
    if (i >= p) return 0.0;
    if (n < N) return At[(size_t)i * ldA + n];
    const int k = n - Ne;
    return (k >= 0 && k <= i) ? Ginv[(size_t)i * p + k] : 0.0;
  };
  double2 v;
  v.x = at(n0);
  v.y = at(n1);
  Vp[((size_t)jt * NKP + kp) * 64 + lane] = v;
}
hipError_t launch_pack_Vx(const double* Vcm, int N, int ld, const double* At, int ldA, const double* Ginv, int p, int Ne, int Nt,
                          double2* Vp, hipStream_t st) {
  hipLaunchKernelGGL(k_pack_Vx, dim3(Nt / 8, Nt / 16), 64, 0, st, Vcm, N, ld, At, ldA, Ginv, p, Ne, Nt / 8, Vp);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_logdet(const double* __restrict__ L, int N, int ld, double* out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) s += log(L[(size_t)i * ld + i]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}
hipError_t launch_logdet(const double* L, int N, int ld, double* out, hipStream_t st) {
  hipLaunchKernelGGL(k_logdet, dim3(1), 256, 0, st, L, N, ld, out);
  return hipGetLastError();
}

// ---- matrix-vector products with V = L^-1 / U = L^-T (what the reference's triangular solves become) --------------
// y0 = M x0 and y1 = M x1 (x1 / y1 may be null) for a column-major N x N matrix M that is lower (tri = 1: columns <= row)
// or upper (tri = 2: columns >= row) triangular with explicit zeros elsewhere.  Stage 1: grid (row blocks of 64,
// column segments of 256) -> part[seg][rhs][row]; stage 2 adds the segments in a fixed order (deterministic).
constexpr int GV_SEG = 256;
__global__ __launch_bounds__(256) void k_gemv2_part(const double* __restrict__ M, int ld, int N, int Nr, int tri,
                                                    const double* __restrict__ x0, const double* __restrict__ x1,
                                                    double* __restrict__ part, double* __restrict__ y0, double* __restrict__ y1) {
  __shared__ double red[2][4][64];
  const int tid = threadIdx.x, rl = tid & 63, cg = tid >> 6;
  const int rb = blockIdx.x * 64, seg = blockIdx.y;
  const int c0 = seg * GV_SEG, c1 = min(N, c0 + GV_SEG);
  const int row = rb + rl;
  double a0 = 0.0, a1 = 0.0;
  const bool live = tri == 1 ? (c0 <= rb + 63) : (tri == 2 ? (c1 - 1 >= rb) : true);
  if (live && row < N) {
    const double* m = M + row;
#pragma unroll 8
    for (int c = c0 + cg; c < c1; c += 4) {
      const double v = m[(size_t)c * ld];
      a0 = __builtin_fma(v, x0[c], a0);
      if (x1) a1 = __builtin_fma(v, x1[c], a1);
    }
  }
  red[0][cg][rl] = a0;
  red[1][cg][rl] = a1;
  __syncthreads();
  if (tid < 128) {
    const int r = tid >> 6, q = tid & 63;
    const double v = ((red[r][0][q] + red[r][1][q]) + red[r][2][q]) + red[r][3][q];
    if (y0 == nullptr) {
      part[((size_t)seg * 2 + r) * Nr + rb + q] = v;
    } else if (rb + q < N) {  // one column segment (N <= 256): the result itself, no k_gemv2_sum launch -- the same `0.0 + v` it would form
      double* y = r == 0 ? y0 : y1;
      if (y) y[rb + q] = 0.0 + v;
    }
  }
}
__global__ void k_gemv2_sum(const double* __restrict__ part, int nseg, int Nr, int N, double* __restrict__ y0, double* __restrict__ y1) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= N) return;
  double s0 = 0.0, s1 = 0.0;
  for (int sg = 0; sg < nseg; ++sg) {
    s0 += part[((size_t)sg * 2 + 0) * Nr + row];
    if (y1) s1 += part[((size_t)sg * 2 + 1) * Nr + row];
  }
  y0[row] = s0;
  if (y1) y1[row] = s1;
}
size_t gemv2_scratch_doubles(int N) {
  const int nrb = (N + 63) / 64, nseg = (N + GV_SEG - 1) / GV_SEG;
  return (size_t)nseg * 2 * nrb * 64;
}
hipError_t launch_gemv2(const double* M, int ld, int N, int tri, const double* x0, const double* x1, double* y0, double* y1,
                        double* scratch, hipStream_t st) {
  const int nrb = (N + 63) / 64, nseg = (N + GV_SEG - 1) / GV_SEG, Nr = nrb * 64;
  if (nseg == 1) {  // N <= 256: one launch (a likelihood evaluation at these sizes is launch-latency bound: ~6 us per launch)
    hipLaunchKernelGGL(k_gemv2_part, dim3(nrb, 1), 256, 0, st, M, ld, N, Nr, tri, x0, x1, scratch, y0, y1);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_gemv2_part, dim3(nrb, nseg), 256, 0, st, M, ld, N, Nr, tri, x0, x1, scratch, (double*)nullptr, (double*)nullptr);
  hipLaunchKernelGGL(k_gemv2_sum, dim3((N + 255) / 256), 256, 0, st, scratch, nseg, Nr, N, y0, y1);
  return hipGetLastError();
}

// ---- rho and the scalars of the concentrated likelihood, on the device (gpr.py:803-808 for p = 1) --------------------
// Ordinary kriging: economic QR of the single column Ft: G = -|Ft|, Q = Ft / G, rho = Yt - Q (Q^T Yt);
// simple kriging: rho = Yt - beta Ft (Ft = L^-1 1).  scal[1] = |Ft|, scal[2] = Ft . Yt, scal[3] = rho . rho.
__device__ __forceinline__ double block_sum_1024(double v, double* red /* [16] */) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) s += red[wv];
  return s;
}
__global__ __launch_bounds__(1024) void k_fit_rho(const double* __restrict__ Yt, const double* __restrict__ Ft, int N,
                                                  int estimate_trend, double beta, double* __restrict__ rho,
                                                  double* __restrict__ scal, const double* __restrict__ Lfac, int ldL,
                                                  double* __restrict__ gcoef, int mode, double s2t_host) {
  __shared__ double red[16];
  if (Lfac != nullptr) {  // sum(log(diag(L))) into scal[0]: k_logdet's partial sums and tree, by the first 256 threads (one launch less)
    __shared__ double red256[256];
    if (threadIdx.x < 256) {
      double s = 0.0;
      for (int i = threadIdx.x; i < N; i += 256) s += log(Lfac[(size_t)i * ldL + i]);
      red256[threadIdx.x] = s;
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red256[threadIdx.x] += red256[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) scal[0] = red256[0];
  }
  double sff = 0.0, sfy = 0.0;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    const double f = Ft[i];
    sff = __builtin_fma(f, f, sff);
    sfy = __builtin_fma(f, Yt[i], sfy);
  }
  sff = block_sum_1024(sff, red);
  sfy = block_sum_1024(sfy, red);
  const double nrm = sqrt(sff);
  double coef;
  if (estimate_trend) {
    const double G = -nrm, qty = sfy / G;
    coef = -(qty / G);
  } else {
    coef = -beta;
  }
  double srr = 0.0;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    const double r = __builtin_fma(coef, Ft[i], Yt[i]);
    rho[i] = r;
    srr = __builtin_fma(r, r, srr);
  }
  srr = block_sum_1024(srr, red);
  if (threadIdx.x == 0) {
    scal[1] = nrm;
    scal[2] = sfy;
    scal[3] = srr;
    if (gcoef != nullptr) {  // one target: k_grad_coef's two weights here (one launch less per likelihood gradient)
      const double s2t = mode == BOGP_MODE_NOISY ? s2t_host : (mode == BOGP_MODE_NOISELESS ? srr / (N - (estimate_trend ? 1 : 0)) : srr / N);
      gcoef[8] = 1.0 / s2t;
      gcoef[0] = 0.0 + 1.0 / s2t;
    }
  }
}
hipError_t launch_fit_rho(const double* Yt, const double* Ft, int N, int estimate_trend, double beta, double* rho, double* scal,
                          hipStream_t st, const double* Lfac, int ldL, double* coef, int mode, double s2t_host) {
  hipLaunchKernelGGL(k_fit_rho, dim3(1), 1024, 0, st, Yt, Ft, N, estimate_trend, beta, rho, scal, Lfac, ldL, coef, mode, s2t_host);
  return hipGetLastError();
}

// Per-target weights of gamma_t gamma_t^T in the likelihood gradient (bogp_nll), formed ON the device from the scalars of
// k_fit_rho so that the gradient kernels can be queued behind the factorisation without a host round trip in between:
//   s2t_t = rho_t.rho_t / (N - k)  [noiseless, target 0; / N for further targets and in noise_estim mode],  s2t_host [noisy]
//   cB[t] = 1 / s2t_t;  cA[t] = cB[t] [noisy]  or  sum_t 1 / s2t_t [otherwise]          (same IEEE operations as the host's)
// coef: cA[0..7], cB[8..15].
__global__ void k_grad_coef(const double* __restrict__ scal, int n_t, int mode, int N, int krank, double s2t_host,
                            double* __restrict__ coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double inv_sum = 0.0;
  for (int t = 0; t < n_t; ++t) {
    const double rss = scal[4 * t + 3];
    const double s2t = mode == BOGP_MODE_NOISY ? s2t_host : (mode == BOGP_MODE_NOISELESS && t == 0 ? rss / (N - krank) : rss / N);
    coef[8 + t] = 1.0 / s2t;
    inv_sum += 1.0 / s2t;
  }
  for (int t = 0; t < n_t; ++t) coef[t] = mode == BOGP_MODE_NOISY ? coef[8 + t] : inv_sum;
}
hipError_t launch_grad_coef(const double* scal, int n_t, int mode, int N, int krank, double s2t_host, double* coef, hipStream_t st) {
  hipLaunchKernelGGL(k_grad_coef, dim3(1), 64, 0, st, scal, n_t, mode, N, krank, s2t_host, coef);
  return hipGetLastError();
}

// The likelihood's read-back in ONE launch instead of two copy commands: the 64 scalars (incl. the factorisation's info word) and
// the nS gradient sums into device-mapped pinned host memory, then a sequence word the host polls (as k_point_finish does).
__global__ void k_fit_gather(const double* __restrict__ scal, const double* __restrict__ S, int nS, double* __restrict__ out_scal,
                             double* __restrict__ out_S, unsigned long long* __restrict__ flag, unsigned long long seq) {
  const int i = threadIdx.x;
  if (i < 64) out_scal[i] = scal[i];
  for (int k = i; k < nS; k += blockDim.x) out_S[k] = S[k];
  __threadfence_system();
  __syncthreads();
  if (i == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_fit_gather(const double* scal, const double* S, int nS, double* out_scal, double* out_S, unsigned long long* flag,
                             unsigned long long seq, hipStream_t st) {
  hipLaunchKernelGGL(k_fit_gather, dim3(1), 128, 0, st, scal, S, nS, out_scal, out_S, flag, seq);
  return hipGetLastError();
}

// out[0] = v . v
__global__ __launch_bounds__(1024) void k_sumsq(const double* __restrict__ v, int N, double* __restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < N; i += blockDim.x) s = __builtin_fma(v[i], v[i], s);
  s = block_sum_1024(s, red);
  if (threadIdx.x == 0) out[0] = s;
}
hipError_t launch_sumsq(const double* v, int N, double* out, hipStream_t st) {
  hipLaunchKernelGGL(k_sumsq, dim3(1), 1024, 0, st, v, N, out);
  return hipGetLastError();
}

// out[0] = trace(Rinv), out[1] = gamma . gamma, out[2] = qv . qv (0 without qv)   (the sigma2 derivative of the NOISY likelihood, gpr.py:1030-1036)
__global__ __launch_bounds__(1024) void k_trace_gg(const double* __restrict__ Rinv, int ld, int nparts, size_t part_stride, int N,
                                                   const double* __restrict__ gamma, const double* __restrict__ qv,
                                                   double* __restrict__ out) {
  __shared__ double red[16];
  double tr = 0.0, gg = 0.0, qq = 0.0;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    for (int q = 0; q < nparts; ++q) tr += Rinv[q * part_stride + (size_t)i * ld + i];
    gg = __builtin_fma(gamma[i], gamma[i], gg);
    if (qv) qq = __builtin_fma(qv[i], qv[i], qq);
  }
  tr = block_sum_1024(tr, red);
  gg = block_sum_1024(gg, red);
  qq = block_sum_1024(qq, red);
  if (threadIdx.x == 0) {
    out[0] = tr;
    out[1] = gg;
    out[2] = qq;
  }
}
hipError_t launch_trace_gg(const double* Rinv, int ld, int nparts, size_t part_stride, int N, const double* gamma,
                           const double* qv, double* out, hipStream_t st) {
  hipLaunchKernelGGL(k_trace_gg, dim3(1), 1024, 0, st, Rinv, ld, nparts, part_stride, N, gamma, qv, out);
  return hipGetLastError();
}

// ---- polynomial trend bases (surrogate/gaussian_process/trend.py:66-142) ----------------------------------------
// column order of F: constant [1]; linear [1, x_0 .. x_{d-1}] (:104-107); quadratic [1, x, then for k = 0..d-1:
// x_k x_j, j = k..d-1] (:130-136).  `emit(col, value)` is called once per column, in order.
template <typename Emit>
__device__ __forceinline__ void trend_basis(int trend, const double* __restrict__ x, int d, Emit emit) {
  emit(0, 1.0);
  if (trend == BOGP_TREND_CONSTANT) return;
  for (int k = 0; k < d; ++k) emit(1 + k, x[k]);
  if (trend == BOGP_TREND_LINEAR) return;
  int col = 1 + d;
  for (int k = 0; k < d; ++k) {
    const double xk = x[k];
    for (int j = k; j < d; ++j) emit(col++, xk * x[j]);
  }
}

// F (N x p, column-major, ld = N) at the training points
__global__ void k_trend_train(int trend, const double* __restrict__ X, int N, int d, double* __restrict__ F) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  trend_basis(trend, X + (size_t)n * d, d, [&](int col, double v) { F[(size_t)col * N + n] = v; });
}
hipError_t launch_trend_train(int trend, const double* X, int N, int d, double* F, hipStream_t st) {
  hipLaunchKernelGGL(k_trend_train, dim3((N + 255) / 256), 256, 0, st, trend, X, N, d, F);
  return hipGetLastError();
}

// per candidate of a chunk: mtrend = f(x*) . beta  and, when T is given (universal kriging), T <- T - f(x*) in place
// (T is Mc x p column-major: T(m, col) = (Ft^T L^-1 r)_col, gpr.py:496-498 before the G solve)
__global__ void k_trend_terms(int trend, const double* __restrict__ Xs, int64_t m0, int64_t mcount, int d, int64_t Mc,
                              const double* __restrict__ beta, double* __restrict__ T, double* __restrict__ mtrend) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= mcount) return;
  double acc = 0.0;
  trend_basis(trend, Xs + (size_t)(m0 + i) * d, d, [&](int col, double v) {
    acc = __builtin_fma(v, beta[col], acc);
    if (T) T[(size_t)col * Mc + i] -= v;
  });
  mtrend[i] = acc;
}
hipError_t launch_trend_terms(int trend, const double* Xs, int64_t m0, int64_t mcount, int d, int64_t Mc, const double* beta,
                              double* T, double* mtrend, hipStream_t st) {
  hipLaunchKernelGGL(k_trend_terms, dim3((unsigned)((mcount + 255) / 256)), 256, 0, st, trend, Xs, m0, mcount, d, Mc, beta, T, mtrend);
  return hipGetLastError();
}

// the extended chunk rows of the trend-rows path (k_pack_Vx): Rext = rT + Ne * Mc holds rows Ne .. Ne + prows - 1 of the chunk:
// row col < p = -f_col(x*_i), rows p .. prows - 1 = 0; candidates beyond mcount (the chunk's tail tile) get zeros; mtrend = f(x*) . beta
__global__ void k_trend_rows(int trend, const double* __restrict__ Xs, int64_t m0, int64_t mcount, int64_t mrows, int d, int64_t Mc,
                             const double* __restrict__ beta, double* __restrict__ Rext, int p, int prows, double* __restrict__ mtrend) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= mrows) return;
  if (i < mcount) {
    double acc = 0.0;
    trend_basis(trend, Xs + (size_t)(m0 + i) * d, d, [&](int col, double v) {
      acc = __builtin_fma(v, beta[col], acc);
      Rext[(size_t)col * Mc + i] = -v;
    });
    mtrend[i] = acc;
  } else {
    for (int col = 0; col < p; ++col) Rext[(size_t)col * Mc + i] = 0.0;
  }
  for (int col = p; col < prows; ++col) Rext[(size_t)col * Mc + i] = 0.0;
}
hipError_t launch_trend_rows(int trend, const double* Xs, int64_t m0, int64_t mcount, int64_t mrows, int d, int64_t Mc, const double* beta,
                             double* Rext, int p, int prows, double* mtrend, hipStream_t st) {
  hipLaunchKernelGGL(k_trend_rows, dim3((unsigned)((mrows + 255) / 256)), 256, 0, st, trend, Xs, m0, mcount, mrows, d, Mc, beta, Rext, p, prows, mtrend);
  return hipGetLastError();
}

// Universal kriging with a SMALL polynomial basis (p <= 32: a linear trend up to d = 31), everything after the fused producer in one
// launch, one thread a candidate: T = the slice sums of W^T r added in slice order, c = T - f(x*) (gpr.py:496-498 before the G
// solve), mtrend = f(x*) . beta (trend.py:34-37), uu = c^T (Ft^T Ft)^-1 c = u^T u.  c lives in LDS ([col][thread]: conflict-free).
__global__ __launch_bounds__(256) void k_trend_small(int trend, const double* __restrict__ Xs, int64_t m0, int64_t mcount, int d, int64_t Mc,
                                                     const double* __restrict__ beta, const double* __restrict__ t_part, int S, int pv, int p,
                                                     const double* __restrict__ Sinv, double* __restrict__ mtrend, double* __restrict__ uu) {
  __shared__ double cs[32 * 256];
  const int tid = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + tid;
  if (i >= mcount) return;
  for (int col = 0; col < p; ++col) {
    double t = 0.0;
    for (int s = 0; s < S; ++s) t += t_part[((size_t)s * pv + col) * Mc + i];
    cs[col * 256 + tid] = t;
  }
  double acc = 0.0;
  trend_basis(trend, Xs + (size_t)(m0 + i) * d, d, [&](int col, double v) {
    acc = __builtin_fma(v, beta[col], acc);
    cs[col * 256 + tid] -= v;
  });
  mtrend[i] = acc;
  double q = 0.0;
  for (int a = 0; a < p; ++a) {
    double row = 0.0;
    for (int b = 0; b < p; ++b) row = __builtin_fma(Sinv[(size_t)b * p + a], cs[b * 256 + tid], row);  // column-major p x p (symmetric)
    q = __builtin_fma(cs[a * 256 + tid], row, q);
  }
  uu[i] = q;
}
hipError_t launch_trend_small(int trend, const double* Xs, int64_t m0, int64_t mcount, int d, int64_t Mc, const double* beta, const double* t_part,
                              int S, int pv, int p, const double* Sinv, double* mtrend, double* uu, hipStream_t st) {
  if (p > 32) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_trend_small, dim3((unsigned)((mcount + 255) / 256)), 256, 0, st, trend, Xs, m0, mcount, d, Mc, beta, t_part, S, pv, p,
                     Sinv, mtrend, uu);
  return hipGetLastError();
}

// uu[m] = sum_col C(m, col) * CS(m, col)   (= u^T u with u = G^-T c, because CS = C (G^T G)^-1)
__global__ void k_rowdot(const double* __restrict__ Cm, const double* __restrict__ CS, int64_t Mc, int64_t mcount, int p,
                         double* __restrict__ uu) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= mcount) return;
  double s = 0.0;
  for (int col = 0; col < p; ++col) s = __builtin_fma(Cm[(size_t)col * Mc + i], CS[(size_t)col * Mc + i], s);
  uu[i] = s;
}
hipError_t launch_rowdot(const double* Cm, const double* CS, int64_t Mc, int64_t mcount, int p, double* uu, hipStream_t st) {
  hipLaunchKernelGGL(k_rowdot, dim3((unsigned)((mcount + 255) / 256)), 256, 0, st, Cm, CS, Mc, mcount, p, uu);
  return hipGetLastError();
}

// dst (row-major N x N, strict upper = 0)  <-  lower triangle of column-major L
__global__ void k_copy_lower(const double* __restrict__ L, int N, int ld, double* __restrict__ dst) {
  const int j = blockIdx.x * 16 + (threadIdx.x & 15);  // column
  const int i = blockIdx.y * 16 + (threadIdx.x >> 4);  // row
  if (i >= N || j >= N) return;
  dst[(size_t)i * N + j] = j <= i ? L[(size_t)j * ld + i] : 0.0;
}
hipError_t launch_copy_lower(const double* L, int N, int ld, double* dst, hipStream_t st) {
  hipLaunchKernelGGL(k_copy_lower, dim3((N + 15) / 16, (N + 15) / 16), 256, 0, st, L, N, ld, dst);
  return hipGetLastError();
}

// out[k] = sum_blk partial[blk][k]  (fixed order: deterministic)
__global__ __launch_bounds__(256) void k_grad_reduce(const double* __restrict__ partial, int nblk, int nout, double* out) {
  __shared__ double red[256];
  const int k = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) s += partial[(size_t)b * nout + k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[k] = red[0];
}
hipError_t launch_grad_reduce(const double* partial, int nblk, int nout, double* out, hipStream_t st) {
  hipLaunchKernelGGL(k_grad_reduce, dim3(nout), 256, 0, st, partial, nblk, nout, out);
  return hipGetLastError();
}

// The tail of a likelihood gradient in ONE launch: k_grad_reduce's column sums (workgroup k: out[k], the same order of additions),
// then the LAST workgroup to finish (a ticket) adds what k_trace_gg computes -- out[nout] = trace(R^-1), out[nout + 1] = gamma.gamma
// (noisy mode) -- and does k_fit_gather's job: the 64 scalars and the nout + 2 sums into device-mapped pinned memory + the
// sequence word the host polls.  Three launches less per evaluation on the general path.
// (returns true, in every thread of the finishing workgroup only, once its record is fenced to the system)
__device__ __forceinline__ bool grad_finish_block(const double* __restrict__ partial, int nblk, int nout, double* __restrict__ out,
                                                  const double* __restrict__ Rinv, int ld, int nparts, size_t part_stride, int N,
                                                  const double* __restrict__ gamma, int with_trace, const double* __restrict__ scal,
                                                  double* __restrict__ out_scal, double* __restrict__ out_S,
                                                  unsigned int* __restrict__ ticket) {
  __shared__ double red[256];
  __shared__ int s_last;
  const int k = blockIdx.x, tid = threadIdx.x;
  double s = 0.0;
  for (int b = tid; b < nblk; b += 256) s += partial[(size_t)b * nout + k];
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    __hip_atomic_store(&out[k], red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    s_last = atomicAdd(ticket, 1u) == (unsigned int)(nout - 1);
  }
  __syncthreads();
  if (!s_last) return false;
  __threadfence();
  double tr = 0.0, gg = 0.0;
  if (with_trace) {
    for (int i = tid; i < N; i += 256) {
      for (int q = 0; q < nparts; ++q) tr += Rinv[q * part_stride + (size_t)i * ld + i];
      gg = __builtin_fma(gamma[i], gamma[i], gg);
    }
    red[tid] = tr;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    tr = red[0];
    __syncthreads();
    red[tid] = gg;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    gg = red[0];
  }
  if (tid < 64) out_scal[tid] = scal[tid];
  for (int i = tid; i < nout; i += 256) out_S[i] = __hip_atomic_load(&out[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) {
    out_S[nout] = tr;
    out_S[nout + 1] = gg;
    out[nout] = tr;
    out[nout + 1] = gg;
    *ticket = 0u;
  }
  __threadfence_system();
  __syncthreads();
  return true;
}
__global__ __launch_bounds__(256) void k_grad_finish(const double* __restrict__ partial, int nblk, int nout, double* __restrict__ out,
                                                     const double* __restrict__ Rinv, int ld, int nparts, size_t part_stride, int N,
                                                     const double* __restrict__ gamma, int with_trace, const double* __restrict__ scal,
                                                     double* __restrict__ out_scal, double* __restrict__ out_S,
                                                     unsigned long long* __restrict__ flag, unsigned long long seq,
                                                     unsigned int* __restrict__ ticket) {
  if (grad_finish_block(partial, nblk, nout, out, Rinv, ld, nparts, part_stride, N, gamma, with_trace, scal, out_scal, out_S, ticket) &&
      threadIdx.x == 0)
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// bogp_nll_batch: blockIdx.y = the parameter vector; slot s leaves its record in bout + s * bout_stride (64 scalars, then the
// nout + 2 sums) and the last SLOT to finish (a second ticket) publishes the sequence word
__global__ __launch_bounds__(256) void k_grad_finish_b(const BatchSlot* __restrict__ slots, int nblk, int nout, int ld, int N, int with_trace,
                                                       double* __restrict__ bout, int bout_stride, unsigned long long* __restrict__ flag,
                                                       unsigned long long seq, unsigned int* __restrict__ gticket) {
  const BatchSlot& sl = slots[blockIdx.y];
  double* rec = bout + (size_t)blockIdx.y * bout_stride;
  if (grad_finish_block(sl.partial, nblk, nout, sl.S, sl.Rinv, ld, 1, (size_t)ld * ld, N, sl.gamma, with_trace, sl.scal, rec, rec + 64, sl.ticket) &&
      threadIdx.x == 0) {
    if (atomicAdd(gticket, 1u) == gridDim.y - 1) {
      *gticket = 0u;
      __threadfence_system();
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// value-only batches: slot s's 64 scalars into its record, the last slot publishes (k_fit_gather's job for P evaluations)
__global__ __launch_bounds__(64) void k_fit_gather_b(const BatchSlot* __restrict__ slots, double* __restrict__ bout, int bout_stride,
                                                     unsigned long long* __restrict__ flag, unsigned long long seq,
                                                     unsigned int* __restrict__ gticket) {
  const BatchSlot& sl = slots[blockIdx.x];
  double* rec = bout + (size_t)blockIdx.x * bout_stride;
  rec[threadIdx.x] = sl.scal[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(gticket, 1u) == gridDim.x - 1) {
    *gticket = 0u;
    __threadfence_system();
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
hipError_t launch_grad_finish_batch(const BatchSlot* slots, int P, int nblk, int nout, int ld, int N, int with_trace, double* bout,
                                    int bout_stride, unsigned long long* flag, unsigned long long seq, unsigned int* gticket, hipStream_t st) {
  hipLaunchKernelGGL(k_grad_finish_b, dim3(nout, P), 256, 0, st, slots, nblk, nout, ld, N, with_trace, bout, bout_stride, flag, seq, gticket);
  return hipGetLastError();
}
hipError_t launch_fit_gather_batch(const BatchSlot* slots, int P, double* bout, int bout_stride, unsigned long long* flag,
                                   unsigned long long seq, unsigned int* gticket, hipStream_t st) {
  hipLaunchKernelGGL(k_fit_gather_b, dim3(P), 64, 0, st, slots, bout, bout_stride, flag, seq, gticket);
  return hipGetLastError();
}
hipError_t launch_grad_finish(const double* partial, int nblk, int nout, double* out, const double* Rinv, int ld, int nparts,
                              size_t part_stride, int N, const double* gamma, int with_trace, const double* scal, double* out_scal,
                              double* out_S, unsigned long long* flag, unsigned long long seq, unsigned int* ticket, hipStream_t st) {
  hipLaunchKernelGGL(k_grad_finish, dim3(nout), 256, 0, st, partial, nblk, nout, out, Rinv, ld, nparts, part_stride, N, gamma,
                     with_trace, scal, out_scal, out_S, flag, seq, ticket);
  return hipGetLastError();
}

// r[n] = corr(theta, |x - X_n|), rdx[k*N + n] = d r[n] / d x_k     (corr_dx, gpr.py:600-661)
//   SE: -2 r theta_k diff_k (:635-636);  Matern-3/2: diff theta / D * (-3 D exp(-sqrt3 D)) (:642-644), the 0/0
//   at D = 0 is a warning the reference turns into an all-zero gradient (:658-659) -- here the affected
//   entries are 0 (their limit);  Matern-5/2 / 1/2: analytic forms (extensions, the reference has `pass`).
template <int KERNEL>
__global__ void k_point_corr(const double* __restrict__ X, int N, int d, const double* __restrict__ theta,
                             const double* __restrict__ x, double* __restrict__ r, double* __restrict__ rdx) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double s2 = 0.0;
  for (int k = 0; k < d; ++k) {
    s2 += dist_term<KERNEL>(theta[k], fabs(x[k] - X[(size_t)n * d + k]));
  }
  const double rv = corr_profile<KERNEL>(s2);
  r[n] = rv;
  const double D = sqrt(s2);
  for (int k = 0; k < d; ++k) {
    const double diff = x[k] - X[(size_t)n * d + k];
    double g;
    if (KERNEL == BOGP_KERNEL_ABSEXP) {
      g = -1.0 * rv * theta[k] * (diff > 0.0 ? 1.0 : (diff < 0.0 ? -1.0 : 0.0));  // -r theta sign(diff) (gpr.py:650-651)
    } else if (KERNEL == BOGP_KERNEL_SE) {
      g = -2 * rv * (theta[k] * diff);
    } else if (KERNEL == BOGP_KERNEL_MATERN32) {
      g = D > 0.0 ? (diff * theta[k] / D) * (-3.0 * D * exp(-1.7320508075688772 * D)) : 0.0;
    } else if (KERNEL == BOGP_KERNEL_MATERN52) {
      g = (-(5.0 / 3.0) * (1.0 + 2.23606797749979 * D) * exp(-2.23606797749979 * D)) * (theta[k] * diff);
    } else {
      g = D > 0.0 ? -diff * theta[k] / D * rv : 0.0;
    }
    rdx[(size_t)k * N + n] = g;
  }
}
hipError_t launch_point_corr(int kernel, const double* X, int N, int d, const double* theta, const double* x,
                             double* r, double* rdx, hipStream_t st) {
  dim3 grid((N + 255) / 256);
  switch (kernel) {
    case BOGP_KERNEL_SE: hipLaunchKernelGGL(k_point_corr<BOGP_KERNEL_SE>, grid, 256, 0, st, X, N, d, theta, x, r, rdx); break;
    case BOGP_KERNEL_MATERN12: hipLaunchKernelGGL(k_point_corr<BOGP_KERNEL_MATERN12>, grid, 256, 0, st, X, N, d, theta, x, r, rdx); break;
    case BOGP_KERNEL_MATERN32: hipLaunchKernelGGL(k_point_corr<BOGP_KERNEL_MATERN32>, grid, 256, 0, st, X, N, d, theta, x, r, rdx); break;
    case BOGP_KERNEL_ABSEXP: hipLaunchKernelGGL(k_point_corr<BOGP_KERNEL_ABSEXP>, grid, 256, 0, st, X, N, d, theta, x, r, rdx); break;
    default: hipLaunchKernelGGL(k_point_corr<BOGP_KERNEL_MATERN52>, grid, 256, 0, st, X, N, d, theta, x, r, rdx); break;
  }
  return hipGetLastError();
}

// ---- Hessian of the posterior mean at one point (GaussianProcess.Hessian, gpr.py:578-598; squared exponential only:
// corr_Hessian :663-734 leaves H undefined for the other kernels) -------------------------------------------------------
// H[k][l] = sum_n gamma_n * (-2) * (g[k][n] * theta_l (x_l - X_nl) + [k == l] r_n theta_k),  g = dr/dx from k_point_corr
__global__ __launch_bounds__(256) void k_point_hessian(const double* __restrict__ X, int N, int d, const double* __restrict__ theta,
                                                       const double* __restrict__ x, const double* __restrict__ r,
                                                       const double* __restrict__ rdx, const double* __restrict__ gamma,
                                                       double* __restrict__ H) {
  __shared__ double red[4];
  const int k = blockIdx.y, l = blockIdx.x;
  const double tl = theta[l], tk = theta[k], xl = x[l];
  double acc = 0.0;
  for (int n = threadIdx.x; n < N; n += 256) {
    double v = rdx[(size_t)k * N + n] * (tl * (xl - X[(size_t)n * d + l]));
    if (k == l) v += r[n] * tk;
    acc = __builtin_fma(gamma[n], -2.0 * v, acc);
  }
  for (int m = 32; m >= 1; m >>= 1) acc += shfl_xor_f64(acc, m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) H[(size_t)k * d + l] = ((red[0] + red[1]) + red[2]) + red[3];
}
hipError_t launch_point_hessian(const double* X, int N, int d, const double* theta, const double* x, const double* r,
                                const double* rdx, const double* gamma, double* H, hipStream_t st) {
  hipLaunchKernelGGL(k_point_hessian, dim3(d, d), 256, 0, st, X, N, d, theta, x, r, rdx, gamma, H);
  return hipGetLastError();
}

// ---- batched input-gradients: B points at once ----------------------------------------------
// k_batch_corr:  r[b*N + n] = corr(theta, |x_b - X_n|),  s2[b*N + n] = the weighted distance it was computed from
template <int KERNEL>
__global__ __launch_bounds__(256) void k_batch_corr(const double* __restrict__ X, int N, int d,
                                                    const double* __restrict__ theta, const double* __restrict__ Xb,
                                                    double* __restrict__ r, double* __restrict__ s2out) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (n >= N) return;
  double s2 = dist_init<KERNEL>();
  const double pexp = kernel_exponent<KERNEL>(theta, d);
  for (int k = 0; k < d; ++k) s2 = dist_fold<KERNEL>(theta[k], Xb[(size_t)b * d + k] - X[(size_t)n * d + k], s2, pexp);
  r[(size_t)b * N + n] = corr_profile<KERNEL>(s2, pexp);
  s2out[(size_t)b * N + n] = s2;
}

// d r / d x_k of one (point, training row) pair; same formulas as k_point_corr (corr_dx, gpr.py:600-661)
template <int KERNEL>
__device__ __forceinline__ double corr_dx_entry(double rv, double s2, double theta_k, double diff) {
  if (KERNEL == BOGP_KERNEL_ABSEXP) return -1.0 * rv * theta_k * (diff > 0.0 ? 1.0 : (diff < 0.0 ? -1.0 : 0.0));
  if (KERNEL == BOGP_KERNEL_SE) return -2 * rv * (theta_k * diff);
  const double D = sqrt(s2);
  if (KERNEL == BOGP_KERNEL_MATERN32) return D > 0.0 ? (diff * theta_k / D) * (-3.0 * D * exp(-1.7320508075688772 * D)) : 0.0;
  if (KERNEL == BOGP_KERNEL_MATERN52)
    return (-(5.0 / 3.0) * (1.0 + 2.23606797749979 * D) * exp(-2.23606797749979 * D)) * (theta_k * diff);
  return D > 0.0 ? -diff * theta_k / D * rv : 0.0;
}

// One workgroup per point b.  out[b][0..d) = sum_n gamma_n dr_n/dx_k, [d..2d) = sum_n z_n dr_n/dx_k,
// [2d..3d) = sum_n w_n dr_n/dx_k, [3d] = sum_n w_n r_n   (z = R^-1 r, column b of Z)
template <int KERNEL>
__global__ __launch_bounds__(256) void k_batch_grad(const double* __restrict__ X, int N, int d,
                                                    const double* __restrict__ theta, const double* __restrict__ Xb,
                                                    const double* __restrict__ r, const double* __restrict__ s2,
                                                    const double* __restrict__ Z, const double* __restrict__ gamma,
                                                    const double* __restrict__ wvec, double* __restrict__ out) {
  __shared__ double red[3][4];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double* rb = r + (size_t)b * N;
  const double* sb = s2 + (size_t)b * N;
  const double* zb = Z + (size_t)b * N;
  double* ob = out + (size_t)b * (3 * d + 1);
  for (int k = 0; k <= d; ++k) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) {
      if (k < d) {
        const double g = corr_dx_entry<KERNEL>(rb[n], sb[n], theta[k], Xb[(size_t)b * d + k] - X[(size_t)n * d + k]);
        a0 += gamma[n] * g;
        a1 += zb[n] * g;
        a2 += wvec[n] * g;
      } else {
        a0 += wvec[n] * rb[n];
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      a0 += shfl_xor_f64(a0, off);
      a1 += shfl_xor_f64(a1, off);
      a2 += shfl_xor_f64(a2, off);
    }
    if (lane == 0) {
      red[0][wv] = a0;
      red[1][wv] = a1;
      red[2][wv] = a2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const double t0 = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
      const double t1 = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
      const double t2 = ((red[2][0] + red[2][1]) + red[2][2]) + red[2][3];
      if (k < d) {
        ob[k] = t0;
        ob[d + k] = t1;
        ob[2 * d + k] = t2;
      } else {
        ob[3 * d] = t0;
      }
    }
    __syncthreads();
  }
}

// Column reductions of the small-batch posterior path: for column b of r (N x B) and rt = V r (N x B):
// mu[b] = sum_n r gamma, wd[b] = sum_n r w, ss[b] = sum_n rt^2   (one workgroup per column, fixed order)
__global__ __launch_bounds__(256) void k_col_reduce(const double* __restrict__ r, const double* __restrict__ rt, int N,
                                                    const double* __restrict__ gamma, const double* __restrict__ wvec,
                                                    double* __restrict__ mu, double* __restrict__ wd, double* __restrict__ ss) {
  __shared__ double red[3][4];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  for (int n = threadIdx.x; n < N; n += 256) {
    const double rv = r[(size_t)b * N + n], tv = rt[(size_t)b * N + n];
    a0 = __builtin_fma(rv, gamma[n], a0);
    a1 = __builtin_fma(rv, wvec[n], a1);
    a2 = __builtin_fma(tv, tv, a2);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    a0 += shfl_xor_f64(a0, off);
    a1 += shfl_xor_f64(a1, off);
    a2 += shfl_xor_f64(a2, off);
  }
  if (lane == 0) {
    red[0][wv] = a0;
    red[1][wv] = a1;
    red[2][wv] = a2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    mu[b] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    wd[b] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    ss[b] = ((red[2][0] + red[2][1]) + red[2][2]) + red[2][3];
  }
}
hipError_t launch_col_reduce(const double* r, const double* rt, int N, int B, const double* gamma, const double* wvec,
                             double* mu, double* wd, double* ss, hipStream_t st) {
  hipLaunchKernelGGL(k_col_reduce, dim3(B), 256, 0, st, r, rt, N, gamma, wvec, mu, wd, ss);
  return hipGetLastError();
}

#define BOGP_DISPATCH_KERNEL(kernel, CALL)                       \
  switch (kernel) {                                              \
    case BOGP_KERNEL_SE: CALL(BOGP_KERNEL_SE); break;            \
    case BOGP_KERNEL_MATERN12: CALL(BOGP_KERNEL_MATERN12); break; \
    case BOGP_KERNEL_MATERN32: CALL(BOGP_KERNEL_MATERN32); break; \
    case BOGP_KERNEL_ABSEXP: CALL(BOGP_KERNEL_ABSEXP); break;    \
    default: CALL(BOGP_KERNEL_MATERN52); break;                  \
  }

hipError_t launch_batch_corr(int kernel, const double* X, int N, int d, const double* theta, const double* Xb, int B,
                             double* r, double* s2, hipStream_t st) {
  dim3 grid((N + 255) / 256, B);
#define CALL(K) hipLaunchKernelGGL(k_batch_corr<K>, grid, 256, 0, st, X, N, d, theta, Xb, r, s2)
  if (kernel == BOGP_KERNEL_MATERN_NU) {
    CALL(BOGP_KERNEL_MATERN_NU);
    return hipGetLastError();
  }
  if (kernel == BOGP_KERNEL_CUBIC || kernel == BOGP_KERNEL_GENEXP) {  // values only (small-batch posterior, prior correlation): no derivative kernels
    if (kernel == BOGP_KERNEL_CUBIC) {
      CALL(BOGP_KERNEL_CUBIC);
    } else {
      CALL(BOGP_KERNEL_GENEXP);
    }
    return hipGetLastError();
  }
  BOGP_DISPATCH_KERNEL(kernel, CALL)
#undef CALL
  return hipGetLastError();
}

hipError_t launch_batch_grad(int kernel, const double* X, int N, int d, const double* theta, const double* Xb, int B,
                             const double* r, const double* s2, const double* Z, const double* gamma,
                             const double* wvec, double* out, hipStream_t st) {
#define CALL(K) hipLaunchKernelGGL(k_batch_grad<K>, dim3(B), 256, 0, st, X, N, d, theta, Xb, r, s2, Z, gamma, wvec, out)
  BOGP_DISPATCH_KERNEL(kernel, CALL)
#undef CALL
  return hipGetLastError();
}

}  // namespace bogp
