// kernels_acq.hip -- posterior finalisation + acquisition criteria + wavefront-reduced argmax (gfx950).
//
// Replaces, per candidate row (the reference evaluates ONE row per call and raises on batches for
// EI/EpsilonPI/MGFI -- SURVEY.md 8a; row i here is exactly what the single-row call returns):
//   gpr.py:490         mu  = mean(X*) + r.gamma            (sum of the slice partials + beta)
//   gpr.py:496-510     MSE = (1 - sum rt^2 + sum u^2) sigma2, negatives -> 0;  u = (w.r - 1)/G (constant trend)
//   acquisition_fun.py:52-64   y_hat = +-mu, sd = sqrt(MSE)
//   acquisition_fun.py:153-176 EI  (guard sd/sqrt(sigma2) < 1e-6 -> 0)
//   acquisition_fun.py:208-217 EpsilonPI
//   acquisition_fun.py:127-135 UCB
//   acquisition_fun.py:265-290 MGFI (t <= 22.36; guard isclose(sd, 0) -> 0; overflow/inf -> 0)
// and np.argmax over the rows (first maximum; NaN is maximal).  HBM-bound: ~8 (S + nJ + 2) bytes per candidate.
#include "bogp_device.h"
#include "bogp_internal.h"

namespace bogp {

__device__ __forceinline__ double acq_value(int id, double par, double y_hat, double sd, double plugin, double sigma2) {
  switch (id) {
    case BOGP_ACQ_EI: {
      if (sd / sqrt(sigma2) < 1e-6) return 0.0;
      const double xcr_ = plugin - y_hat;
      const double xcr = xcr_ / sd;
      return xcr_ * ndtr(xcr) + sd * norm_pdf(xcr);
    }
    case BOGP_ACQ_EPSILON_PI: {
      const double coef = y_hat > 0 ? 1 - par : 1 + par;
      return ndtr((plugin - coef * y_hat) / sd);
    }
    case BOGP_ACQ_UCB: return y_hat + par * sd;
    default: {  // MGFI
      const double t = fmin(par, 22.36);
      if (fabs(sd) <= 1e-8) return 0.0;  // np.isclose(sd, 0)
      const double sd2 = sd * sd;
      const double y_hat_p = y_hat - t * sd2;
      const double beta_p = (plugin - y_hat_p) / sd;
      const double term = t * (plugin - y_hat - 1);
      const double e = exp(term + (t * t) * sd2 / 2.0);
      const double f = ndtr(beta_p) * e;
      return (isfinite(e) && isfinite(f)) ? f : 0.0;
    }
  }
}

__global__ __launch_bounds__(256) void k_acquisition(AcqArgs a) {
  __shared__ double sv[4];
  __shared__ int64_t si[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // row inside the chunk
  const bool valid = i < a.mcount;
  double y_hat = 0.0, sd = 0.0;
  if (valid) {
    double mu = 0.0, wd = 0.0, ss = 0.0;
    for (int s = 0; s < a.S; ++s) {
      mu += a.mu_part[(size_t)s * a.Mc + i];
      wd += a.w_part[(size_t)s * a.Mc + i];
    }
    for (int j = 0; j < a.nJ; ++j) ss += a.ss_part[(size_t)j * a.Mc + i];
    mu = a.beta + mu;
    double u2 = 0.0;
    if (a.estimate_trend) {
      const double u = (wd - 1.0) / a.G;
      u2 = u * u;
    }
    double mse = (1.0 - ss + u2) * a.sigma2;
    if (mse < 0.0) mse = 0.0;
    if (a.mu_out) a.mu_out[a.m0 + i] = mu;
    if (a.mse_out) a.mse_out[a.m0 + i] = mse;
    y_hat = a.minimize ? mu : -1 * mu;
    sd = sqrt(mse);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int c = 0; c < a.q; ++c) {
    double v = -INFINITY;
    int64_t idx = INT64_MAX;
    if (valid) {
      v = acq_value(a.acq_id[c], a.acq_par[c], y_hat, sd, a.plugin, a.sigma2);
      idx = a.m0 + i;
      if (a.acq_out) a.acq_out[(size_t)c * a.M + idx] = v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double ov = shfl_xor_f64(v, off);
      const int64_t oi = shfl_xor_i64(idx, off);
      if (better(ov, oi, v, idx)) {
        v = ov;
        idx = oi;
      }
    }
    if (lane == 0) {
      sv[w] = v;
      si[w] = idx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int k = 1; k < 4; ++k)
        if (better(sv[k], si[k], v, idx)) {
          v = sv[k];
          idx = si[k];
        }
      a.blk_val[(size_t)c * a.nblk_total + a.blk_offset + blockIdx.x] = v;
      a.blk_idx[(size_t)c * a.nblk_total + a.blk_offset + blockIdx.x] = idx;
    }
    __syncthreads();
  }
}

// one workgroup per criterion: reduce the per-block partials (deterministic, index tie-break)
__global__ __launch_bounds__(256) void k_argmax_final(const double* blk_val, const int64_t* blk_idx, int64_t nblk,
                                                      int64_t stride, double* out_val, int64_t* out_idx) {
  __shared__ double sv[4];
  __shared__ int64_t si[4];
  const int c = blockIdx.x;
  double v = -INFINITY;
  int64_t idx = INT64_MAX;
  for (int64_t k = threadIdx.x; k < nblk; k += 256) {
    const double ov = blk_val[(size_t)c * stride + k];
    const int64_t oi = blk_idx[(size_t)c * stride + k];
    if (better(ov, oi, v, idx)) {
      v = ov;
      idx = oi;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double ov = shfl_xor_f64(v, off);
    const int64_t oi = shfl_xor_i64(idx, off);
    if (better(ov, oi, v, idx)) {
      v = ov;
      idx = oi;
    }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    sv[w] = v;
    si[w] = idx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k)
      if (better(sv[k], si[k], v, idx)) {
        v = sv[k];
        idx = si[k];
      }
    out_val[c] = v;
    out_idx[c] = idx;
  }
}

// top-k support: per-block argmax over stored acquisition values of criterion c, skipping indices already taken
struct ExclArgs {
  int n;
  int64_t idx[BOGP_MAX_TOPK];
};
__global__ __launch_bounds__(256) void k_block_argmax_excl(const double* __restrict__ vals, int64_t M, ExclArgs ex,
                                                           double* __restrict__ blk_val, int64_t* __restrict__ blk_idx) {
  __shared__ double sv[4];
  __shared__ int64_t si[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double v = -INFINITY;
  int64_t idx = INT64_MAX;
  if (i < M) {
    bool taken = false;
    for (int e = 0; e < ex.n; ++e) taken |= (ex.idx[e] == i);
    if (!taken) {
      v = vals[i];
      idx = i;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double ov = shfl_xor_f64(v, off);
    const int64_t oi = shfl_xor_i64(idx, off);
    if (better(ov, oi, v, idx)) {
      v = ov;
      idx = oi;
    }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    sv[w] = v;
    si[w] = idx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k)
      if (better(sv[k], si[k], v, idx)) {
        v = sv[k];
        idx = si[k];
      }
    blk_val[blockIdx.x] = v;
    blk_idx[blockIdx.x] = idx;
  }
}

hipError_t launch_block_argmax_excl(const double* vals, int64_t M, const int64_t* excl, int nexcl, double* blk_val,
                                    int64_t* blk_idx, hipStream_t st) {
  ExclArgs ex;
  ex.n = nexcl;
  for (int e = 0; e < nexcl; ++e) ex.idx[e] = excl[e];
  hipLaunchKernelGGL(k_block_argmax_excl, dim3((unsigned)((M + 255) / 256)), 256, 0, st, vals, M, ex, blk_val, blk_idx);
  return hipGetLastError();
}

hipError_t launch_acquisition(const AcqArgs& a, hipStream_t st) {
  const unsigned nblk = (unsigned)((a.mcount + 255) / 256);
  hipLaunchKernelGGL(k_acquisition, dim3(nblk), 256, 0, st, a);
  return hipGetLastError();
}

hipError_t launch_argmax_final(const double* blk_val, const int64_t* blk_idx, int64_t nblk, int64_t stride, int q,
                               double* out_val, int64_t* out_idx, hipStream_t st) {
  hipLaunchKernelGGL(k_argmax_final, dim3(q), 256, 0, st, blk_val, blk_idx, nblk, stride, out_val, out_idx);
  return hipGetLastError();
}

}  // namespace bogp
