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
    mu = (a.mtrend ? a.mtrend[i] : a.beta) + mu;
    double u2 = 0.0;
    if (a.nJ_plus > 0) {
      for (int j = a.nJ; j < a.nJ + a.nJ_plus; ++j) u2 += a.ss_part[(size_t)j * a.Mc + i];
    } else if (a.uu) {
      u2 = a.uu[i];
    } else if (a.estimate_trend) {
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
                                                      int64_t stride, double* out_val, int64_t* out_idx, int out_stride,
                                                      int out_off) {
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
    out_val[(size_t)c * out_stride + out_off] = v;
    out_idx[(size_t)c * out_stride + out_off] = idx;
  }
}

// top-k support: rank r of all q criteria in one launch (grid = blocks x q): per-block argmax over the stored criterion
// values, skipping the r winners so far, which are read from DEVICE memory (taken[c][0..r)) -- so the k ranks are k queued
// launch pairs with no host round trip between them.
__global__ __launch_bounds__(256) void k_block_argmax_excl(const double* __restrict__ vals, int64_t M,
                                                           const int64_t* __restrict__ taken, int kstride, int r,
                                                           double* __restrict__ blk_val, int64_t* __restrict__ blk_idx,
                                                           int64_t nblk) {
  __shared__ double sv[4];
  __shared__ int64_t si[4];
  const int c = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double v = -INFINITY;
  int64_t idx = INT64_MAX;
  if (i < M) {
    bool tk = false;
    for (int e = 0; e < r; ++e) tk |= (taken[(size_t)c * kstride + e] == i);
    if (!tk) {
      v = vals[(size_t)c * M + i];
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
    blk_val[(size_t)c * nblk + blockIdx.x] = v;
    blk_idx[(size_t)c * nblk + blockIdx.x] = idx;
  }
}

// k best candidates of q criteria from their stored values vals[q][M]: out_val / out_idx are [q][k] on the device
// (a rank beyond the number of candidates comes back as (-inf, INT64_MAX)).  2 k launches, nothing read back here.
hipError_t launch_topk(const double* vals, int64_t M, int q, int k, double* blk_val, int64_t* blk_idx, double* out_val,
                       int64_t* out_idx, hipStream_t st) {
  const int64_t nblk = (M + 255) / 256;
  for (int r = 0; r < k; ++r) {
    hipLaunchKernelGGL(k_block_argmax_excl, dim3((unsigned)nblk, q), 256, 0, st, vals, M, out_idx, k, r, blk_val, blk_idx, nblk);
    hipLaunchKernelGGL(k_argmax_final, dim3(q), 256, 0, st, blk_val, blk_idx, nblk, nblk, out_val, out_idx, k, r);
  }
  return hipGetLastError();
}

// ---- on-device candidate generation ---------------------------------------------------------------------
// Uniform points in a box from the counter-based Philox4x32-10 generator (Salmon et al., SC'11; constants below
// are the published ones).  Element e = row * d + k of the M x d candidate array is a pure function of
// (seed, first_row * d + e): counter = (E >> 1, 0, 0, 0), key = (seed_lo, seed_hi); the pair of 32-bit outputs
// (0,1) serves even E, (2,3) odd E; u = ((a >> 5) * 2^26 + (b >> 6)) * 2^-53 in [0, 1); x = lo + (hi - lo) * u.
// Replaces the host-side `RealSpace._sample` (search_space.py:742-754) + the 8 d M byte H2D copy; the oracle
// restates the same integer arithmetic in NumPy (oracle/philox.py), so parity is bit-exact.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ __launch_bounds__(256) void k_generate_uniform(double* __restrict__ Xs, int64_t n_elem, int d,
                                                          const double* __restrict__ lo, const double* __restrict__ hi,
                                                          uint64_t seed, uint64_t first_elem) {
#pragma clang fp contract(off)
  // one thread per PAIR of consecutive GLOBAL elements (2P, 2P+1) = one Philox call with counter P
  const uint64_t P = (first_elem >> 1) + (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (2 * P >= first_elem + (uint64_t)n_elem) return;
  uint32_t w[4];
  philox4x32_10((uint32_t)P, (uint32_t)(P >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint64_t E = 2 * P + h;
    if (E < first_elem || E >= first_elem + (uint64_t)n_elem) continue;
    const double u = ((double)(w[2 * h] >> 5) * 67108864.0 + (double)(w[2 * h + 1] >> 6)) * (1.0 / 9007199254740992.0);
    const int k = (int)(E % (uint64_t)d);
    // separately rounded multiply and add (fp contract is off in this kernel): bit-identical to the NumPy restatement
    const double width = hi[k] - lo[k];
    const double scaled = width * u;
    Xs[E - first_elem] = lo[k] + scaled;
  }
}

hipError_t launch_generate_uniform(double* Xs, int64_t n_elem, int d, const double* lo, const double* hi, uint64_t seed,
                                   uint64_t first_elem, hipStream_t st) {
  const int64_t npair = (n_elem + 1) / 2 + 1;
  hipLaunchKernelGGL(k_generate_uniform, dim3((unsigned)((npair + 255) / 256)), 256, 0, st, Xs, n_elem, d, lo, hi, seed, first_elem);
  return hipGetLastError();
}

// Latin hypercube in a box (RealSpace._sample method "LHS", search_space.py:747-751, which calls pyDOE's lhs:
// one point per stratum and dimension, strata shuffled independently per dimension).  Counter-based so that rank
// shards are independent: row i of dimension k lands in stratum pi_k(i), a keyed pseudo-random PERMUTATION of
// [0, n_strata) evaluated per element -- a 6-round balanced Feistel network on 2*hb bits (hb = ceil(bits(n-1)/2))
// with cycle walking, round function fmix32 (the MurmurHash3 finaliser), round keys = Philox(counter (k, 'LHS', r/4),
// key seed).  The jitter u inside the stratum is the element's word of the uniform stream above;
// x = lo + (hi - lo) * ((pi + u) / n).  pyDOE's "maximin" criterion (5 O(n^2 d) pdist passes) is not reproduced: it
// is infeasible at sweep sizes in the reference too (n = 1e6 -> 4 TB of pair distances).  oracle/philox.py restates
// the integer arithmetic, so parity is bit-exact.
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

__global__ __launch_bounds__(256) void k_generate_lhs(double* __restrict__ Xs, int64_t n_elem, int d,
                                                      const double* __restrict__ lo, const double* __restrict__ hi,
                                                      uint64_t seed, uint64_t first_elem, uint64_t n_strata, int hb) {
#pragma clang fp contract(off)
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_elem) return;
  const uint64_t E = first_elem + (uint64_t)e;
  const uint32_t k = (uint32_t)(E % (uint64_t)d);
  uint64_t i = E / (uint64_t)d;
  uint32_t rk[8];
  {
    uint32_t w[4];
    philox4x32_10(k, 0x4C4853u, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    rk[0] = w[0]; rk[1] = w[1]; rk[2] = w[2]; rk[3] = w[3];
    philox4x32_10(k, 0x4C4853u, 1u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    rk[4] = w[0]; rk[5] = w[1]; rk[6] = w[2]; rk[7] = w[3];
  }
  const uint32_t mask = (uint32_t)((1ull << hb) - 1ull);
  if (n_strata > 1) {
    do {
      uint32_t L = (uint32_t)(i >> hb), R = (uint32_t)i & mask;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const uint32_t F = fmix32(R ^ rk[r]) & mask;
        const uint32_t t = L ^ F;
        L = R;
        R = t;
      }
      i = ((uint64_t)L << hb) | (uint64_t)R;
    } while (i >= n_strata);
  } else {
    i = 0;
  }
  const uint64_t P = E >> 1;
  uint32_t w[4];
  philox4x32_10((uint32_t)P, (uint32_t)(P >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
  const int h2 = (int)(E & 1ull);
  const double u = ((double)(w[2 * h2] >> 5) * 67108864.0 + (double)(w[2 * h2 + 1] >> 6)) * (1.0 / 9007199254740992.0);
  const double t = ((double)i + u) / (double)n_strata;
  const double width = hi[k] - lo[k];
  const double scaled = width * t;
  Xs[e] = lo[k] + scaled;
}

hipError_t launch_generate_lhs(double* Xs, int64_t n_elem, int d, const double* lo, const double* hi, uint64_t seed,
                               uint64_t first_elem, uint64_t n_strata, hipStream_t st) {
  int bits = 0;
  while (bits < 63 && ((n_strata - 1) >> bits) != 0) ++bits;
  const int hb = bits < 2 ? 1 : (bits + 1) / 2;
  hipLaunchKernelGGL(k_generate_lhs, dim3((unsigned)((n_elem + 255) / 256)), 256, 0, st, Xs, n_elem, d, lo, hi, seed, first_elem,
                     n_strata, hb);
  return hipGetLastError();
}

// Sobol' points in a box (RealSpace._sample method "sobol", search_space.py:752-753).  The direction numbers are an
// INPUT (d x bits integers, sv[k][b] for bit b of the Gray code, as scipy.stats.qmc.Sobol holds them in `_sv`), so
// the library carries no table; point n is XOR_{b in gray(n)} sv[k][b], scaled by 2^-bits -- exactly the sequence
// scipy's unscrambled generator produces, any block [first_index, first_index + M) of it independently per rank.
__global__ __launch_bounds__(256) void k_generate_sobol(double* __restrict__ Xs, int64_t n_elem, int d,
                                                        const double* __restrict__ lo, const double* __restrict__ hi,
                                                        const uint64_t* __restrict__ sv, int bits, uint64_t first_elem) {
#pragma clang fp contract(off)
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_elem) return;
  const uint64_t E = first_elem + (uint64_t)e;
  const int k = (int)(E % (uint64_t)d);
  const uint64_t n = E / (uint64_t)d;
  uint64_t g = n ^ (n >> 1), v = 0;
  const uint64_t* svk = sv + (size_t)k * bits;
  while (g) {
    const int b = __builtin_ctzll(g);
    v ^= svk[b];
    g &= g - 1;
  }
  const double t = (double)v * __builtin_ldexp(1.0, -bits);
  const double width = hi[k] - lo[k];
  const double scaled = width * t;
  Xs[e] = lo[k] + scaled;
}

hipError_t launch_generate_sobol(double* Xs, int64_t n_elem, int d, const double* lo, const double* hi, const uint64_t* sv,
                                 int bits, uint64_t first_elem, hipStream_t st) {
  hipLaunchKernelGGL(k_generate_sobol, dim3((unsigned)((n_elem + 255) / 256)), 256, 0, st, Xs, n_elem, d, lo, hi, sv, bits, first_elem);
  return hipGetLastError();
}

// Post-processing of generated candidates exactly where RealSpace._sample does it (search_space.py:754:
// `self.round(self.to_linear_scale(X))`): the designs above are drawn in the TRANSFORMED box (variable.py:246
// `_bounds_transformed`), each coordinate then goes back to the linear scale with the variable's inverse transform
// (variable.py:40-55: log -> exp, log10 -> 10^x, logit -> 1 / (1 + exp(-x)), bilog -> sign(x) (exp|x| - 1)) and, if the
// variable has a precision, is rounded to that many decimals and clipped to its bounds (variable.py:250-257; np.round is
// rint(x * 10^p) / 10^p).  spec per dimension k: [scale id, precision or -1, lo, hi] as four doubles.
__global__ __launch_bounds__(256) void k_candidates_transform(double* __restrict__ Xs, int64_t n_elem, int d,
                                                              const double* __restrict__ spec) {
#pragma clang fp contract(off)
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_elem) return;
  const int k = (int)(e % d);
  const int scale = (int)spec[4 * k];
  const int prec = (int)spec[4 * k + 1];
  double x = Xs[e];
  if (scale == BOGP_SCALE_LOG) {
    x = exp(x);
  } else if (scale == BOGP_SCALE_LOG10) {
    x = pow(10.0, x);
  } else if (scale == BOGP_SCALE_LOGIT) {
    x = 1.0 / (1.0 + exp(-x));
  } else if (scale == BOGP_SCALE_BILOG) {
    const double a = exp(fabs(x)) - 1.0;
    x = x > 0.0 ? a : (x < 0.0 ? -a : 0.0 * a);
  }
  if (prec >= 0) {
    double mult = 1.0;
    for (int i = 0; i < prec; ++i) mult *= 10.0;
    const double scaled = x * mult;
    x = rint(scaled) / mult;
    x = fmin(fmax(x, spec[4 * k + 2]), spec[4 * k + 3]);
  }
  Xs[e] = x;
}
hipError_t launch_candidates_transform(double* Xs, int64_t n_elem, int d, const double* spec, hipStream_t st) {
  hipLaunchKernelGGL(k_candidates_transform, dim3((unsigned)((n_elem + 255) / 256)), 256, 0, st, Xs, n_elem, d, spec);
  return hipGetLastError();
}

hipError_t launch_acquisition(const AcqArgs& a, hipStream_t st) {
  const unsigned nblk = (unsigned)((a.mcount + 255) / 256);
  hipLaunchKernelGGL(k_acquisition, dim3(nblk), 256, 0, st, a);
  return hipGetLastError();
}

hipError_t launch_argmax_final(const double* blk_val, const int64_t* blk_idx, int64_t nblk, int64_t stride, int q,
                               double* out_val, int64_t* out_idx, hipStream_t st) {
  hipLaunchKernelGGL(k_argmax_final, dim3(q), 256, 0, st, blk_val, blk_idx, nblk, stride, out_val, out_idx, 1, 0);
  return hipGetLastError();
}

}  // namespace bogp
