// bogp_device.h -- device-side helpers shared by the gfx950 kernels of libbogp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bogp.h"

namespace bogp {

// ---------------------------------------------------------------------------------------------------
// Radial profile of the correlation functions (surrogate/gaussian_process/kernel.py, see bogp.h).
// s2 = sum_k theta_k (x_k - y_k)^2 computed by the caller.  Operation order follows the reference:
//   matern: dists = sqrt(s2); K = dists * sqrt(nu2); (1 + K [+ K^2/3]) * exp(-K)      (kernel.py:186-200)
// ---------------------------------------------------------------------------------------------------
// One term of the weighted distance: theta_k d_k^2 for the radial kernels, theta_k |d_k| for absolute_exponential.
template <int KERNEL>
__device__ __forceinline__ double dist_term(double theta_k, double diff) {
  return KERNEL == BOGP_KERNEL_ABSEXP ? theta_k * fabs(diff) : theta_k * (diff * diff);
}
// cubic (kernel.py:419-466) is a PRODUCT over the dimensions, not a function of a summed distance:
//   td = min(1, theta_k |d_k|);  factor = 1 - td^2 (3 - 2 td);  r = prod_k factor
// so the per-pair accumulator starts at dist_init (1 for cubic, 0 otherwise), folds one dimension at a time and the
// radial profile of cubic is the identity.
__device__ __forceinline__ double cubic_factor(double td_abs) {
  const double td = td_abs > 1.0 ? 1.0 : td_abs;
  return 1.0 - (td * td) * (3.0 - 2.0 * td);
}
template <int KERNEL>
__device__ __forceinline__ double dist_init() {
  return KERNEL == BOGP_KERNEL_CUBIC ? 1.0 : 0.0;
}
// generalized_exponential (kernel.py:332-379): exp(-sum_k theta_k |d_k|^p); the exponent p travels as entry d of the
// device theta arrays (`pexp` below; unused by the other kernels)
template <int KERNEL>
__device__ __forceinline__ double kernel_exponent(const double* theta_like, int d) {
  return (KERNEL == BOGP_KERNEL_GENEXP || KERNEL == BOGP_KERNEL_MATERN_NU) ? theta_like[d] : 0.0;
}
// fold dimension k (unscaled coordinates, weight theta_k) into the accumulator
template <int KERNEL>
__device__ __forceinline__ double dist_fold(double theta_k, double diff, double acc, double pexp = 0.0) {
  if (KERNEL == BOGP_KERNEL_CUBIC) return acc * cubic_factor(fabs(diff) * theta_k);
  if (KERNEL == BOGP_KERNEL_GENEXP) return acc + theta_k * pow(fabs(diff), pexp);
  return acc + dist_term<KERNEL>(theta_k, fabs(diff));
}
// the same when both points were pre-scaled (by sqrt(theta_k); by theta_k for absolute_exponential and cubic; by
// theta_k^(1/p) for generalized_exponential)
template <int KERNEL>
__device__ __forceinline__ double dist_accumulate(double diff_scaled, double acc, double pexp = 0.0) {
  if (KERNEL == BOGP_KERNEL_CUBIC) return acc * cubic_factor(fabs(diff_scaled));
  if (KERNEL == BOGP_KERNEL_GENEXP) return acc + pow(fabs(diff_scaled), pexp);
  return KERNEL == BOGP_KERNEL_ABSEXP ? acc + fabs(diff_scaled) : __builtin_fma(diff_scaled, diff_scaled, acc);
}

// ---- modified Bessel function of the second kind K_nu(x), real order nu >= 0, x > 0 -----------------------------------------------
// What the reference gets from scipy.special.kv in the general-nu arm of its Matern kernel (kernel.py:201-207).  Method: with
// nu = mu + n, |mu| <= 1/2, K_mu and K_mu+1 come from Temme's series (N. M. Temme, J. Comput. Phys. 19 (1975)) for x <= 2 and from
// the continued fraction CF2 evaluated by Steed's algorithm (I. J. Thompson, A. R. Barnett, J. Comput. Phys. 64 (1986)) for x > 2;
// the order is then raised by the (upward-stable) recurrence K_{m+1} = K_{m-1} + (2 m / x) K_m.  Relative accuracy ~1e-14 (measured
// against scipy on the CPU restatement of the same steps, tests/test_oracle_golden.py), far inside the path's 1e-6.
// 1 / Gamma(1 +- mu) come from the device's tgamma; their scaled difference gam1 = (1/Gamma(1-mu) - 1/Gamma(1+mu)) / (2 mu)
// switches to its Taylor form -(g + a3 mu^2) below |mu| = 1e-4 (g = Euler's constant, a3 = the cubic coefficient of 1 / Gamma(1 + z),
// Abramowitz & Stegun 6.1.34), where the difference would cancel.
__device__ __forceinline__ double bessel_k_nu(double nu, double x) {
  const double EPS = 1.0e-16, PI = 3.141592653589793;
  const int nl = (int)(nu + 0.5);
  const double mu = nu - (double)nl, mu2 = mu * mu;
  double kmu, kmu1;
  if (x <= 2.0) {
    const double gampl = 1.0 / tgamma(1.0 + mu), gammi = 1.0 / tgamma(1.0 - mu);
    const double gam1 = fabs(mu) < 1.0e-4 ? -(0.5772156649015329 + (-0.0420026350340952) * mu2) : (gammi - gampl) / (2.0 * mu);
    const double gam2 = 0.5 * (gammi + gampl);
    const double b = 0.5 * x;
    double dd = -log(b);
    double e = mu * dd;
    const double fact2 = fabs(e) < EPS ? 1.0 : sinh(e) / e;
    const double pimu = PI * mu;
    const double fact = fabs(pimu) < EPS ? 1.0 : pimu / sin(pimu);
    double ff = fact * (gam1 * cosh(e) + gam2 * fact2 * dd);
    double sum = ff;
    e = exp(e);
    double p = 0.5 * e / gampl;
    double q = 0.5 / (e * gammi);
    double c = 1.0;
    dd = b * b;
    double sum1 = p;
    for (int i = 1; i <= 500; ++i) {
      const double di = (double)i;
      ff = (di * ff + p + q) / (di * di - mu2);
      c *= dd / di;
      p /= di - mu;
      q /= di + mu;
      const double del = c * ff;
      sum += del;
      sum1 += c * (p - di * ff);
      if (fabs(del) < fabs(sum) * EPS) break;
    }
    kmu = sum;
    kmu1 = sum1 * (2.0 / x);
  } else {
    double b = 2.0 * (1.0 + x);
    double dd = 1.0 / b;
    double h = dd, delh = dd;
    double q1 = 0.0, q2 = 1.0;
    const double a1 = 0.25 - mu2;
    double q = a1, c = a1;
    double a = -a1;
    double s = 1.0 + q * delh;
    for (int i = 2; i <= 500; ++i) {
      a -= 2.0 * (double)(i - 1);
      c = -a * c / (double)i;
      const double qnew = (q1 - b * q2) / a;
      q1 = q2;
      q2 = qnew;
      q += c * qnew;
      b += 2.0;
      dd = 1.0 / (b + a * dd);
      delh = (b * dd - 1.0) * delh;
      h += delh;
      const double dels = q * delh;
      s += dels;
      if (fabs(dels / s) < EPS) break;
    }
    h = a1 * h;
    kmu = sqrt(PI / (2.0 * x)) * exp(-x) / s;
    kmu1 = kmu * (mu + x + 0.5 - h) / x;
  }
  for (int i = 1; i <= nl; ++i) {
    const double knew = (mu + (double)i) * (2.0 / x) * kmu1 + kmu;
    kmu = kmu1;
    kmu1 = knew;
  }
  return kmu;
}

// `pexp`: the exponent p of generalized_exponential (used in dist_fold, not here) / the order nu of the general Matern kernel
template <int KERNEL>
__device__ __forceinline__ double corr_profile(double s2, double pexp = 0.0) {
  if (KERNEL == BOGP_KERNEL_CUBIC) return s2;  // the accumulator already is the product
  if (KERNEL == BOGP_KERNEL_MATERN_NU) {
    // kernel.py:201-207: K = dists; zeros += eps; tmp = sqrt(2 nu) K; (2^(1 - nu) / gamma(nu)) tmp^nu kv(nu, tmp)
    double K = sqrt(s2);
    if (K == 0.0) K += 2.220446049250313e-16;
    const double tmp = sqrt(2.0 * pexp) * K;
    double r = pow(2.0, 1.0 - pexp) / tgamma(pexp);
    r *= pow(tmp, pexp);
    r *= bessel_k_nu(pexp, tmp);
    return r;
  }
  if (KERNEL == BOGP_KERNEL_SE || KERNEL == BOGP_KERNEL_ABSEXP || KERNEL == BOGP_KERNEL_GENEXP) return exp(-s2);
  const double dists = sqrt(s2);
  if (KERNEL == BOGP_KERNEL_MATERN12) return exp(-dists);
  if (KERNEL == BOGP_KERNEL_MATERN32) {
    const double K = dists * 1.7320508075688772;  // math.sqrt(3)
    return (1.0 + K) * exp(-K);
  }
  const double K = dists * 2.23606797749979;  // math.sqrt(5)
  // K**2 / 3.0 in the reference; a multiply by the rounded reciprocal differs by <= 1 ulp of that term and saves
  // an FP64 division sequence (~20 DP ops) per pair
  return (1.0 + K + (K * K) * 0.3333333333333333) * exp(-K);
}

// -h(D) such that dR0/dtheta_k = -(x_ik - x_jk)^2 * h  (gpr.py:736-770 corr_grad_theta):
//   SE r; Matern-3/2 1.5 exp(-sqrt3 D); [extensions: Matern-5/2 (5/6)(1+sqrt5 D)exp(-sqrt5 D); Matern-1/2 r/(2D)]
//   absolute_exponential: dR0/dtheta_k = -|x_ik - x_jk| * r  (:761-762) -- same h = r, first power of the distance
template <int KERNEL>
__device__ __forceinline__ double dtheta_weight(double diff) {
  return KERNEL == BOGP_KERNEL_ABSEXP ? fabs(diff) : diff * diff;
}
// r0 = corr_profile(s2) and h (the comment above) with the square root and the exponential they share evaluated once: the likelihood gradient's
// pair work (k_grad_contract, k_nll_small)
template <int KERNEL>
__device__ __forceinline__ void corr_pair(double s2, double& r0, double& h) {
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

// scipy.special.ndtr (cephes ndtr.c) branch structure on top of the device erf/erfc:
//   x = a * sqrt(1/2); z = |x|; z < sqrt(1/2): .5 + .5 erf(x); else y = .5 erfc(z), x > 0 -> 1 - y
__device__ __forceinline__ double ndtr(double a) {
  if (isnan(a)) return a;
  const double x = a * 0.70710678118654752440;
  const double z = fabs(x);
  if (z < 0.70710678118654752440) return 0.5 + 0.5 * erf(x);
  double y = 0.5 * erfc(z);
  if (x > 0) y = 1.0 - y;
  return y;
}

// scipy.stats.norm.pdf: exp(-x**2/2.0) / sqrt(2*pi)
__device__ __forceinline__ double norm_pdf(double x) { return exp(-(x * x) / 2.0) / 2.5066282746310002; }

// The q acquisition criteria of one row, guards as selects (acquisition_fun.py:127-135, 153-176, 208-217, 265-290); shared by
// k_acquisition (chunked sweep) and k_sweep_small (fused small-N sweep) so that both evaluate the same expressions.
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

// posterior of one row from its three sums (gpr.py:490, 496-510): mu = beta + r.gamma, MSE = (1 - |L^-1 r|^2 + u^2) sigma2
// clipped at 0, u = (w.r - 1) / G under ordinary kriging with the constant basis
__device__ __forceinline__ void posterior_of_sums(double rgamma, double wr, double ss, double beta, double G, int estimate_trend,
                                                  double sigma2, double& mu, double& mse) {
  mu = beta + rgamma;
  double u2 = 0.0;
  if (estimate_trend) {
    const double u = (wr - 1.0) / G;
    u2 = u * u;
  }
  mse = (1.0 - ss + u2) * sigma2;
  if (mse < 0.0) mse = 0.0;
}

// argmax ordering identical to np.argmax over a 1-D float64 array: first maximal element, where a NaN
// (if any) is maximal.  (value, index) pairs; `better(a,b)` == a should replace b.
struct ArgMax {
  double v;
  int64_t i;
};
__device__ __forceinline__ bool better(double av, int64_t ai, double bv, int64_t bi) {
  const bool an = isnan(av), bn = isnan(bv);
  if (an != bn) return an;
  if (!an && av != bv) return av > bv;
  return ai < bi;
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  int lo = __shfl_xor(__double2loint(v), mask, 64);
  int hi = __shfl_xor(__double2hiint(v), mask, 64);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int64_t shfl_xor_i64(int64_t v, int mask) {
  int lo = __shfl_xor((int)(v & 0xffffffffll), mask, 64);
  int hi = __shfl_xor((int)(v >> 32), mask, 64);
  return ((int64_t)hi << 32) | (uint32_t)lo;
}

// ---- 4 x 4 pivot arithmetic of the in-place elimination (kernels_nllsmall.hip: k_nll_small; kernels_chol.hip: elim_diag) ----
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

}  // namespace bogp
