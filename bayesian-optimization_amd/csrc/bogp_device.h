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
// What the reference gets from scipy.special.kv in the general-nu arm of its Matern kernel (kernel.py:201-207).  scipy's kv (AMOS zbesk) is
// itself up to ~500 eps from the true value on nu in (0, 10], x in [1e-8, 700] (profiles/r05_kv_accuracy.txt, against mpmath at 40 digits), so
// the target here is the TRUE value: <= 5 eps over that domain (same file; tests/test_gpu_special.py holds the device to a committed mpmath table).
// Method, with nu = mu + n, |mu| <= 1/2:
//   * x <= 1: Temme's series (N. M. Temme, J. Comput. Phys. 19 (1975) 324) for K_mu, K_mu+1.  1/Gamma(1 -+ mu) and their scaled difference
//     gam1 = (1/Gamma(1-mu) - 1/Gamma(1+mu)) / (2 mu), gam2 = their mean, come from Chebyshev expansions in 8 mu^2 - 1 (coefficients by mpmath,
//     tools/kv_coefficients.py) -- no cancellation at any mu; (x/2)^(-+mu) from pow(), whose exponent reduction is exact, not from exp(mu log(x/2)),
//     which loses |mu log(x/2)| eps at small x;
//   * x > 1: the trapezoidal rule on K_mu(x) = e^-x int_0^inf exp(-2 x sinh^2(t/2)) cosh(mu t) dt with step min(0.2, 0.55 / sqrt(x)): the
//     integrand is analytic in |Im t| < pi/2, so the rule converges geometrically (~20 points for 1e-17), every term is positive (no cancellation)
//     and e^-x is factored out so that the exponent's rounding error is relative to 2 x sinh^2, not to x.  (Steed's CF2, used until r04,
//     accumulates ~10 eps of recurrence rounding over its 30-170 iterations at x in [1, 3].)
//   * the order is raised by the upward-stable recurrence K_{m+1} = K_{m-1} + (2 m / x) K_m in double-double arithmetic, so that the n <= 10
//     steps add nothing to the error of the two starting values.
// `rgamma_nu` (out): 1 / Gamma(nu) from the same expansion, Gamma(nu) = Gamma(1 + mu) prod_{i=1}^{n-1} (mu + i)  (Gamma(1 + mu) / mu for n = 0).
struct DD {
  double h, l;
};
__device__ __forceinline__ DD dd_fast_sum(double a, double b) {  // |a| >= |b|
  const double s = a + b;
  return {s, b - (s - a)};
}
__device__ __forceinline__ DD dd_sum(double a, double b) {
  const double s = a + b, bb = s - a;
  return {s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ DD dd_mul(DD a, DD b) {
  const double p = a.h * b.h;
  double e = __builtin_fma(a.h, b.h, -p);
  e = __builtin_fma(a.h, b.l, __builtin_fma(a.l, b.h, e));
  return dd_fast_sum(p, e);
}
__device__ __forceinline__ DD dd_add(DD a, DD b) {
  DD s = dd_sum(a.h, b.h);
  s.l += a.l + b.l;
  return dd_fast_sum(s.h, s.l);
}
__device__ __forceinline__ DD dd_div_d(DD a, double x) {
  const double q1 = a.h / x;
  const double r = __builtin_fma(-q1, x, a.h) + a.l;
  return dd_fast_sum(q1, r / x);
}
__device__ __forceinline__ double cheb_even(const double* c, int n, double t) {  // Clenshaw: sum_j c_j T_j(t)
  double b1 = 0.0, b2 = 0.0;
  const double t2 = 2.0 * t;
  for (int j = n - 1; j >= 1; --j) {
    const double b0 = __builtin_fma(t2, b1, c[j] - b2);
    b2 = b1;
    b1 = b0;
  }
  return __builtin_fma(t, b1, c[0] - b2);
}
__device__ __forceinline__ double bessel_k_nu(double nu, double x, double* rgamma_nu = nullptr) {
  const double PI = 3.141592653589793;
  // gam1, gam2 on |mu| <= 1/2 as Chebyshev series in t = 8 mu^2 - 1 (truncation < 1e-18)
  const double G1[9] = {-0.5710113401855839203,    0.0065165112670736880645,  0.00030870901730853682431,
                        -3.470626964904317836e-6,  6.9437664486674495957e-9,  3.6779539885744101652e-11,
                        -1.3563951023664248708e-13, -3.6802984806357979599e-17, 5.4582162333769858553e-19};
  const double G2[10] = {0.92187029365045265648,     -0.07685284084478667369,    0.0012719271366545622927, -4.9717367041957398581e-6,
                         -3.3126119768180852711e-8,  2.4230957900482704055e-10,  -1.7023776642512729175e-13, -1.4943667065169001769e-15,
                         2.3826220476859635824e-18,  2.9017595056104745456e-21};
  const int nl = (int)(nu + 0.5);
  const double mu = nu - (double)nl, mu2 = mu * mu;
  const double tc = __builtin_fma(8.0, mu2, -1.0);
  const double gam1 = cheb_even(G1, 9, tc), gam2 = cheb_even(G2, 10, tc);
  const double gampl = __builtin_fma(-mu, gam1, gam2), gammi = __builtin_fma(mu, gam1, gam2);  // 1 / Gamma(1 + mu), 1 / Gamma(1 - mu)
  if (rgamma_nu) {
    DD g = dd_div_d(DD{1.0, 0.0}, gampl);  // Gamma(1 + mu)
    if (nl == 0) {
      g = dd_div_d(g, mu);
    } else {
      for (int i = 1; i < nl; ++i) g = dd_mul(g, dd_sum((double)i, mu));
    }
    *rgamma_nu = 1.0 / (g.h + g.l);
  }
  double kmu, kmu1;
  if (x <= 1.0) {
    const double b = 0.5 * x;
    double dd = -log(b);
    const double e = mu * dd;
    const double pw = pow(b, -mu), pwi = 1.0 / pw;  // exp(+-e)
    double fact2, ch;
    if (fabs(e) < 0.5) {
      fact2 = fabs(e) < 1e-8 ? 1.0 : sinh(e) / e;
      ch = cosh(e);
    } else {
      fact2 = 0.5 * (pw - pwi) / e;
      ch = 0.5 * (pw + pwi);
    }
    const double pimu = PI * mu;
    const double fact = fabs(pimu) < 1e-8 ? 1.0 : pimu / sin(pimu);
    double ff = fact * (gam1 * ch + gam2 * fact2 * dd);
    double sum = ff;
    double p = 0.5 * pw / gampl;
    double q = 0.5 * pwi / gammi;
    double c = 1.0;
    dd = b * b;
    double sum1 = p;
    for (int i = 1; i <= 60; ++i) {
      const double di = (double)i;
      ff = (di * ff + p + q) / (di * di - mu2);
      c *= dd / di;
      p /= di - mu;
      q /= di + mu;
      const double del = c * ff;
      sum += del;
      sum1 += c * (p - di * ff);
      if (fabs(del) < fabs(sum) * 1.0e-17) break;
    }
    kmu = sum;
    kmu1 = sum1 * (2.0 / x);
  } else {
    const double h = fmin(0.2, 0.55 / sqrt(x));
    double s0 = 0.5, s1 = 0.5;
    for (int k = 1; k <= 200; ++k) {
      const double t = (double)k * h;
      const double sh = sinh(0.5 * t);
      const double w = exp(-2.0 * x * sh * sh);
      const double wb = w * cosh((mu + 1.0) * t);
      s0 = __builtin_fma(w, cosh(mu * t), s0);
      s1 += wb;
      if (wb < 1.0e-18 * s1) break;
    }
    const double ex = exp(-x) * h;
    kmu = ex * s0;
    kmu1 = ex * s1;
  }
  DD k0{kmu, 0.0}, k1{kmu1, 0.0};
  for (int i = 1; i <= nl; ++i) {
    DD c = dd_sum((double)i, mu);
    c.h *= 2.0;
    c.l *= 2.0;
    const DD kn = dd_add(dd_mul(dd_div_d(c, x), k1), k0);
    k0 = k1;
    k1 = kn;
  }
  return k0.h + k0.l;
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
    double rg;
    const double kv = bessel_k_nu(pexp, tmp, &rg);
    double r = pow(2.0, 1.0 - pexp) * rg;
    r *= pow(tmp, pexp);
    r *= kv;
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
// reciprocal square root estimate + ONE third-order (Halley) step, y' = y + (y e)(1/2 + 3 e / 8) with e = 1 - p y^2 -- FOUR dependent
// operations after v_rsq_f64 (t, e, {y e | 1/2 + 3 e / 8}, fma; r05: the two factors of the correction formed side by side -- the r02 form
// y + y ((1/2 + 3 e / 8) e) had five) where sqrt() followed by a division is ~45; a dependent FP64 operation costs ~26 cycles in a lone
// wave and this chain is on the critical path of EVERY step (kernels_chol.hip: rsqrt_nr, the r02 arithmetic, serves the opt-in chain only)
__device__ __forceinline__ void ns_sqrt_rsqrt(double p, double& root, double& inv) {
  const double y = __builtin_amdgcn_rsq(p);
  const double t = p * y;
  const double e = __builtin_fma(-t, y, 1.0);
  const double ye = y * e;
  const double q = __builtin_fma(0.375, e, 0.5);
  inv = __builtin_fma(ye, q, y);
  root = p * inv;
}

// acc - (a0 b0 + a1 b1 + a2 b2 + a3 b3), the rank-4 correction of one entry, as two chains of two products and a subtraction: three
// dependent operations instead of four (the diagonal block's entry (0, 0) is the head of a step's chain)
__device__ __forceinline__ double ns_dot4_sub(double acc, double a0, double b0, double a1, double b1, double a2, double b2, double a3, double b3) {
  double s1 = __builtin_fma(-a0, b0, acc);
  double s2 = a2 * b2;
  s1 = __builtin_fma(-a1, b1, s1);
  s2 = __builtin_fma(a3, b3, s2);
  return s1 - s2;
}

// 4 x 4 Cholesky of the lower triangle of a: l (strict lower part) and inv[c] = 1 / l_cc -- the panel's rows are then solved by
// substitution, o = M L^-T, column c of o as soon as pivot c is known: off the pivots' dependent chain except for one product.
// A pivot that is not in (0, 1e300) is reported (the first one, 1-based) and replaced by 1 as before, but the test no longer sits in
// front of the reciprocal square root: the estimate + Halley step run on the raw pivot while the comparison is evaluated, and ONE select
// on the result (inv = 1, l_cc = 1 for a rejected pivot -- the values the replaced pivot gave) closes the chain: a compare less per pivot.
__device__ __forceinline__ int ns_factor4_sub(const double (&a)[4][4], double (&l)[4][4], double (&inv)[4], double& pivprod) {
  int bad = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double p = a[c][c];
#pragma unroll
    for (int m = 0; m < c; ++m) p = __builtin_fma(-l[c][m], l[c][m], p);
    const bool ok = (p > 0.0) && (p < 1e300);
    if (!ok && !bad) bad = c + 1;
    double lc, ic;
    ns_sqrt_rsqrt(p, lc, ic);
    inv[c] = ok ? ic : 1.0;
    lc = ok ? lc : 1.0;
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
