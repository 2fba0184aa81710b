// bogp_lbfgsb.h -- a bound-constrained limited-memory BFGS minimiser written as a re-entrant state machine (host code).
//
// Why it exists: the reference fits its hyper-parameters with scipy.optimize.fmin_l_bfgs_b, one restart after the other
// (gpr.py:1127-1162), and every objective call is one likelihood evaluation.  To evaluate the restarts TOGETHER (bogp_nll_batch)
// the optimiser must hand out its next trial point and take (f, g) back later -- `x()` / `tell()` below -- so that R instances can
// be advanced in lock step by one batched device call per round (bogp_mle.hip).  scipy's driver cannot be suspended that way
// from C, and its Python-side overhead was ~40 % of a small fit (profiles/r03_bo_loop.txt).
//
// Algorithm: L-BFGS-B as published -- R. H. Byrd, P. Lu, J. Nocedal, C. Zhu, "A limited memory algorithm for bound constrained
// optimization", SIAM J. Sci. Comput. 16 (1995): generalised Cauchy point along the projected steepest-descent path (their
// Algorithm CP), direct primal subspace minimisation over the free variables with the compact representation
// B = theta I - W M W^T, projection of the subspace point with the descent safeguard of J. L. Morales, J. Nocedal, "Remark on
// Algorithm 778" (ACM TOMS 38, 2011); line search of J. J. More', D. J. Thuente, "Line search algorithms with guaranteed
// sufficient decrease" (ACM TOMS 20, 1994) with ftol = 1e-3, gtol = 0.9, xtol = 0.1; the stopping rules and defaults are the ones
// scipy documents for fmin_l_bfgs_b (m = 10, factr = 1e7, pgtol = 1e-5, maxls = 20), so that a restart here stops where the
// reference's would.  Dense algebra on (2 m) x (2 m) matrices: n is the number of hyper-parameters (<= a few hundred), not N.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

namespace bogp {

// ---- the line search: sufficient decrease + curvature, safeguarded cubic / quadratic steps -------------------------------------
class MoreThuente {
 public:
  enum Result { EVALUATE, CONVERGED, WARNING };
  void start(double f0, double g0, double stp0, double stpmin, double stpmax, double ftol, double gtol, double xtol) {
    ftol_ = ftol; gtol_ = gtol; xtol_ = xtol; stpmin_ = stpmin; stpmax_ = stpmax;
    brackt_ = false; stage_ = 1;
    finit_ = f0; ginit_ = g0; gtest_ = ftol * g0;
    width_ = stpmax - stpmin; width1_ = 2.0 * width_;
    stx_ = 0; fx_ = f0; gx_ = g0;
    sty_ = 0; fy_ = f0; gy_ = g0;
    stmin_ = 0; stmax_ = stp0 + 4.0 * stp0;
    stp_ = stp0;
  }
  double step() const { return stp_; }
  // f, g: value and directional derivative at step(); on EVALUATE step() is the next trial
  Result advance(double f, double g) {
    const double stp = stp_;
    const double ftest = finit_ + stp * gtest_;
    if (stage_ == 1 && f <= ftest && g >= 0.0) stage_ = 2;
    bool warn = false;
    if (brackt_ && (stp <= stmin_ || stp >= stmax_)) warn = true;              // rounding errors prevent progress
    if (brackt_ && stmax_ - stmin_ <= xtol_ * stmax_) warn = true;             // the interval of uncertainty is at its tolerance
    if (stp == stpmax_ && f <= ftest && g <= gtest_) warn = true;              // at the upper bound
    if (stp == stpmin_ && (f > ftest || g >= gtest_)) warn = true;             // at the lower bound
    if (f <= ftest && std::fabs(g) <= gtol_ * (-ginit_)) return CONVERGED;
    if (warn) return WARNING;
    if (stage_ == 1 && f <= fx_ && f > ftest) {
      // first stage: the auxiliary function psi(stp) = f(stp) - f(0) - ftol stp f'(0) decides the interval update
      double fm = f - stp * gtest_, fxm = fx_ - stx_ * gtest_, fym = fy_ - sty_ * gtest_;
      double gm = g - gtest_, gxm = gx_ - gtest_, gym = gy_ - gtest_;
      trial(stx_, fxm, gxm, sty_, fym, gym, stp_, fm, gm, stmin_, stmax_);
      fx_ = fxm + stx_ * gtest_; fy_ = fym + sty_ * gtest_;
      gx_ = gxm + gtest_; gy_ = gym + gtest_;
    } else {
      trial(stx_, fx_, gx_, sty_, fy_, gy_, stp_, f, g, stmin_, stmax_);
    }
    if (brackt_) {  // force a sufficient decrease of the interval
      if (std::fabs(sty_ - stx_) >= 0.66 * width1_) stp_ = stx_ + 0.5 * (sty_ - stx_);
      width1_ = width_;
      width_ = std::fabs(sty_ - stx_);
    }
    if (brackt_) {
      stmin_ = std::min(stx_, sty_);
      stmax_ = std::max(stx_, sty_);
    } else {
      stmin_ = stp_ + 1.1 * (stp_ - stx_);
      stmax_ = stp_ + 4.0 * (stp_ - stx_);
    }
    stp_ = std::min(std::max(stp_, stpmin_), stpmax_);
    if ((brackt_ && (stp_ <= stmin_ || stp_ >= stmax_)) || (brackt_ && stmax_ - stmin_ <= xtol_ * stmax_)) stp_ = stx_;
    return EVALUATE;
  }
  // a trial step whose objective was not finite: the admissible range ends before it
  void shrink_to(double frac) {
    stpmax_ = stp_;
    stp_ = stx_ + frac * (stp_ - stx_);
    if (brackt_) { stmin_ = std::min(stx_, sty_); stmax_ = std::max(stx_, sty_); }
  }
  double best_step() const { return stx_; }
  double best_value() const { return fx_; }

 private:
  // the safeguarded step of More' & Thuente, section 4 (MINPACK-2 dcstep): cases by the relative position of the new point and the best
  // point.  [lo, hi] is the MOVING interval dcsrch hands to dcstep -- [stmin, stmax] as left by the previous call: the bracket once there
  // is one, else [stp + 1.1 (stp - stx), stp + 4 (stp - stx)] -- so an un-bracketed extrapolation (cases 3 and 4) goes at most four
  // step-lengths beyond the best point, not straight to the global stpmax (that was the box edge in the MLE, 1e10 with open bounds).
  void trial(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy, double& stp, double fp, double dp, double lo,
             double hi) {
    const double sgnd = dp * (dx / std::fabs(dx));
    double stpf, stpc, stpq;
    if (fp > fx) {  // higher value: the minimum is bracketed
      const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
      const double s = std::max(std::fabs(theta), std::max(std::fabs(dx), std::fabs(dp)));
      double gamma = s * std::sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
      if (stp < stx) gamma = -gamma;
      const double p = (gamma - dx) + theta, q = ((gamma - dx) + gamma) + dp, r = p / q;
      stpc = stx + r * (stp - stx);
      stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
      stpf = std::fabs(stpc - stx) < std::fabs(stpq - stx) ? stpc : stpc + (stpq - stpc) / 2.0;
      brackt_ = true;
    } else if (sgnd < 0.0) {  // lower value, derivatives of opposite sign: bracketed
      const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
      const double s = std::max(std::fabs(theta), std::max(std::fabs(dx), std::fabs(dp)));
      double gamma = s * std::sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
      if (stp > stx) gamma = -gamma;
      const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dx, r = p / q;
      stpc = stp + r * (stx - stp);
      stpq = stp + (dp / (dp - dx)) * (stx - stp);
      stpf = std::fabs(stpc - stp) > std::fabs(stpq - stp) ? stpc : stpq;
      brackt_ = true;
    } else if (std::fabs(dp) < std::fabs(dx)) {  // lower value, same sign, the derivative decreases in magnitude
      const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
      const double s = std::max(std::fabs(theta), std::max(std::fabs(dx), std::fabs(dp)));
      double gamma = s * std::sqrt(std::max(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
      if (stp > stx) gamma = -gamma;
      const double p = (gamma - dp) + theta, q = (gamma + (dx - dp)) + gamma, r = p / q;
      if (r < 0.0 && gamma != 0.0) stpc = stp + r * (stx - stp);
      else stpc = stp > stx ? hi : lo;
      stpq = stp + (dp / (dp - dx)) * (stx - stp);
      if (brackt_) {
        stpf = std::fabs(stpc - stp) < std::fabs(stpq - stp) ? stpc : stpq;
        stpf = stp > stx ? std::min(stp + 0.66 * (sty - stp), stpf) : std::max(stp + 0.66 * (sty - stp), stpf);
      } else {
        stpf = std::fabs(stpc - stp) > std::fabs(stpq - stp) ? stpc : stpq;
        stpf = std::max(lo, std::min(hi, stpf));
      }
    } else {  // lower value, same sign, the derivative does not decrease
      if (brackt_) {
        const double theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
        const double s = std::max(std::fabs(theta), std::max(std::fabs(dy), std::fabs(dp)));
        double gamma = s * std::sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
        if (stp > sty) gamma = -gamma;
        const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dy, r = p / q;
        stpf = stp + r * (sty - stp);
      } else {
        stpf = stp > stx ? hi : lo;
      }
    }
    if (fp > fx) {
      sty = stp; fy = fp; dy = dp;
    } else {
      if (sgnd < 0.0) { sty = stx; fy = fx; dy = dx; }
      stx = stp; fx = fp; dx = dp;
    }
    stp = stpf;
  }
  double ftol_ = 0, gtol_ = 0, xtol_ = 0, stpmin_ = 0, stpmax_ = 0;
  bool brackt_ = false;
  int stage_ = 1;
  double finit_ = 0, ginit_ = 0, gtest_ = 0, width_ = 0, width1_ = 0;
  double stx_ = 0, fx_ = 0, gx_ = 0, sty_ = 0, fy_ = 0, gy_ = 0, stmin_ = 0, stmax_ = 0, stp_ = 0;
};

// ---- the minimiser ------------------------------------------------------------------------------------------------------------
class Lbfgsb {
 public:
  enum Status {
    RUNNING = -1,
    CONVERGED_PGTOL = 0,   // max |projected gradient| <= pgtol
    CONVERGED_FACTR = 1,   // (f_k - f_{k+1}) / max(|f_k|, |f_{k+1}|, 1) <= factr * epsmch
    STOP_MAXFUN = 2,       // evaluation budget (own or shared) reached at an iterate
    STOP_MAXITER = 3,
    ABNORMAL = 4,          // the line search failed twice from a fresh memory
    BAD_START = 5          // f or g not finite at the starting point
  };
  struct Options {
    int m = 10;
    double factr = 1e7, pgtol = 1e-5;
    int maxfun = 15000, maxiter = 15000, maxls = 20;
    const long* shared_evals = nullptr;  // lock-step restarts: evaluations made by ALL of them ...
    long shared_budget = 0;              // ... against this budget (0: none); checked where maxfun is, at new iterates
  };

  void start(int n, const double* x0, const double* lo, const double* hi, const Options& opt) {
    n_ = n; opt_ = opt; m_ = std::max(1, opt.m);
    lo_.assign(lo, lo + n); hi_.assign(hi, hi + n);
    x_.assign(x0, x0 + n);
    for (int i = 0; i < n; ++i) x_[i] = std::min(std::max(x_[i], lo_[i]), hi_[i]);
    // a bound of +-1e300 or beyond (or infinite) is "none": scipy's bounds=None / nbd = 0
    cnstnd_ = false; boxed_ = true;
    for (int i = 0; i < n; ++i) {
      const bool has_lo = lo_[i] > -1e300, has_hi = hi_[i] < 1e300;
      cnstnd_ = cnstnd_ || has_lo || has_hi;
      boxed_ = boxed_ && has_lo && has_hi;
    }
    xt_ = x_;
    g_.assign(n, 0.0); gold_.assign(n, 0.0); xold_.assign(n, 0.0); d_.assign(n, 0.0); xcp_.assign(n, 0.0);
    S_.assign((size_t)n * m_, 0.0); Y_.assign((size_t)n * m_, 0.0);
    col_ = 0; theta_ = 1.0;
    nfev_ = 0; nit_ = 0; nskip_ = 0;
    status_ = RUNNING; phase_ = FIRST;
    f_ = std::numeric_limits<double>::infinity();
  }
  bool running() const { return status_ == RUNNING; }
  const double* x() const { return xt_.data(); }           // where (f, g) is wanted next -- while running()
  const double* best_x() const { return x_.data(); }       // the last accepted iterate
  double best_f() const { return f_; }
  int nfev() const { return nfev_; }
  int nit() const { return nit_; }
  Status status() const { return status_; }

  // (f, g) at x(); returns running()
  bool tell(double f, const double* g) {
    if (status_ != RUNNING) return false;
    ++nfev_;
    if (phase_ == FIRST) {
      bool fin = std::isfinite(f);
      for (int i = 0; i < n_ && fin; ++i) fin = std::isfinite(g[i]);
      if (!fin) { f_ = f; status_ = BAD_START; return false; }
      f_ = f;
      std::copy(g, g + n_, g_.begin());
      if (proj_grad_norm() <= opt_.pgtol) { status_ = CONVERGED_PGTOL; return false; }
      return begin_iteration();
    }
    // ---- inside a line search: xt_ = xold_ + stp d_
    bool fin = std::isfinite(f);
    for (int i = 0; i < n_ && fin; ++i) fin = std::isfinite(g[i]);
    ++ls_evals_;
    if (!fin) {  // outside the domain (a factorisation that broke down): the step was too long
      if (ls_evals_ >= opt_.maxls) return line_search_failed();
      ls_.shrink_to(0.25);
      set_trial(ls_.step());
      return true;
    }
    double gd = 0.0;
    for (int i = 0; i < n_; ++i) gd += g[i] * d_[i];
    const MoreThuente::Result r = ls_.advance(f, gd);
    if (r == MoreThuente::EVALUATE) {
      if (ls_evals_ >= opt_.maxls) return line_search_failed();
      set_trial(ls_.step());
      return true;
    }
    // the step is accepted (with a warning: the best point the search has is the current one when it satisfies the decrease)
    if (r == MoreThuente::WARNING && !(f <= fold_)) return line_search_failed();
    const double stp = ls_.step();
    x_ = xt_;
    f_ = f;
    std::copy(g, g + n_, g_.begin());
    ++nit_;
    if (proj_grad_norm() <= opt_.pgtol) { status_ = CONVERGED_PGTOL; return false; }
    const double ddum = std::max(std::max(std::fabs(fold_), std::fabs(f_)), 1.0);
    if (fold_ - f_ <= std::numeric_limits<double>::epsilon() * opt_.factr * ddum) { status_ = CONVERGED_FACTR; return false; }
    // ---- curvature pair: s = stp d, y = g - g_old; skipped when s.y is not safely positive
    double sy = 0.0, yy = 0.0, gdold = 0.0;
    for (int i = 0; i < n_; ++i) {
      const double yi = g_[i] - gold_[i];
      sy += yi * d_[i];
      yy += yi * yi;
      gdold += gold_[i] * d_[i];
    }
    sy *= stp;
    const double dd = -gdold * stp;
    if (sy <= std::numeric_limits<double>::epsilon() * dd) {
      ++nskip_;
    } else {
      push_pair(stp);
      theta_ = yy / sy;
      if (!form_middle()) { col_ = 0; theta_ = 1.0; }
    }
    if (nit_ >= opt_.maxiter) { status_ = STOP_MAXITER; return false; }
    if (over_budget()) { status_ = STOP_MAXFUN; return false; }
    return begin_iteration();
  }
  // stop from outside (a shared budget ran out between iterates): the last accepted iterate stands
  void stop(Status s) { if (status_ == RUNNING) status_ = s; }

 private:
  enum Phase { FIRST, SEARCH };
  // scipy's driver tests `nfev > maxfun` when a new iterate has been accepted (never inside a line search): the same here, for the
  // restart's own limit and for the budget the lock-step restarts share
  bool over_budget() const {
    if (nfev_ > opt_.maxfun) return true;
    return opt_.shared_evals && opt_.shared_budget > 0 && *opt_.shared_evals > opt_.shared_budget;
  }

  double proj_grad_norm() const {
    double nrm = 0.0;
    for (int i = 0; i < n_; ++i) {
      double gi = g_[i];
      gi = gi < 0.0 ? std::max(x_[i] - hi_[i], gi) : std::min(x_[i] - lo_[i], gi);
      nrm = std::max(nrm, std::fabs(gi));
    }
    return nrm;
  }
  void set_trial(double stp) {
    if (stp == 1.0) {
      // (a full step lands exactly on the subspace point: bounds reached by it are reached without rounding)
      for (int i = 0; i < n_; ++i) xt_[i] = xbar_at(i);
    } else {
      for (int i = 0; i < n_; ++i) xt_[i] = std::min(std::max(xold_[i] + stp * d_[i], lo_[i]), hi_[i]);
    }
  }
  double xbar_at(int i) const { return std::min(std::max(xold_[i] + d_[i], lo_[i]), hi_[i]); }

  // W = [Y, theta S] row i as a 2 col vector
  void wrow(int i, double* w) const {
    for (int j = 0; j < col_; ++j) {
      w[j] = Y_[(size_t)i * m_ + j];
      w[col_ + j] = theta_ * S_[(size_t)i * m_ + j];
    }
  }
  void push_pair(double stp) {
    if (col_ == m_) {  // drop the oldest pair
      for (int i = 0; i < n_; ++i) {
        for (int j = 1; j < m_; ++j) {
          S_[(size_t)i * m_ + j - 1] = S_[(size_t)i * m_ + j];
          Y_[(size_t)i * m_ + j - 1] = Y_[(size_t)i * m_ + j];
        }
      }
      --col_;
    }
    for (int i = 0; i < n_; ++i) {
      S_[(size_t)i * m_ + col_] = stp * d_[i];
      Y_[(size_t)i * m_ + col_] = g_[i] - gold_[i];
    }
    ++col_;
  }
  // M = [[-D, L^T], [L, theta S^T S]]^-1 by Gauss-Jordan with partial pivoting (2 col <= 2 m rows)
  bool form_middle() {
    const int c = col_, k = 2 * c;
    std::vector<double> A((size_t)k * k, 0.0);
    for (int a = 0; a < c; ++a)
      for (int b = 0; b < c; ++b) {
        double sty = 0.0, sts = 0.0;
        for (int i = 0; i < n_; ++i) {
          sty += S_[(size_t)i * m_ + a] * Y_[(size_t)i * m_ + b];
          sts += S_[(size_t)i * m_ + a] * S_[(size_t)i * m_ + b];
        }
        if (a == b) A[(size_t)a * k + b] = -sty;                  // -D
        if (a > b) {                                              // L (strictly lower) and its transpose
          A[(size_t)(c + a) * k + b] = sty;
          A[(size_t)b * k + (c + a)] = sty;
        }
        A[(size_t)(c + a) * k + (c + b)] = theta_ * sts;
      }
    M_.assign((size_t)k * k, 0.0);
    for (int i = 0; i < k; ++i) M_[(size_t)i * k + i] = 1.0;
    for (int p = 0; p < k; ++p) {
      int piv = p;
      for (int r = p + 1; r < k; ++r)
        if (std::fabs(A[(size_t)r * k + p]) > std::fabs(A[(size_t)piv * k + p])) piv = r;
      const double pv = A[(size_t)piv * k + p];
      if (!(std::fabs(pv) > 0.0) || !std::isfinite(pv)) return false;
      if (piv != p)
        for (int j = 0; j < k; ++j) {
          std::swap(A[(size_t)p * k + j], A[(size_t)piv * k + j]);
          std::swap(M_[(size_t)p * k + j], M_[(size_t)piv * k + j]);
        }
      const double ip = 1.0 / pv;
      for (int j = 0; j < k; ++j) { A[(size_t)p * k + j] *= ip; M_[(size_t)p * k + j] *= ip; }
      for (int r = 0; r < k; ++r) {
        if (r == p) continue;
        const double fct = A[(size_t)r * k + p];
        if (fct == 0.0) continue;
        for (int j = 0; j < k; ++j) {
          A[(size_t)r * k + j] -= fct * A[(size_t)p * k + j];
          M_[(size_t)r * k + j] -= fct * M_[(size_t)p * k + j];
        }
      }
    }
    for (double v : M_)
      if (!std::isfinite(v)) return false;
    return true;
  }
  void mvec(const double* v, double* out) const {  // out = M v
    const int k = 2 * col_;
    for (int i = 0; i < k; ++i) {
      double s = 0.0;
      for (int j = 0; j < k; ++j) s += M_[(size_t)i * k + j] * v[j];
      out[i] = s;
    }
  }

  // generalised Cauchy point: the first local minimiser of the quadratic model along P(x - t g); leaves xcp_ and c_ = W^T (xcp - x)
  void cauchy_point() {
    const int k = 2 * col_;
    const double INF = std::numeric_limits<double>::infinity();
    std::vector<double> t(n_), dd(n_), p(k, 0.0), w(k), mw(k);
    c_.assign(k, 0.0);
    std::vector<int> order;
    order.reserve(n_);
    double fp = 0.0;
    for (int i = 0; i < n_; ++i) {
      const double gi = g_[i];
      t[i] = gi < 0.0 ? (x_[i] - hi_[i]) / gi : (gi > 0.0 ? (x_[i] - lo_[i]) / gi : INF);
      dd[i] = t[i] == 0.0 ? 0.0 : -gi;
      xcp_[i] = x_[i];
      if (dd[i] != 0.0) {
        fp -= dd[i] * dd[i];
        if (k) {
          wrow(i, w.data());
          for (int j = 0; j < k; ++j) p[j] += w[j] * dd[i];
        }
        if (t[i] < INF) order.push_back(i);
      }
    }
    std::sort(order.begin(), order.end(), [&](int a, int b) { return t[a] < t[b] || (t[a] == t[b] && a < b); });
    double fpp = -theta_ * fp;
    if (k) {
      mvec(p.data(), mw.data());
      for (int j = 0; j < k; ++j) fpp -= p[j] * mw[j];
    }
    const double fpp0 = -theta_ * fp;
    double dtmin = fpp > 0.0 ? -fp / fpp : INF;
    double told = 0.0;
    size_t next = 0;
    bool free_left = fp < 0.0;  // (fp == 0: nothing moves)
    while (free_left && next < order.size()) {
      const int b = order[next];
      const double dt = t[b] - told;
      if (dtmin < dt) break;
      // variable b reaches its bound: fix it there and update the model's derivatives along the remaining path
      ++next;
      const double gb = g_[b];
      xcp_[b] = dd[b] > 0.0 ? hi_[b] : lo_[b];
      const double zb = xcp_[b] - x_[b];
      for (int j = 0; j < k; ++j) c_[j] += dt * p[j];
      double wmc = 0.0, wmp = 0.0, wmw = 0.0;
      if (k) {
        wrow(b, w.data());
        mvec(c_.data(), mw.data());
        for (int j = 0; j < k; ++j) wmc += w[j] * mw[j];
        mvec(p.data(), mw.data());
        for (int j = 0; j < k; ++j) wmp += w[j] * mw[j];
        mvec(w.data(), mw.data());
        for (int j = 0; j < k; ++j) wmw += w[j] * mw[j];
      }
      fp = fp + dt * fpp + gb * gb + theta_ * gb * zb - gb * wmc;
      fpp = fpp - theta_ * gb * gb - 2.0 * gb * wmp - gb * gb * wmw;
      fpp = std::max(std::numeric_limits<double>::epsilon() * fpp0, fpp);
      for (int j = 0; j < k; ++j) p[j] += gb * w[j];
      dd[b] = 0.0;
      told = t[b];
      dtmin = fpp > 0.0 ? -fp / fpp : INF;
      free_left = false;
      for (int i = 0; i < n_ && !free_left; ++i) free_left = dd[i] != 0.0;
      if (!(fp < 0.0)) { dtmin = 0.0; break; }
    }
    if (!free_left) dtmin = 0.0;
    dtmin = std::max(dtmin, 0.0);
    if (!std::isfinite(dtmin)) dtmin = 0.0;  // (an unbounded descent path cannot occur: every variable has both bounds in the MLE; guard anyway)
    told += dtmin;
    for (int i = 0; i < n_; ++i)
      if (dd[i] != 0.0) xcp_[i] = std::min(std::max(x_[i] + told * dd[i], lo_[i]), hi_[i]);
    for (int j = 0; j < k; ++j) c_[j] += dtmin * p[j];
  }

  // minimise the quadratic model over the variables that are free at the Cauchy point; the result is xbar (in d_ as xbar - x)
  void subspace_min(std::vector<double>& xbar) {
    const int k = 2 * col_;
    xbar = xcp_;
    std::vector<int> fr;
    for (int i = 0; i < n_; ++i)
      if (xcp_[i] > lo_[i] && xcp_[i] < hi_[i]) fr.push_back(i);
    const int nf = (int)fr.size();
    if (nf == 0 || col_ == 0) return;
    std::vector<double> mc(k), w(k), r(nf), v(k, 0.0), mv(k);
    mvec(c_.data(), mc.data());
    for (int a = 0; a < nf; ++a) {
      const int i = fr[a];
      wrow(i, w.data());
      double wmc = 0.0;
      for (int j = 0; j < k; ++j) wmc += w[j] * mc[j];
      r[a] = g_[i] + theta_ * (xcp_[i] - x_[i]) - wmc;
      for (int j = 0; j < k; ++j) v[j] += w[j] * r[a];
    }
    mvec(v.data(), mv.data());  // M W_Z^T r
    // N = I - (1 / theta) M (W_Z^T W_Z); solve N u = M W_Z^T r
    std::vector<double> WtW((size_t)k * k, 0.0), Nm((size_t)k * k, 0.0);
    for (int a = 0; a < nf; ++a) {
      wrow(fr[a], w.data());
      for (int p = 0; p < k; ++p)
        for (int q = 0; q < k; ++q) WtW[(size_t)p * k + q] += w[p] * w[q];
    }
    for (int p = 0; p < k; ++p)
      for (int q = 0; q < k; ++q) {
        double s = 0.0;
        for (int j = 0; j < k; ++j) s += M_[(size_t)p * k + j] * WtW[(size_t)j * k + q];
        Nm[(size_t)p * k + q] = (p == q ? 1.0 : 0.0) - s / theta_;
      }
    std::vector<double> u = mv;
    if (!solve_dense(Nm, u, k)) return;  // (singular: the Cauchy point stands)
    std::vector<double> du(nf);
    bool fin = true;
    for (int a = 0; a < nf; ++a) {
      wrow(fr[a], w.data());
      double wu = 0.0;
      for (int j = 0; j < k; ++j) wu += w[j] * u[j];
      du[a] = -r[a] / theta_ - wu / (theta_ * theta_);
      fin = fin && std::isfinite(du[a]);
    }
    if (!fin) return;
    // project the subspace point onto the box; if the resulting direction is not a descent direction, fall back to the
    // largest feasible fraction of the subspace step (Morales & Nocedal 2011)
    bool projected = false;
    for (int a = 0; a < nf; ++a) {
      const int i = fr[a];
      const double xi = xcp_[i] + du[a];
      xbar[i] = std::min(std::max(xi, lo_[i]), hi_[i]);
      projected = projected || xbar[i] != xi;
    }
    if (projected) {
      double dg = 0.0;
      for (int i = 0; i < n_; ++i) dg += (xbar[i] - x_[i]) * g_[i];
      if (dg > 0.0) {
        double alpha = 1.0;
        for (int a = 0; a < nf; ++a) {
          const int i = fr[a];
          if (du[a] > 0.0) alpha = std::min(alpha, (hi_[i] - xcp_[i]) / du[a]);
          else if (du[a] < 0.0) alpha = std::min(alpha, (lo_[i] - xcp_[i]) / du[a]);
        }
        alpha = std::max(alpha, 0.0);
        for (int a = 0; a < nf; ++a) {
          const int i = fr[a];
          xbar[i] = std::min(std::max(xcp_[i] + alpha * du[a], lo_[i]), hi_[i]);
        }
      }
    }
  }
  static bool solve_dense(std::vector<double>& A, std::vector<double>& b, int k) {
    for (int p = 0; p < k; ++p) {
      int piv = p;
      for (int r = p + 1; r < k; ++r)
        if (std::fabs(A[(size_t)r * k + p]) > std::fabs(A[(size_t)piv * k + p])) piv = r;
      const double pv = A[(size_t)piv * k + p];
      if (!(std::fabs(pv) > 0.0) || !std::isfinite(pv)) return false;
      if (piv != p) {
        for (int j = 0; j < k; ++j) std::swap(A[(size_t)p * k + j], A[(size_t)piv * k + j]);
        std::swap(b[p], b[piv]);
      }
      for (int r = p + 1; r < k; ++r) {
        const double fct = A[(size_t)r * k + p] / pv;
        if (fct == 0.0) continue;
        for (int j = p; j < k; ++j) A[(size_t)r * k + j] -= fct * A[(size_t)p * k + j];
        b[r] -= fct * b[p];
      }
    }
    for (int p = k - 1; p >= 0; --p) {
      double s = b[p];
      for (int j = p + 1; j < k; ++j) s -= A[(size_t)p * k + j] * b[j];
      b[p] = s / A[(size_t)p * k + p];
      if (!std::isfinite(b[p])) return false;
    }
    return true;
  }

  // search direction of iteration nit_ from (x_, f_, g_); sets the first trial point
  bool begin_iteration() {
    for (int attempt = 0; attempt < 2; ++attempt) {
      cauchy_point();
      std::vector<double> xbar;
      subspace_min(xbar);
      double gd = 0.0, dnorm2 = 0.0;
      for (int i = 0; i < n_; ++i) {
        d_[i] = xbar[i] - x_[i];
        gd += g_[i] * d_[i];
        dnorm2 += d_[i] * d_[i];
      }
      if (gd < 0.0 && dnorm2 > 0.0) {
        // largest step that keeps x + stp d inside the box (L-BFGS-B's lnsrlb: 1 on the first iteration of a CONSTRAINED problem -- the
        // model has no curvature yet --, 1e10 when no variable has a bound)
        double stpmax = 1e10;
        if (!cnstnd_) {
        } else if (nit_ == 0) {
          stpmax = 1.0;
        } else {
          for (int i = 0; i < n_; ++i) {
            const double a1 = d_[i];
            if (a1 < 0.0) {
              const double a2 = lo_[i] - x_[i];
              if (a2 >= 0.0) stpmax = 0.0;
              else if (a1 * stpmax < a2) stpmax = a2 / a1;
            } else if (a1 > 0.0) {
              const double a2 = hi_[i] - x_[i];
              if (a2 <= 0.0) stpmax = 0.0;
              else if (a1 * stpmax > a2) stpmax = a2 / a1;
            }
          }
        }
        if (stpmax > 0.0) {
          xold_ = x_;
          gold_ = g_;
          fold_ = f_;
          // first trial (lnsrlb): 1 / |d| on the first iteration unless every variable is boxed, else the quasi-Newton step 1
          const double stp0 = (nit_ == 0 && !boxed_) ? std::min(1.0 / std::sqrt(dnorm2), stpmax) : std::min(1.0, stpmax);
          ls_.start(f_, gd, stp0, 0.0, stpmax, 1e-3, 0.9, 0.1);
          ls_evals_ = 0;
          phase_ = SEARCH;
          set_trial(ls_.step());
          return true;
        }
      }
      // not a descent direction (or no room to move): forget the curvature pairs and try the projected steepest descent once
      if (col_ == 0) break;
      col_ = 0;
      theta_ = 1.0;
    }
    // the projected gradient path itself offers no descent: a stationary point of the bound-constrained problem to working precision
    status_ = proj_grad_norm() <= opt_.pgtol ? CONVERGED_PGTOL : ABNORMAL;
    return false;
  }
  // no acceptable step: back to the iterate the search started from; with curvature pairs in memory, drop them and search along the
  // projected steepest descent, otherwise give up (scipy: ABNORMAL_TERMINATION_IN_LNSRCH)
  bool line_search_failed() {
    x_ = xold_;
    g_ = gold_;
    f_ = fold_;
    xt_ = x_;
    if (col_ == 0) { status_ = ABNORMAL; return false; }
    col_ = 0;
    theta_ = 1.0;
    if (over_budget()) { status_ = STOP_MAXFUN; return false; }
    return begin_iteration();
  }

  int n_ = 0, m_ = 10, col_ = 0, nfev_ = 0, nit_ = 0, nskip_ = 0, ls_evals_ = 0;
  bool cnstnd_ = true, boxed_ = true;  // any variable bounded / every variable bounded on both sides (lnsrlb's first-step rules)
  Options opt_;
  Status status_ = RUNNING;
  Phase phase_ = FIRST;
  double f_ = 0, fold_ = 0, theta_ = 1.0;
  std::vector<double> lo_, hi_, x_, xt_, g_, gold_, xold_, d_, xcp_, S_, Y_, M_, c_;
  MoreThuente ls_;
};

}  // namespace bogp
