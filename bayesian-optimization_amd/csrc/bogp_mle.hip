// bogp_mle.hip -- the restarts of the hyper-parameter MLE in lock step (bogp_mle_batch), and the optimiser on a host callback
// (bogp_lbfgsb_minimize) so that it can be tested against scipy without a GPU.
//
// Replaces the restart loop of GaussianProcess._optimize_hyperparameter (gpr.py:1127-1162): there, `random_start` runs of
// fmin_l_bfgs_b follow each other, each objective call being one likelihood + gradient evaluation (gpr.py:1113-1123).  Here the R
// runs advance together -- every round gathers the trial points of the runs that are still active, evaluates them with ONE
// bogp_nll_batch call and hands (f, g) back to each run's own optimiser state (bogp_lbfgsb.h).  What is kept from the reference:
// the search is over log10(parameters) inside log10 bounds, the objective is -llf(10^x), and the gradient handed to the optimiser is
// -d llf / d par, NOT multiplied by ln(10) 10^x; a parameter vector the likelihood rejects gives +inf with a
// zero gradient (gpr.py:946-947, 981-982).  What changes, and why it is opt-in (GaussianProcess(restart_batch=R)): the shared
// evaluation budget is consumed by all runs at once and the stagnation counter `wait_iter` has nothing to count -- no run waits for
// another -- so R runs always start, where the sequential loop may stop after wait_iter of them bring no improvement.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../../include/bogp.h"
#include "bogp_handle.h"
#include "bogp_lbfgsb.h"

using namespace bogp;

extern "C" int bogp_lbfgsb_minimize(int n, int m, double* x, const double* lo, const double* hi, double factr, double pgtol, int maxfun,
                                    int maxiter, bogp_objective_fn fn, void* user, double* f, int* nfev, int* nit, int* status) {
  if (n <= 0 || !x || !lo || !hi || !fn) return BOGP_ERR_INVALID;
  for (int i = 0; i < n; ++i)
    if (!(lo[i] <= hi[i])) return BOGP_ERR_INVALID;
  Lbfgsb::Options opt;
  opt.m = m > 0 ? m : 10;
  opt.factr = factr;
  opt.pgtol = pgtol;
  opt.maxfun = maxfun > 0 ? maxfun : 15000;
  opt.maxiter = maxiter > 0 ? maxiter : 15000;
  Lbfgsb o;
  o.start(n, x, lo, hi, opt);
  std::vector<double> g((size_t)n);
  while (o.running()) {
    double fv = 0.0;
    fn(o.x(), n, &fv, g.data(), user);
    o.tell(fv, g.data());
  }
  memcpy(x, o.best_x(), (size_t)n * sizeof(double));
  if (f) *f = o.best_f();
  if (nfev) *nfev = o.nfev();
  if (nit) *nit = o.nit();
  if (status) *status = (int)o.status();
  return BOGP_OK;
}

extern "C" int bogp_mle_batch(bogp_handle* h, int kernel, int mode, int restricted, int R, const double* x0, int n_par, const double* lo,
                              const double* hi, double noise_var, int trend, int estimate_trend, double beta, int eval_budget, int m,
                              double factr, double pgtol, int flags, int prune_reserve, double* xopt, double* fopt, int* n_evals, int* status,
                              int* n_rounds) {
  if (!h) return BOGP_ERR_INVALID;
  if (!x0 || !lo || !hi || !xopt || !fopt || R <= 0 || n_par <= 0) FAIL(h, BOGP_ERR_INVALID, "bogp_mle_batch: x0 / lo / hi / xopt / fopt must be non-null, R and n_par > 0");
  for (int i = 0; i < n_par; ++i)
    if (!(lo[i] <= hi[i]) || !std::isfinite(lo[i]) || !std::isfinite(hi[i])) FAIL(h, BOGP_ERR_INVALID, "bogp_mle_batch: bounds must be finite with lo <= hi (parameter %d)", i);
  if (!h->dX) FAIL(h, BOGP_ERR_INVALID, "no training set: call bogp_set_train first");
  long evals = 0;
  Lbfgsb::Options opt;
  opt.m = m > 0 ? m : 10;
  opt.factr = factr > 0 ? factr : 1e7;
  opt.pgtol = pgtol > 0 ? pgtol : 1e-5;
  opt.maxfun = eval_budget > 0 ? eval_budget : 15000;
  opt.shared_evals = &evals;
  opt.shared_budget = eval_budget > 0 ? eval_budget : 0;
  std::vector<Lbfgsb> runs((size_t)R);
  for (int r = 0; r < R; ++r) runs[(size_t)r].start(n_par, x0 + (size_t)r * n_par, lo, hi, opt);
  std::vector<int> act;
  std::vector<double> par, llf, grad, g((size_t)n_par);
  std::vector<int> info;
  int rounds = 0;
  for (;;) {
    act.clear();
    for (int r = 0; r < R; ++r)
      if (runs[(size_t)r].running()) act.push_back(r);
    // A shared budget spread over R runs starves all of them alike.  With prune_reserve > 0 the budget is concentrated as it runs out:
    // while fewer than prune_reserve evaluations per active run are left, the run with the worst value so far is stopped (it keeps
    // its last iterate), so that the leading runs can still converge -- what the sequential loop gives its first restarts.
    if (prune_reserve > 0 && opt.shared_budget > 0) {
      while (act.size() > 1 && opt.shared_budget - evals < (long)prune_reserve * (long)act.size()) {
        size_t worst = 0;
        for (size_t i = 1; i < act.size(); ++i) {
          const double fi = runs[(size_t)act[i]].best_f(), fw = runs[(size_t)act[worst]].best_f();
          if (fi > fw || (fi == fw && act[i] > act[worst]) || (std::isnan(fi) && !std::isnan(fw))) worst = i;
        }
        runs[(size_t)act[worst]].stop(Lbfgsb::STOP_MAXFUN);
        act.erase(act.begin() + (long)worst);
      }
    }
    if (act.empty()) break;
    const int P = (int)act.size();
    par.resize((size_t)P * n_par);
    llf.resize((size_t)P);
    grad.assign((size_t)P * n_par, 0.0);
    info.assign((size_t)P, 0);
    for (int i = 0; i < P; ++i) {
      const double* x = runs[(size_t)act[(size_t)i]].x();
      for (int k = 0; k < n_par; ++k) par[(size_t)i * n_par + k] = std::pow(10.0, x[k]);  // 10.0 ** log10param (gpr.py:1116)
    }
    if (!restricted) {
      const int rc = bogp_nll_batch(h, kernel, mode, P, par.data(), n_par, noise_var, trend, estimate_trend, beta, llf.data(), grad.data(), info.data());
      if (rc != BOGP_OK) return rc;
    } else {
      // REML (gpr.py:813-918): its device path is the one-evaluation one; the restarts still share this loop (no host optimiser
      // overhead between evaluations).  exp(llf) > 1 is -inf WITH the gradient of the finite value, as the reference returns it
      // r05: dealt over the caller's handle and the library's helper handles (bogp_batch.hip: nll_team), one host thread each
      auto run_slot = [&](bogp_handle* hh, int i) -> int {
        const double* p = par.data() + (size_t)i * n_par;
        bool ok = true;
        for (int k = 0; k < n_par; ++k) ok = ok && std::isfinite(p[k]) && p[k] > 0;
        int rc = BOGP_ERR_INVALID;
        if (ok) rc = bogp_nll_restricted(hh, kernel, mode, p, n_par, noise_var, trend, estimate_trend, beta, &llf[(size_t)i], grad.data() + (size_t)i * n_par);
        if (rc != BOGP_OK && rc != BOGP_ERR_LLF_POSITIVE)
          for (int k = 0; k < n_par; ++k) grad[(size_t)i * n_par + k] = 0.0;
        info[(size_t)i] = rc;
        return rc;
      };
      auto fatal = [](int rc) { return rc == BOGP_ERR_HIP || rc == BOGP_ERR_UNSUPPORTED || rc == BOGP_ERR_NO_DEVICE; };
      std::vector<bogp_handle*> team = nll_team(h, P);
      const int W = (int)team.size();
      std::vector<int> rcs((size_t)W, BOGP_OK);
      std::vector<std::thread> threads;
      for (int w = 1; w < W; ++w)
        threads.emplace_back([&, w] {
          for (int i = w; i < P && !fatal(rcs[(size_t)w]); i += W) rcs[(size_t)w] = run_slot(team[(size_t)w], i);
        });
      for (int i = 0; i < P && !fatal(rcs[0]); i += W) rcs[0] = run_slot(h, i);
      for (auto& t : threads) t.join();
      for (int w = 0; w < W; ++w)
        if (fatal(rcs[(size_t)w])) return rcs[(size_t)w];
    }
    ++rounds;
    evals += P;
    for (int i = 0; i < P; ++i) {
      const double fv = info[(size_t)i] == BOGP_OK ? -1.0 * llf[(size_t)i] : std::numeric_limits<double>::infinity();
      for (int k = 0; k < n_par; ++k) g[(size_t)k] = -1.0 * grad[(size_t)i * n_par + k];
      if (flags & BOGP_MLE_CHAIN_RULE)  // d / d log10(par) = ln(10) par d / d par: the gradient OF the function being minimised
        for (int k = 0; k < n_par; ++k) g[(size_t)k] *= 2.302585092994046 * par[(size_t)i * n_par + k];
      runs[(size_t)act[(size_t)i]].tell(fv, g.data());
    }
  }
  for (int r = 0; r < R; ++r) {
    const Lbfgsb& o = runs[(size_t)r];
    memcpy(xopt + (size_t)r * n_par, o.best_x(), (size_t)n_par * sizeof(double));
    fopt[r] = o.best_f();
    if (n_evals) n_evals[r] = o.nfev();
    if (status) status[r] = (int)o.status();
  }
  if (n_rounds) *n_rounds = rounds;
  return BOGP_OK;
}
