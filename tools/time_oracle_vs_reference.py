"""certify that the CPU baseline bench.py times (the oracle, `cpu_baseline.kind = "port"`) is not slower than
the reference it restates.  Runs ONLY in the build container (imports /root/reference); same model, same candidates,
same 1024-row chunks, same BLAS threads."""
import functools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, "/root/reference", os.path.join(ROOT, "oracle", "shims")]
import numpy as np
import warnings
warnings.filterwarnings("ignore")
from bayes_optim.surrogate import GaussianProcess
from bayes_optim.surrogate.gaussian_process.kernel import matern
from oracle import gp_oracle as O

def run(N, d, M, theta, kernel_ref, kid, label):
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    par = np.r_[np.full(d, theta), 0.9]
    gp = GaussianProcess(corr=kernel_ref, thetaL=[1e-4] * d, thetaU=[1e2] * d, nugget=1e-6)
    gp._check_data(X, y); env = {}
    gp.log_likelihood_concentrated(par, env)
    gp.theta_ = par[:d]; gp.noise_var = env["noise_var"]; gp.sigma2 = np.atleast_1d(env["sigma2"])
    gp.rho, gp.Yt, gp.C = env["rho"], env["Yt"], env["C"]; gp.compute_beta_gamma(); gp.is_fitted = True
    st = O.make_state(par, X, y, kid, O.MODE_NOISY, 1e-6)
    Xs = rng.uniform(-5, 5, size=(M, d))
    def ref():
        out = [gp.predict(Xs[i:i + 1024], eval_MSE=True) for i in range(0, M, 1024)]
        return np.vstack([o[0] for o in out]), np.vstack([o[1] for o in out])
    def port():
        return O.predict_chunked(st, Xs, 1024)
    res = {}
    for name, fn in (("reference", ref), ("oracle", port), ("reference", ref), ("oracle", port)):
        t0 = time.perf_counter(); mu, mse = fn(); res.setdefault(name, []).append(time.perf_counter() - t0); res[name + "_out"] = (mu, mse)
    dmu = np.max(np.abs(res["reference_out"][0].ravel() - res["oracle_out"][0].ravel()))
    dms = np.max(np.abs(res["reference_out"][1].ravel() - res["oracle_out"][1].ravel()))
    tr, to = min(res["reference"]), min(res["oracle"])
    print("%-34s reference %7.2f s = %7.0f cand/s | oracle %7.2f s = %7.0f cand/s | oracle/reference time %.2f | max|dmu| %.1e max|dmse| %.1e"
          % (label, tr, M / tr, to, M / to, to / tr, dmu, dms))

print("cores:", os.cpu_count(), " numpy", np.__version__)
run(512, 10, 8192, 0.02, "squared_exponential", O.KERNEL_SE, "C2 model (N=512,d=10,SE)")
run(2048, 20, 4096, 0.01, functools.partial(matern, nu=2.5), O.KERNEL_MATERN52, "C3 model (N=2048,d=20,Matern-5/2)")
