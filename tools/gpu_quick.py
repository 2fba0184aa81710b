"""Quick GPU parity probe used while iterating (the real tests live in tests/ -m gpu)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bogp import _lib as L
from oracle import gp_oracle as O
from conftest import load_golden, state_from_golden

def rel(a, b): 
    a = np.asarray(a, float).ravel(); b = np.asarray(b, float).ravel()
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-300)))
def relmax(a, b):
    a = np.asarray(a, float).ravel(); b = np.asarray(b, float).ravel()
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))

eng = L.Engine(0)
for name in ["G1_se_sk_noisy", "G2_m32_ok_noisy", "G3_m52_sk_noisy", "G4_se_ok_noiseless", "G5_se_sk_noise_estim"]:
    g = load_golden(name); st = state_from_golden(g)
    mode, kernel = int(g["mode"]), int(g["kernel"]); est = bool(g["estimate_trend"])
    nv = float(g["noise_var"][0]) if mode == 1 else 0.0
    eng.set_train(g["X"], g["y"])
    llf = eng.commit(kernel, mode, g["par"], nv, est, 0.0)
    s = eng.get_state()
    print(name, "llf", llf, float(g["llf"]), "C", relmax(s["C"], g["C"]), "gamma", relmax(s["gamma"], g["gamma"]), "beta", s["beta"], float(g["beta"].ravel()[0]))
    eng.upload_candidates(g["Xs"])
    mu, mse = eng.predict()
    print("   mu", relmax(mu, g["mu"]), "mse", rel(mse, g["mse"]), relmax(mse, g["mse"]))
    pl = O.plugin_value(st.y, True)
    acq = [(O.ACQ_EI, 0.0), (O.ACQ_EPSILON_PI, 1e-10), (O.ACQ_UCB, 0.5), (O.ACQ_MGFI, 1.0), (O.ACQ_MGFI, 2.0), (O.ACQ_MGFI, 100.0)]
    best, idx, vals = eng.sweep(acq, pl, True, return_values=True)
    keys = ["EI", "EpsilonPI_1e-10", "UCB_0.5", "MGFI_1", "MGFI_2", "MGFI_100"]
    for k, b, i, v in zip(keys, best, idx, vals):
        print("   %-16s idx %4d ref %4d  val %.6e ref %.6e  maxrel %.2e" % (k, i, int(g["argmax_" + k][0]), b, g[k][int(g["argmax_" + k][0])], rel(v, g[k])))
    if "grad_mu" in g:
        dmu, dmse = eng.gradient(g["Xs"][0])
        print("   grad mu", relmax(dmu, g["grad_mu"][0]), "mse", relmax(dmse, g["grad_mse"][0]))
    if mode != 99:
        out = eng.nll(kernel, mode, g["par"], nv, est, 0.0, eval_grad=(kernel != 3))
        ref = O.log_likelihood_concentrated(g["par"], g["X"], g["y"], kernel, mode, nv, 0, est, 0.0, eval_grad=True)
        if kernel != 3: print("   nll", out[0], ref[0], "grad", relmax(out[1], ref[1]))

# bigger random case vs oracle
for (N, d, M, kernel) in [(512, 10, 4096, 0), (700, 7, 3000, 2), (2048, 20, 8192, 3)]:
    rng = np.random.default_rng(N)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    th = {512: 0.02, 700: 0.05, 2048: 0.01}[N]
    par = np.r_[np.full(d, th), 0.9]
    t0 = time.time(); st = O.make_state(par, X, y, kernel, 1, 1e-6); t_or = time.time() - t0
    eng.set_train(X, y); t0 = time.time(); llf = eng.commit(kernel, 1, par, 1e-6, False, 0.0); t_g = time.time() - t0
    Xs = rng.uniform(-5, 5, size=(M, d))
    eng.upload_candidates(Xs)
    t0 = time.time(); mu, mse = eng.predict(); t_p = time.time() - t0
    t0 = time.time(); rmu, rmse = O.predict_chunked(st, Xs, 1024); t_o = time.time() - t0
    print("N=%d d=%d M=%d k=%d: llf %.10g vs %.10g | mu %.2e mse %.2e | commit %.3fs (oracle %.3fs) predict %.4fs (oracle %.2fs)" % (N, d, M, kernel, llf, st.llf, relmax(mu, rmu), rel(mse, rmse), t_g, t_or, t_p, t_o), eng.last_timing())
    pl = O.plugin_value(y, True)
    acq = [(O.ACQ_EI, 0.0), (O.ACQ_MGFI, 2.0), (O.ACQ_UCB, 0.5)]
    best, idx = eng.sweep(acq, pl, True)
    obest, oidx = O.sweep(st, Xs, acq, pl, True)
    print("   sweep idx", idx, oidx, "val rel", rel(best, obest))
