"""Full-size argmax fixtures: the REFERENCE's posterior over every candidate of BASELINE.json's C2..C5 workloads.

Run (build container only; hours of CPU):   python oracle/make_fullsize_golden.py [C2 C3 C4 C5]

`tests/test_gpu_parity.py::test_full_size_properties` sweeps M = 1e5 / 1e6 / 5e5 seeded candidates per configuration
on the device; until now nothing held the reference's answer for those very sweeps ("weak" item 1).
This script imports `/root/reference/bayes_optim`, pins the model state exactly as `oracle/make_golden.py` does,
and pushes ALL M candidates through the reference's own `GaussianProcess.predict(eval_MSE=True)` in 1024-row chunks
(`gpr.py:486-510`; the caller has to chunk, `:513-535` is dead).  The q criteria are then evaluated with the oracle's
vectorised closed forms (pinned row by row against the reference's classes by G1..G7) and, for the stored top rows,
once more through the reference's own acquisition classes one row at a time (`acquisition_fun.py:153-176,265-290`).

Stored per configuration (tests/golden/G20_c2_full.npz ... G23_c5_full.npz; a few hundred KB each):
  * `top_idx[q,16]`, `top_val[q,16]`: the 16 best candidates of each criterion (stable order: value descending,
    ties -> lowest index, i.e. repeated np.argmax), `top_mu`, `top_mse` of those rows, `gap[q]` = relative gap between
    the winner and the runner-up (how much rounding the index can absorb), `ref_rowwise[q,16]` = the reference's
    own classes on those rows;
  * `slice_mu`, `slice_mse` of the fixed slice rows `slice_rows` (4096 rows: every (M // 4096)-th);
  * `count_pos[q]` = how many candidates have a strictly positive criterion value, and `sum_mu`, `sum_mse`
    (float64 sums over all M rows: a checksum of the whole posterior, compared to 1e-9 relative).
The workload generator below is byte-for-byte the one the GPU test uses (`tests/support/workloads.py`).
Intermediate mu/MSE of all M rows are cached under /tmp/bogp_fullsize so that an interrupted run resumes.
"""
import functools
import os
import sys
import time
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "shims"))

import numpy as np  # noqa: E402
import scipy  # noqa: E402

import bayes_optim  # noqa: E402,F401
from bayes_optim.acquisition import acquisition_fun as AF  # noqa: E402
from bayes_optim.surrogate import GaussianProcess  # noqa: E402
from bayes_optim.surrogate.gaussian_process.kernel import matern  # noqa: E402

from oracle import gp_oracle as O  # noqa: E402
from oracle.make_golden import pin  # noqa: E402
from tests.support.workloads import FULL_SIZE, full_size_problem  # noqa: E402

warnings.filterwarnings("ignore")
OUT = os.path.join(ROOT, "tests", "golden")
CACHE = "/tmp/bogp_fullsize"
FILES = {"C2": "G20_c2_full", "C3": "G21_c3_full", "C4": "G22_c4_full", "C5": "G23_c5_full"}
TOPK = 16


def reference_model(kernel, d):
    corr = {O.KERNEL_SE: "squared_exponential", O.KERNEL_MATERN52: functools.partial(matern, nu=2.5)}[kernel]
    return GaussianProcess(corr=corr, thetaL=[1e-5] * d, thetaU=[1e2] * d, nugget=1e-6)


def posterior_all_rows(cfg, gp, Xs):
    os.makedirs(CACHE, exist_ok=True)
    M = len(Xs)
    key = "N%d_d%d_M%d_k%d" % (gp.X.shape[0], gp.X.shape[1], M, FULL_SIZE[cfg]["kernel"])  # C3 and C4 share the posterior
    fmu, fmse, fpos = (os.path.join(CACHE, "%s_%s.npy" % (key, k)) for k in ("mu", "mse", "pos"))
    if os.path.exists(fpos):
        mu, mse, pos = np.load(fmu), np.load(fmse), int(np.load(fpos))
    else:
        mu, mse, pos = np.empty(M), np.empty(M), 0
    chunk = 1024 if gp.X.shape[0] <= 2048 else 256  # N = 8192: 1024 rows would hold a 3.4 GB |dx| temporary
    t0 = time.time()
    while pos < M:
        b = min(M, pos + chunk)
        m, s = gp.predict(Xs[pos:b], eval_MSE=True)
        mu[pos:b], mse[pos:b] = m[:, 0], s[:, 0]
        pos = b
        if (pos // chunk) % 64 == 0 or pos == M:
            np.save(fmu, mu), np.save(fmse, mse), np.save(fpos, np.array(pos))
            print("%s: %d / %d rows, %.0f s" % (cfg, pos, M, time.time() - t0), flush=True)
    return mu, mse


def reference_rowwise(gp, acq, plugin, Xrows):
    out = np.empty((len(acq), Xrows.shape[1]))
    for k, (a, p) in enumerate(acq):
        c = {O.ACQ_EI: lambda: AF.EI(model=gp, minimize=True, plugin=plugin),
             O.ACQ_UCB: lambda: AF.UCB(model=gp, minimize=True, alpha=p),
             O.ACQ_MGFI: lambda: AF.MGFI(model=gp, minimize=True, plugin=plugin, t=p)}[a]()  # fmt: skip
        for i, x in enumerate(Xrows[k]):
            out[k, i] = float(np.asarray(c(x.reshape(1, -1)), dtype=float).ravel()[0])
    return out


def make(cfg):
    w = FULL_SIZE[cfg]
    X, y, par, Xs = full_size_problem(cfg)
    N, d, M = w["N"], w["d"], w["M"]
    gp = reference_model(w["kernel"], d)
    llf = pin(gp, X, y, par)
    mu, mse = posterior_all_rows(cfg, gp, Xs)
    plugin = O.plugin_value(y, True)
    sigma2 = float(gp.sigma2[0])
    q = len(w["acq"])
    top_idx = np.empty((q, TOPK), np.int64)
    top_val, gap, count_pos = np.empty((q, TOPK)), np.empty(q), np.empty(q, np.int64)
    for k, (a, p) in enumerate(w["acq"]):
        v = O.acquisition(a, p, mu, mse, plugin, sigma2, True)
        assert not np.isnan(v).any()
        order = np.argsort(-v, kind="stable")[:TOPK]
        assert order[0] == int(np.argmax(v))
        top_idx[k], top_val[k] = order, v[order]
        gap[k] = (v[order[0]] - v[order[1]]) / abs(v[order[0]]) if v[order[0]] != 0 else 0.0
        count_pos[k] = int(np.count_nonzero(v > 0))
    rows = np.arange(0, M, M // 4096)[:4096]
    # the oracle restatement on the slice: certifies once more, on THIS workload, that it is the reference
    st = O.make_state(par, X, y, w["kernel"], O.MODE_NOISY, 1e-6)
    omu, omse = O.predict_chunked(st, Xs[rows[:512]], 256)
    np.testing.assert_allclose(omu[:, 0], mu[rows[:512]], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(omse[:, 0], mse[rows[:512]], rtol=1e-9, atol=1e-14)
    ref_rows = reference_rowwise(gp, w["acq"], plugin, Xs[top_idx])
    np.testing.assert_allclose(ref_rows, top_val, rtol=1e-8, atol=1e-300)  # batched vs single-row BLAS rounding (x z^3)
    out = dict(
        cfg=cfg, N=N, d=d, M=M, kernel=w["kernel"], par=par, llf=float(llf), plugin=plugin, sigma2=sigma2,
        acq=np.array(w["acq"], float), top_idx=top_idx, top_val=top_val, top_mu=mu[top_idx], top_mse=mse[top_idx],
        gap=gap, ref_rowwise=ref_rows, count_pos=count_pos, slice_rows=rows, slice_mu=mu[rows], slice_mse=mse[rows],
        sum_mu=float(np.sum(mu)), sum_mse=float(np.sum(mse)), x_checksum=float(np.sum(Xs[::997])),
        numpy=np.__version__, scipy=scipy.__version__,
    )  # fmt: skip
    np.savez_compressed(os.path.join(OUT, FILES[cfg] + ".npz"), **out)
    print(cfg, "argmax", top_idx[:, 0].tolist(), "gap", gap.tolist(), "positive", count_pos.tolist(), flush=True)


if __name__ == "__main__":
    for c in sys.argv[1:] or ["C2", "C3", "C4", "C5"]:
        make(c)
