"""Every BOGP_* environment switch of libbogp.so that had no test of its own (tools/README.md, "Switches", lists all of them with
their tests): the non-default setting against the default, each in a process of its own -- the library reads most switches once --
on a problem that reaches the code the switch selects.  Bit-identical where the switch only picks a schedule (two-stream sweep, the
LDS-staged contraction, helper handles of the batched likelihood, its memory cap, the phase stamps), to rounding where it picks
another algorithm for the same quantity (the likelihood without the elimination kernels).  Needs a real MI355X: `pytest -m gpu`."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import hashlib, json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from bogp import _lib
what = sys.argv[1]
out = {}
def digest(*arrs):
    h = hashlib.sha1()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()
rng = np.random.default_rng(12)
if what == "sweep":      # chunked sweep in ~6 chunks (BOGP_CHUNK_MB=8 in every process), the three trend paths under universal kriging
    for trend, d in ((0, 5), (1, 5), (2, 8)):  # constant; linear (fused into the producer); quadratic with p = 45 columns (trend rows: h->dmtrend)
        N, M = 700, 9000
        X = rng.uniform(-5, 5, (N, d)); y = np.sum(np.sin(X), axis=1); y = ((y - y.mean()) / y.std() + 0.2 * rng.standard_normal(N)).reshape(-1, 1)
        eng = _lib.Engine(0); eng.set_train(X, y)
        eng.commit(3, 1, np.r_[np.full(d, 0.3 / d), 0.9], 1e-4, True, 0.0, trend=trend)
        eng.upload_candidates(rng.uniform(-5, 5, (M, d)))
        mu, mse = eng.predict()
        best, idx, vals = eng.sweep([(0, 0.0), (3, 2.0)], float(y.min()), True, return_values=True)
        out["trend%%d" %% trend] = digest(mu, mse, best, idx, vals)
        eng.close()
elif what == "small":    # the one-launch sweep (N <= 512)
    N, d, M = 300, 4, 40000
    X = rng.uniform(-5, 5, (N, d)); y = np.sum(np.sin(X), axis=1); y = ((y - y.mean()) / y.std() + 0.2 * rng.standard_normal(N)).reshape(-1, 1)
    eng = _lib.Engine(0); eng.set_train(X, y)
    eng.commit(0, 1, np.r_[np.full(d, 0.3 / d), 0.9], 1e-4, True, 0.0)
    eng.upload_candidates(rng.uniform(-5, 5, (M, d)))
    mu, mse = eng.predict()
    best, idx = eng.sweep([(0, 0.0)], float(y.min()), True)
    out["small"] = digest(mu, mse, best, idx)
elif what == "nll":      # likelihood + gradient on the elimination path and a batch on the general path (helper handles from N = 192)
    vals = []
    for (N, d, trend) in ((400, 4, 0), (900, 6, 0), (500, 3, 1)):
        X = rng.uniform(-5, 5, (N, d)); y = np.sum(np.sin(X), axis=1); y = ((y - y.mean()) / y.std() + 0.3 * rng.standard_normal(N)).reshape(-1, 1)
        eng = _lib.Engine(0); eng.set_train(X, y)
        par = np.r_[np.full(d, 0.3 / d), 0.8]
        llf, g = eng.nll(2, 1, par, 1e-2, True, 0.0, eval_grad=True, trend=trend)
        pars = np.vstack([par * f for f in (1.0, 1.3, 0.8, 1.1, 0.9)])
        bl, bg, bi = eng.nll_batch(2, 1, pars, 1e-2, True, 0.0, eval_grad=True, trend=trend)
        vals += [float(llf)] + [float(v) for v in np.ravel(g)] + [float(v) for v in bl] + [float(v) for v in np.ravel(bg)]
        eng.close()
    out["nll"] = vals
print("RESULT " + json.dumps(out))
'''


def run(what, env):
    e = dict(os.environ)
    e["BOGP_CHUNK_MB"] = "8"
    e.update(env)
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}, what], cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


_DEFAULT = {}


def default(what):
    if what not in _DEFAULT:
        _DEFAULT[what] = run(what, {})
    return _DEFAULT[what]


@pytest.mark.parametrize("what,env", [
    ("sweep", {"BOGP_OVERLAP": "1"}),            # producer of chunk c + 1 on a second stream (incl. the trend-rows wait of r06)
    ("sweep", {"BOGP_CONTRACT_DIRECT": "0"}),    # k_contract16<4> (LDS-staged) instead of k_contract16d
    ("small", {"BOGP_SMALL_STAMPS": "1"}),       # phase stamps of k_sweep_small: a measurement aid must not change a bit
    ("nll", {"BOGP_NLL_WORKERS": "1"}),          # the batch's slots on the caller's handle only
    ("nll", {"BOGP_BATCH_MAX_MB": "1"}),         # the batched elimination in as many passes as 1 MB of workspaces allows
])  # fmt: skip
def test_schedule_switch_gives_the_default_bits(what, env):
    assert run(what, env) == default(what)


def test_likelihood_without_the_elimination_kernels_agrees_to_rounding():
    """BOGP_NLL_ELIM=0: Cholesky / recursive-doubling inverse / U U^T instead of the one-pass elimination -- another algorithm for the
    same numbers (different summation orders): 1e-9 relative on the likelihood, 1e-7 on the gradient."""
    a, b = np.array(default("nll")["nll"]), np.array(run("nll", {"BOGP_NLL_ELIM": "0"})["nll"])
    np.testing.assert_allclose(b, a, rtol=1e-7, atol=1e-9 * np.abs(a).max())
