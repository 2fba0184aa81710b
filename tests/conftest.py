import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built library (*.so is git-ignored): build it once, exactly as __graft_entry__.build() does
    if not os.path.exists(os.path.join(ROOT, "bayesian-optimization_amd", "libbogp.so")):
        import subprocess

        subprocess.run(["make", "-C", os.path.join(ROOT, "bayesian-optimization_amd", "csrc")], check=True, capture_output=True)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def state_from_golden(g):
    """Rebuild an oracle GPState from a golden file's (X, y, par, kernel, mode) -- via the ORACLE's own
    likelihood call, so that comparing its C/gamma/... with the stored reference values is a real check."""
    from oracle import gp_oracle as O

    mode, kernel = int(g["mode"]), int(g["kernel"])
    est = bool(g["estimate_trend"]) if "estimate_trend" in g else False
    nv = float(g["noise_var"][0]) if mode == O.MODE_NOISY else 0.0
    trend = int(g["trend"]) if "trend" in g else O.TREND_CONSTANT
    beta = None if est else (np.asarray(g["beta"], float).ravel() if trend != O.TREND_CONSTANT else 0.0)
    return O.make_state(g["par"], g["X"], g["y"], kernel, mode, noise_var=nv, trend=trend, estimate_trend=est, beta=beta)
