"""The batched elimination picks its launch plan by size (fused / split steps, steps grouped by 2 or 4 block columns, 64 x 64 or
128 x 128 update tiles, XCD-local work lists: csrc/kernels_chol.hip launch_elim_batch).  Every plan must give the
bits of sequential bogp_nll; the plans are chosen once per process from the environment, so each variant re-runs
tests/test_gpu_nll_batch.py in a process of its own with the thresholds forced down to the test sizes.
Needs a real MI355X: `pytest -m gpu`."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = {
    "split": {"BOGP_ELIM_SPLIT_BLOCKS": "1", "BOGP_ELIM_SUPER": "0"},
    "split-ungrouped": {"BOGP_ELIM_SPLIT_BLOCKS": "1", "BOGP_ELIM_SUPER": "0", "BOGP_ELIM_GROUP": "1"},
    "split-pairs": {"BOGP_ELIM_SPLIT_BLOCKS": "1", "BOGP_ELIM_SUPER": "0", "BOGP_ELIM_GROUP": "2"},
    "split-no-xcd": {"BOGP_ELIM_SPLIT_BLOCKS": "1", "BOGP_ELIM_SUPER": "0", "BOGP_ELIM_XCD": "0"},
    "split-fused-substeps": {"BOGP_ELIM_SPLIT_BLOCKS": "1", "BOGP_ELIM_SUPER": "0", "BOGP_ELIM_SUBSTEP": "1000000"},
    "super-fused-substeps": {"BOGP_ELIM_SPLIT_BLOCKS": "1", "BOGP_ELIM_SUPER": "1", "BOGP_ELIM_SUBSTEP": "1000000"},
    "split-separate-panels": {"BOGP_ELIM_SPLIT_BLOCKS": "1", "BOGP_ELIM_SUPER": "0", "BOGP_ELIM_SUBSTEP": "0"},
    "super": {"BOGP_ELIM_SPLIT_BLOCKS": "1", "BOGP_ELIM_SUPER": "1"},
    "super-pairs": {"BOGP_ELIM_SPLIT_BLOCKS": "1", "BOGP_ELIM_SUPER": "1", "BOGP_ELIM_GROUP": "2"},
    "super-no-xcd": {"BOGP_ELIM_SPLIT_BLOCKS": "1", "BOGP_ELIM_SUPER": "1", "BOGP_ELIM_XCD": "0"},
    # r05: ONE evaluation's fused steps on row pairs (k_elim_stepS: two blocks a workgroup, the next diagonal block alone in workgroup 0) at every
    # size of the elimination path, even and odd numbers of block rows -- against batches on the block kernels
    "row-pair-steps": {"BOGP_ELIM_STEP_PAIRS": "1"},
    "block-steps-only": {"BOGP_ELIM_STEP_PAIRS": "0"},
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_launch_plan_gives_the_sequential_bits(name):
    env = dict(os.environ)
    env.update(VARIANTS[name])
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_nll_batch.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "%s: %s\n%s" % (name, r.stdout[-3000:], r.stderr[-2000:])


@pytest.mark.parametrize("N", [833, 900, 1100, 1409, 1472, 1600, 1900, 1920])
def test_row_pair_steps_at_their_default_sizes(N):
    """N = 833 .. 1920 (14 .. 30 block rows): bogp_nll runs its elimination steps on row pairs by default, a batch of three on the block
    kernels (grouped steps): slot s of the batch = the bits of the sequential call, value and gradient -- and the oracle's value."""
    import numpy as np

    sys.path.insert(0, ROOT)
    from bogp import _lib
    from oracle import gp_oracle as O

    d = 6
    rng = np.random.default_rng(N)
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(np.sin(X), axis=1)
    y = ((y - y.mean()) / y.std() + 0.3 * rng.standard_normal(N)).reshape(-1, 1)
    pars = np.vstack([np.r_[np.full(d, 0.3 / d) * f, 0.8] for f in (1.0, 1.4, 0.7)])
    eng = _lib.Engine(0)
    eng.set_train(X, y)
    bl, bg, info = eng.nll_batch(O.KERNEL_MATERN52, O.MODE_NOISY, pars, 1e-2, True, 0.0, eval_grad=True)
    for s_ in range(3):
        l1, g1 = eng.nll(O.KERNEL_MATERN52, O.MODE_NOISY, pars[s_], 1e-2, True, 0.0, eval_grad=True)
        assert info[s_] == 0 and bl[s_] == l1
        np.testing.assert_array_equal(bg[s_], g1)
    ol = O.log_likelihood_concentrated(pars[0], X, y, O.KERNEL_MATERN52, O.MODE_NOISY, 1e-2, estimate_trend=True, beta=0.0, eval_grad=False)
    ol = ol[0] if isinstance(ol, tuple) else ol
    np.testing.assert_allclose(bl[0], float(np.ravel(ol)[0]), rtol=1e-9)
    eng.close()
