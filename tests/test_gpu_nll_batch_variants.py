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
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_launch_plan_gives_the_sequential_bits(name):
    env = dict(os.environ)
    env.update(VARIANTS[name])
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_nll_batch.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "%s: %s\n%s" % (name, r.stdout[-3000:], r.stderr[-2000:])
