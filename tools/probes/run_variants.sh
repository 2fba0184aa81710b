# A/B of library variants on the GPU box: for every variants/libbogp_<tag>.so, put it in place of the package's library and time the
# elimination path.  The box's copy of the tree is scratch; nothing is changed in the repository.  (Experiment variants may return wrong
# values or "not positive definite": the timing loop below ignores the outcome.)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/variants
mkdir -p $O
for f in $R/variants/libbogp_*.so; do
  tag=$(basename $f .so)
  cp $f $R/bayesian-optimization_amd/libbogp.so
  echo "== $tag"
  python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import numpy as np
from bogp import _lib
eng = _lib.Engine(0)
for N, d in ((512, 10), (1024, 20), (2048, 20)):
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std() + 0.05 * rng.standard_normal(N)).reshape(-1, 1)
    par = np.r_[np.full(d, 0.2 / d), 0.9]
    eng.set_train(X, y)
    def call():
        try:
            return eng.nll(2, 1, par, 1e-6, True, 0.0, eval_grad=False)[0]
        except Exception as e:
            return str(e)[:40]
    call()
    t0 = time.perf_counter()
    for _ in range(30): r = call()
    print("N=%d: llf %.0f us (%s)" % (N, (time.perf_counter() - t0) / 30 * 1e6, r))
PY
done | tee $O/sizes.txt
