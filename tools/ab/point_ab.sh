# A/B of the row-block split count of k_point_tri at C3 size (one point): BOGP_POINT_SPLIT = workgroups per 64-row block
for s in 1 2 4 8 16 32; do
  echo "== BOGP_POINT_SPLIT=$s"
  BOGP_POINT_SPLIT=$s python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from bogp import _lib
for N, d in ((2048, 20), (512, 10)):
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    eng = _lib.Engine(0); eng.set_train(X, y)
    eng.commit(_lib.KERNEL_MATERN52, _lib.MODE_NOISY, np.r_[np.full(d, 0.2 / d), 0.9], 1e-6, False, 0.0)
    x = rng.uniform(-5, 5, size=d); acq = [(_lib.ACQ_EI, 0.0)]; pl = float(y.min())
    for _ in range(50): eng.point_eval(x, acq, pl, True)
    t0 = time.perf_counter()
    for _ in range(1000): eng.point_eval(x, acq, pl, True)
    print("N=%d d=%d: engine.point_eval %.1f us" % (N, d, (time.perf_counter() - t0) / 1000 * 1e6))
    eng.close()
PY
done
