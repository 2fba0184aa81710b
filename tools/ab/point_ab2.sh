for nc in 22 12; do for s in 4 8 16; do
  echo "== BOGP_POINT_NC=$nc BOGP_POINT_SPLIT=$s"
  BOGP_POINT_NC=$nc BOGP_POINT_SPLIT=$s python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from bogp import _lib
N, d = 2048, 20
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
eng = _lib.Engine(0); eng.set_train(X, y)
eng.commit(_lib.KERNEL_MATERN52, _lib.MODE_NOISY, np.r_[np.full(d, 0.2 / d), 0.9], 1e-6, False, 0.0)
x = rng.uniform(-5, 5, size=d); acq = [(_lib.ACQ_EI, 0.0)]; pl = float(y.min())
for _ in range(50): eng.point_eval(x, acq, pl, True)
t0 = time.perf_counter()
for _ in range(1000): eng.point_eval(x, acq, pl, True)
print("one point: %.1f us" % ((time.perf_counter() - t0) / 1000 * 1e6))
PY
done; done
for nc in 22 12; do
  echo "== BOGP_POINT_NC=$nc batch (no split)"
  BOGP_POINT_NC=$nc python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from bogp import _lib
N, d = 2048, 20
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
eng = _lib.Engine(0); eng.set_train(X, y)
eng.commit(_lib.KERNEL_MATERN52, _lib.MODE_NOISY, np.r_[np.full(d, 0.2 / d), 0.9], 1e-6, False, 0.0)
acq = [(_lib.ACQ_EI, 0.0)]; pl = float(y.min())
for B in (32, 128):
    Xb = rng.uniform(-5, 5, size=(B, d))
    for _ in range(10): eng.point_eval_batch(Xb, acq, pl, True)
    t0 = time.perf_counter()
    for _ in range(100): eng.point_eval_batch(Xb, acq, pl, True)
    print("B=%d: %.1f us" % (B, (time.perf_counter() - t0) / 100 * 1e6))
PY
done
