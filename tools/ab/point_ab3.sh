for s in 0 1 2 4 8 16; do
  echo "== BOGP_POINT_SPLIT=$s (0 = the library's own rule)"
  if [ $s = 0 ]; then unset BOGP_POINT_SPLIT; else export BOGP_POINT_SPLIT=$s; fi
  python - <<'PY'
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
out = "one point %.1f us" % ((time.perf_counter() - t0) / 1000 * 1e6)
for B in (8, 32, 128):
    Xb = rng.uniform(-5, 5, size=(B, d))
    for _ in range(10): eng.point_eval_batch(Xb, acq, pl, True)
    t0 = time.perf_counter()
    for _ in range(100): eng.point_eval_batch(Xb, acq, pl, True)
    out += " | B=%d %.1f us" % (B, (time.perf_counter() - t0) / 100 * 1e6)
print(out)
PY
done
