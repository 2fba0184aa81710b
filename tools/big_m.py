"""Large-M robustness: 1.2e7 candidates generated on the device, N = 512, full predict + sweep; checks the argmax against
np.argmax of the returned values and reports time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bogp import _lib
eng = _lib.Engine(0)
N, d, M = 512, 10, 12_000_003
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
eng.set_train(X, y); eng.commit(0, 1, np.r_[np.full(d, 0.02), 0.9], 1e-6)
t0 = time.perf_counter(); eng.generate_candidates([-5.0] * d, [5.0] * d, M, seed=1); t_gen = time.perf_counter() - t0
t0 = time.perf_counter(); best, idx, vals = eng.sweep([(0, 0.0), (3, 2.0)], float(y.min()), True, return_values=True); t_sw = time.perf_counter() - t0
assert idx[0] == int(np.argmax(vals[0])) and idx[1] == int(np.argmax(vals[1])) and best[0] == vals[0][idx[0]]
t0 = time.perf_counter(); mu, mse = eng.predict(); t_pr = time.perf_counter() - t0
assert np.all(np.isfinite(mu)) and np.all(mse >= 0) and len(mu) == M
print("M=%d N=%d: generate %.3fs  sweep(+q x M values to host) %.3fs  predict(+2M to host) %.3fs  argmax %s ok; timing %s" % (M, N, t_gen, t_sw, t_pr, idx.tolist(), eng.last_timing()))
