"""r05: do two INDEPENDENT likelihood evaluations above N = 2048 overlap on one GPU?  Two engines (own stream, own factor buffers) on two host
threads against one engine doing the same evaluations one after the other.  usage: python tools/time_nll_two_engines.py [N d]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bogp import _lib

N, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 50)
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
th = 0.004 if N > 4096 else 0.01
pars = [np.r_[np.full(d, th * f), 0.9] for f in (1.0, 1.1, 0.9, 1.2, 0.8, 1.05, 0.95, 1.15)]
K = len(pars)
TREND = int(os.environ.get("TREND", "0"))
engs = [_lib.Engine(0) for _ in range(int(os.environ.get("WORKERS", "2")))]
for e in engs:
    e.set_train(X, y)
    e.nll(0, _lib.MODE_NOISY, pars[0], 1e-6, True, 0.0, eval_grad=True, trend=TREND)  # warm-up / allocation
ref = [engs[0].nll(0, _lib.MODE_NOISY, p, 1e-6, True, 0.0, eval_grad=True, trend=TREND) for p in pars]
t0 = time.perf_counter()
for p in pars:
    engs[0].nll(0, _lib.MODE_NOISY, p, 1e-6, True, 0.0, eval_grad=True, trend=TREND)
t_seq = (time.perf_counter() - t0) / K
out = [None] * K
def work(w):
    for i in range(w, K, len(engs)):
        out[i] = engs[w].nll(0, _lib.MODE_NOISY, pars[i], 1e-6, True, 0.0, eval_grad=True, trend=TREND)
for rep in range(2):
    ths = [threading.Thread(target=work, args=(w,)) for w in range(len(engs))]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    t_par = (time.perf_counter() - t0) / K
same = all(o[0] == r[0] and np.array_equal(o[1], r[1]) for o, r in zip(out, ref))
print("N = %d, d = %d, trend %d: sequential %.3f ms per llf + gradient; %d engines on %d threads %.3f ms per evaluation (x%.2f); results bit-identical: %s" % (
    N, d, TREND, t_seq * 1e3, len(engs), len(engs), t_par * 1e3, t_seq / t_par, same))
