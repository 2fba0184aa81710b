"""Where a fused elimination step (k_elim_step) spends its time, from wall-clock stamps (100 MHz) left by a profiling build
(`make -C bayesian-optimization_amd/csrc EXTRA=-DELIM_PROFILE` after touching kernels_chol.hip / bogp_api.hip): per step k the workgroup of the
next diagonal block (entry, operands staged, panel products done, update done, stored, diagonal staged, factored) and the entry / exit of
the first and the last workgroup of the grid; the distance between a step's last stamp and the next step's first is the kernel boundary."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bogp import _lib

lib = _lib.load()
eng = _lib.Engine(0)
for N, d in [(int(a), 20) for a in sys.argv[1:]] or ((512, 10), (1024, 20), (2048, 20)):
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std() + 0.05 * rng.standard_normal(N)).reshape(-1, 1)
    par = np.r_[np.full(d, 0.2 / d), 0.9]
    eng.set_train(X, y)
    for _ in range(4): eng.nll(2, 1, par, 1e-6, True, 0.0, eval_grad=False)
    buf = (C.c_ulonglong * (64 * 16))()
    assert lib.bogp_debug_elim_stamps(buf) == 0
    s = np.array(buf[:], dtype=np.int64).reshape(64, 16) * 10  # ns
    nb = (N + 63) // 64
    print("N=%d (nb=%d): times in us relative to the diagonal workgroup's entry at step k" % (N, nb))
    print("  k | staged prod upd stored diagst factored || wg0: in out | last wg: in out || boundary to next step's first entry (diag wg / wg0)")
    for k in range(nb - 1):
        r = s[k]
        t0 = r[0]
        nxt = s[k + 1]
        first_next = min(nxt[0] if k + 2 < nb else nxt[8], nxt[8])
        print("  %2d | %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f || %5.2f %5.2f | %5.2f %5.2f || %5.2f %5.2f" % (
            k, *[(r[j] - t0) / 1e3 for j in (1, 2, 3, 4, 5, 6)], (r[8] - t0) / 1e3, (r[9] - t0) / 1e3, (r[10] - t0) / 1e3, (r[11] - t0) / 1e3,
            (nxt[0] - r[6]) / 1e3 if k + 2 < nb else float("nan"), (nxt[8] - r[6]) / 1e3))
    if N > 1900:
        print("  row-pair workgroup gridDim/2 (us from its entry): operands in | panel products done | tiles written | second update done | out")
        for k in range(0, nb - 1, 5):
            r = s[k]
            print("  %2d | %5.2f %5.2f %5.2f %5.2f %5.2f   (entry %.2f us after the diagonal workgroup's)" % (k, *[(r[j] - r[8]) / 1e3 for j in (12, 13, 14, 15, 9)], (r[8] - r[0]) / 1e3))
