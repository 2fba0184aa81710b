"""llf / llf+gradient / commit at N = 4096 and 8192 (d = 50), 128-tile path vs the 64-block path (BOGP_NO_BIG_FIT=1); --big-only: the 128-tile path alone."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bogp import _lib
BIG_ONLY = "--big-only" in sys.argv
SHA = "--sha" in sys.argv  # + sha1 of (llf, gradient, committed factor): bit-identity of library variants
for N in [int(a) for a in sys.argv[1:] if not a.startswith("--")] or (4096, 8192):
    d = 50
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    par = np.r_[np.full(d, 0.004), 0.9]
    for tag, flag in (("128-tile", "0"),) + (() if BIG_ONLY else (("64-block", "1"),)):
        os.environ["BOGP_NO_BIG_FIT"] = flag
        eng = _lib.Engine(0)
        eng.set_train(X, y)
        def call(fn):  # (experiments that break the numerics on purpose still queue and time the same launches)
            try:
                return fn()
            except _lib.BogpError:
                return (float("nan"), np.zeros(1))
        call(lambda: eng.nll(0, 1, par, 1e-6, False, 0.0, eval_grad=True))
        res = []
        for eg in (False, True):
            t0 = time.perf_counter()
            for _ in range(5): out = call(lambda: eng.nll(0, 1, par, 1e-6, False, 0.0, eval_grad=eg))
            res.append((time.perf_counter() - t0) / 5 * 1e3)
        if not eg or not isinstance(out, tuple): out = (out, np.zeros(1))
        t0 = time.perf_counter(); call(lambda: eng.commit(0, 1, par, 1e-6, False, 0.0)); tc = (time.perf_counter() - t0) * 1e3
        sha = ""
        if SHA:
            import hashlib
            hsh = hashlib.sha1(np.float64(out[0]).tobytes() + np.ascontiguousarray(out[1]).tobytes())
            try:
                hsh.update(np.ascontiguousarray(eng.get_state()["C"]).tobytes())
            except _lib.BogpError:
                pass
            sha = "  sha1 " + hsh.hexdigest()[:16]
        print("N=%d %-9s llf %.2f ms  llf+grad %.2f ms  commit %.2f ms  (llf %.6f, |grad| %.6e)%s" % (N, tag, res[0], res[1], tc, out[0], np.abs(out[1]).sum(), sha))
        eng.close()
