"""Shader-clock phase breakdown of k_nll_small (build csrc with -DNS_PROFILE): prologue, the three phases of a step as thread 0
sees them (wait for W = the diagonal factor of the other thread; wait for the panel = the publishers' trsm; its own update),
epilogue.  Reads the extra words the profiled kernel leaves in the pinned scalar block."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bogp import _lib  # noqa: E402

lib = _lib.load()
lib.bogp_debug_fit_scalars.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
eng = _lib.Engine(0)
for N, d in ((16, 2), (32, 5), (64, 5), (128, 5), (128, 10)):
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(X**2, axis=1)
    y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    eng.set_train(X, y)
    par = np.r_[np.full(d, 0.2 / d), 0.9]
    for grad in (False, True):
        for _ in range(5):
            eng.nll(2, 1, par, 1e-6, True, 0.0, eval_grad=grad)
        blk = (C.c_double * 64)()
        lib.bogp_debug_fit_scalars(eng._h, blk)
        pro, tA, tB, tC, epi = blk[20], blk[21], blk[22], blk[23], blk[24]
        nb = (N + 3) // 4
        print("N=%3d d=%2d grad=%d: prologue %6.0f clk | per step: wait-W %5.0f  wait-panel %5.0f  update %5.0f | epilogue %6.0f clk | total %7.0f clk"
              % (N, d, grad, pro, tA / nb, tB / nb, tC / nb, epi, pro + tA + tB + tC + epi))

