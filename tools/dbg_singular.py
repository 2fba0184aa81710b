import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from bogp import _lib
from oracle import gp_oracle as O
eng=_lib.Engine(0)
rng = np.random.default_rng(0)
for N, dup in ((40, (3, 17)), (200, (150, 199)), (130, (5, 129))):
    X = rng.uniform(-5, 5, size=(N, 2)); X[dup[1]] = X[dup[0]]
    y = rng.standard_normal((N, 1))
    eng.set_train(X, y)
    try:
        r=eng.nll(O.KERNEL_SE, O.MODE_NOISELESS, np.r_[0.3, 0.2], 0.0, False, 0.0, eval_grad=True)
        print(N,dup,"no raise",r[0])
    except Exception as e:
        print(N,dup,type(e).__name__,e)
    import scipy.linalg
    R=O.corr(O.KERNEL_SE, np.r_[0.3,0.2], X, X) if hasattr(O,'corr') else None
    try:
        scipy.linalg.cholesky(R, lower=True); print("  scipy: ok")
    except Exception as e: print("  scipy:", e)
