import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from conftest import load_golden
from oracle import gp_oracle as O
from bogp import _lib
g = load_golden("G25_cubic_ok_noisy")
P = g["t_m0_sk_par"]; L = g["t_m0_sk_llf"]
print(P, L)
for p_, l_ in zip(P, L):
    for N in (60, 64, 65, 68, 70):
        for kid in (5, 0):
            e2 = _lib.Engine(0); e2.set_train(g["X"][:N], g["y"][:N])
            try:
                v = e2.nll(kid, 0, p_, 0.0, False, 0.0)
            except Exception as e:
                v = str(e)[30:100]
            try:
                o = O.log_likelihood_concentrated(p_, g["X"][:N], g["y"][:N], kid, 0, 0.0, beta=0.0)
            except Exception as e:
                o = "oracle:" + str(e)[:30]
            print(N, kid, v, o)
    break
