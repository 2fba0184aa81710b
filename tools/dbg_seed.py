"""Re-run ONE fuzz problem (tools/fuzz_parity.py's generator) and print what differs at the two argmax rows: python tools/dbg_seed.py <seed> [--wide|--trend]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import numpy as np
import fuzz_parity as F
from bogp import _lib
seed = int(sys.argv[1])
eng, orc = _lib.Engine(0), F.OracleEngine()
# monkeypatch sweep to capture both sides
cap = {}
osw = orc.sweep
def osweep(acq, plugin, minimize, return_values=False):
    r = osw(acq, plugin, minimize, return_values=True)
    cap["acq"], cap["plugin"], cap["ref"] = acq, plugin, r
    return r
orc.sweep = osweep
fails, note = F.one(seed, eng, orc)
print("fails:", fails, "note:", note)
acq, plugin = cap["acq"], cap["plugin"]
b, i, v = eng.sweep(acq, plugin, True, return_values=True)
rb, ri, rv = cap["ref"]
mu, mse = eng.predict(); rmu, rmse = orc.predict()
for c in range(len(acq)):
    for row in sorted({int(i[c]), int(ri[c])}):
        print("criterion %d %s row %d: device value %r oracle %r | mu %r / %r | mse %r / %r" % (c, acq[c], row, v[c][row], rv[c][row], mu[row], np.ravel(rmu)[row], mse[row], np.ravel(rmse)[row]))
print("device nan count", int(np.isnan(v).sum()), "oracle nan count", int(np.isnan(rv).sum()))
