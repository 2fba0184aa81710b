# durations of the within-panel update launches (k_chol_update) of ONE factorisation at N = 8192, 128-tile path, with their grid sizes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/bigchol
mkdir -p $O
cat > /tmp/big1.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from bogp import _lib
N, d = 8192, 50
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
par = np.r_[np.full(d, 0.004), 0.9]
eng = _lib.Engine(0); eng.set_train(X, y)
for _ in range(3): eng.nll(0, 1, par, 1e-6, False, 0.0, eval_grad=bool(int(os.environ.get("GRAD", "0"))))
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- env BOGP_BIG_CHOL=${BIG:-0} python /tmp/big1.py > $O/run.log 2>&1
python - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/bigchol/tr/**/*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last evaluation: from the last k_build_R on
start = max(i for i, r in enumerate(rows) if "k_build_R" in r["Kernel_Name"])
ev = rows[start:]
t0 = int(ev[0]["Start_Timestamp"])
tot = {}
for r in ev:
    n = r["Kernel_Name"].split("(")[0].replace("bogp::", "").replace("void ", "")[:28]
    tot.setdefault(n, [0, 0.0]); tot[n][0] += 1; tot[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("one llf evaluation at N = 8192: %.2f ms from first start to last end" % ((int(ev[-1]["End_Timestamp"]) - t0) / 1e6))
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]): print("   %-28s %4d launches %9.1f us" % (n, c, t))
print("k_chol_update launches in order: workgroups / us")
print(" ".join("%d/%.0f" % (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) if "Grid_Size_X" in r else -1, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in ev if "k_chol_update" in r["Kernel_Name"]))
print("k_chol_panel + k_chol_first + k_mm128 in order (name/us):")
print(" ".join("%s/%.0f" % (r["Kernel_Name"].split("(")[0][-6:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in ev if any(x in r["Kernel_Name"] for x in ("k_chol_panel", "k_chol_first", "k_mm128"))))
print("k_mm128 launches in order: us")
print(" ".join("%.0f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in ev if "k_mm128" in r["Kernel_Name"]))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
