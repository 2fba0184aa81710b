# kernel-by-kernel timeline of ONE likelihood + gradient evaluation on the elimination path (N = 256, 512): start offsets and durations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/one_eval
mkdir -p $O
cat > /tmp/one_eval.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from bogp import _lib
eng = _lib.Engine(0)
for N, d in ((256, 10), (512, 10)):
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std() + 0.05 * rng.standard_normal(N)).reshape(-1, 1)
    par = np.r_[np.full(d, 0.2 / d), 0.9]
    eng.set_train(X, y)
    for _ in range(5): eng.nll(2, 1, par, 1e-6, True, 0.0, eval_grad=True)
    t0 = time.perf_counter()
    for _ in range(50): eng.nll(2, 1, par, 1e-6, True, 0.0, eval_grad=True)
    print("N=%d llf+grad %.1f us (host wall time)" % (N, (time.perf_counter() - t0) / 50 * 1e6))
PY
python /tmp/one_eval.py 2>&1 | grep N=
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python /tmp/one_eval.py > $O/run.log 2>&1
python - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/one_eval/tr/**/*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# split into evaluations at k_build_R
evs, cur = [], []
for r in rows:
    n = r["Kernel_Name"]
    if "k_build_R" in n and cur:
        evs.append(cur); cur = []
    cur.append(r)
evs.append(cur)
for pick in (20, 80):   # one evaluation of each size
    ev = evs[pick]
    t0 = int(ev[0]["Start_Timestamp"])
    print("evaluation %d: %d kernels, %.1f us from first start to last end" % (pick, len(ev), (int(ev[-1]["End_Timestamp"]) - t0) / 1e3))
    for r in ev:
        print("   +%7.1f us  %6.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0][:60]))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
