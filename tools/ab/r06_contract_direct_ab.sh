# r06: k_contract16p (persistent waves) / k_contract16d (no LDS, no barrier; BOGP_CONTRACT_PERSIST=0) against k_contract16<4> (BOGP_CONTRACT_DIRECT=0) on the workloads in $WL (default C3 C5 C2b);
# gpurun -- 'bash tools/ab/r06_contract_direct_ab.sh'.  Also checks that the two kernels give bit-identical sweep results.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_direct_ab
mkdir -p $OUT
cd $ROOT
for WLK in ${WL:-C3 C5}; do
  for MODE in ${MODES:-1_1 1_0 0_0 1_1 1_0 0_0}; do
    set -- ${MODE/_/ }
    echo "== $WLK BOGP_CONTRACT_DIRECT=$1 BOGP_CONTRACT_PERSIST=$2" | tee -a $OUT/times.txt
    BOGP_CONTRACT_DIRECT=$1 BOGP_CONTRACT_PERSIST=$2 python - $WLK <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $OUT/times.txt
import os, sys, hashlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import numpy as np, torch, bench
from bogp import _lib
w = bench.WORKLOADS[sys.argv[1]]
N, d, M = w["N"], w["d"], w["M"]
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
eng = _lib.Engine(0); eng.set_train(X, y)
eng.commit(w["kernel"], _lib.MODE_NOISY, np.r_[np.full(d, w["theta"]), 0.9], 1e-6, False, 0.0)
torch.manual_seed(0)
Xs = (torch.rand((M, d), dtype=torch.float64, device="cuda") * 10 - 5).contiguous()
eng.bind_candidates(Xs.data_ptr(), M, owner=Xs)
ts = []
for i in range(6):
    r = eng.sweep(w["acq"], float(y.min()), True); t = eng.last_timing()
    if i >= 1: ts.append((t["corr_ms"], t["contract_ms"]))
ts = np.array(ts)
mu, mse = eng.predict(True)
hh = hashlib.sha1(mse.tobytes()).hexdigest()[:16]
print("   corr %.3f  contract %.3f (min %.3f) ms ; MSE sha1 %s ; argmax %s val %s" % (np.median(ts[:, 0]), np.median(ts[:, 1]), ts[:, 1].min(), hh, r[1].tolist(), [repr(v) for v in r[0].tolist()]))
PY
  done
done
