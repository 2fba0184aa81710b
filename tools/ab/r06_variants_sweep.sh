# r06: time one C3 sweep (or the workload in $WL) per library variant (gpurun -- 'bash tools/ab/r06_variants_sweep.sh tagA tagB ...'); "product" = the shipped library.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_variants
mkdir -p $OUT
cd $ROOT
cp bayesian-optimization_amd/libbogp.so /tmp/libbogp_product.so
for TAG in "$@"; do
  if [ "$TAG" = "product" ]; then cp /tmp/libbogp_product.so bayesian-optimization_amd/libbogp.so; else cp variants/libbogp_$TAG.so bayesian-optimization_amd/libbogp.so; fi
  for WLK in ${WL:-C3}; do
    echo "== $TAG $WLK" | tee -a $OUT/sweep_times.txt
    python - $WLK <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep_times.txt
import os, sys
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
print("   corr %.3f  contract %.3f (min %.3f) ms ; argmax %s val %s" % (np.median(ts[:, 0]), np.median(ts[:, 1]), ts[:, 1].min(), r[1].tolist(), r[0].tolist()))
PY
  done
done
cp /tmp/libbogp_product.so bayesian-optimization_amd/libbogp.so
