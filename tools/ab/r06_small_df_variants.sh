# r06: library variants of k_sweep_small_df at C2-like sizes (gpurun -- 'bash tools/ab/r06_small_df_variants.sh product dfp2 ...')
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_small_df
mkdir -p $OUT
cd $ROOT
cp bayesian-optimization_amd/libbogp.so /tmp/libbogp_product.so
for TAG in "$@"; do
  if [ "$TAG" = "product" ] || [ "$TAG" = "nodf" ]; then cp /tmp/libbogp_product.so bayesian-optimization_amd/libbogp.so; else cp variants/libbogp_$TAG.so bayesian-optimization_amd/libbogp.so; fi
  DF=1; if [ "$TAG" = "nodf" ]; then DF=0; fi
  echo "== $TAG" | tee -a $OUT/variants.txt
  BOGP_SMALL_DF=$DF python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $OUT/variants.txt
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import numpy as np, torch
from bogp import _lib
for (N, d, kern, name) in ((512, 10, _lib.KERNEL_SE, "SE"), (512, 10, _lib.KERNEL_MATERN52, "M52")):
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    eng = _lib.Engine(0); eng.set_train(X, y)
    eng.commit(kern, _lib.MODE_NOISY, np.r_[np.full(d, 0.2 / d), 0.9], 1e-6, False, 0.0)
    for M in (100_000, 1_000_000):
        torch.manual_seed(0)
        Xs = (torch.rand((M, d), dtype=torch.float64, device="cuda") * 10 - 5).contiguous()
        eng.bind_candidates(Xs.data_ptr(), M, owner=Xs)
        ts = []
        for i in range(8):
            r = eng.sweep([(_lib.ACQ_EI, 0.0)], float(y.min()), True)
            t = eng.last_timing()
            if i >= 2: ts.append(t["contract_ms"])
        print("   N=%d d=%d %s M=%d: device %.4f ms (min %.4f); argmax %s %r" % (N, d, name, M, np.median(ts), min(ts), r[1].tolist(), r[0].tolist()))
    eng.close()
PY
done
cp /tmp/libbogp_product.so bayesian-optimization_amd/libbogp.so
