# r06: k_sweep_small_df (dataflow of persistent waves) against k_sweep_small (BOGP_SMALL_DF=0): ms per sweep, device time of the launch(es);
# gpurun -- 'bash tools/ab/r06_small_df_ab.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_small_df
mkdir -p $OUT
cd $ROOT
for DF in 1 0 1 0; do
  echo "== BOGP_SMALL_DF=$DF" | tee -a $OUT/times.txt
  BOGP_SMALL_DF=$DF python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $OUT/times.txt
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import numpy as np, torch
from bogp import _lib
for (N, d, kern, name) in ((512, 10, _lib.KERNEL_SE, "SE"), (512, 10, _lib.KERNEL_MATERN52, "M52"), (384, 6, _lib.KERNEL_SE, "SE"), (500, 18, _lib.KERNEL_MATERN52, "M52")):
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    eng = _lib.Engine(0); eng.set_train(X, y)
    eng.commit(kern, _lib.MODE_NOISY, np.r_[np.full(d, 0.2 / d), 0.9], 1e-6, False, 0.0)
    for M in (100_000, 1_000_000):
        torch.manual_seed(0)
        Xs = (torch.rand((M, d), dtype=torch.float64, device="cuda") * 10 - 5).contiguous()
        eng.bind_candidates(Xs.data_ptr(), M, owner=Xs)
        ts, ws = [], []
        for i in range(8):
            t0 = time.perf_counter()
            r = eng.sweep([(_lib.ACQ_EI, 0.0)], float(y.min()), True)
            ws.append((time.perf_counter() - t0) * 1e3)
            t = eng.last_timing()
            if i >= 2: ts.append(t["contract_ms"])
        Np = (N + 31) // 32 * 32
        fl = M * (N * N + N * (3 * d + 5))
        print("   N=%d d=%d %s M=%d: device %.4f ms (min %.4f), wall %.4f ms; %.1f TF/s = %.3f of peak; argmax %s %r" % (
            N, d, name, M, np.median(ts), min(ts), np.median(ws[2:]), fl / np.median(ts) / 1e9, fl / np.median(ts) / 1e9 / 78.6, r[1].tolist(), r[0].tolist()))
    eng.close()
PY
done
