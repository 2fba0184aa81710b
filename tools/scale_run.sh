#!/bin/bash
# The first 8-GPU run as ONE command.  No 1 -> 8 curve has been measured in any round: the build container has
# no GPU and gpurun boxes have one; the driver's SCALE / MULTICHIP records are `skipped`.  On a node with 8 MI355X:
#     bash tools/scale_run.sh            # C3 weak (1e6 candidates per GPU), then C4 / C5 strong (8e6 / 4e6 in all)
# For every N in 1 2 4 8 it runs bench.py exactly as the driver does (bench.py launches its own ranks when RANK is unset), asserts
#   * rccl_world == N          (the library's own RCCL communicator really spans the N ranks -- not the torch.distributed fall-back),
#   * one JSON line, exchange through bogp_exchange_argmax,
#   * STRONG scaling: the global argmax (value, global row) is the same for every N (the same candidate grid, contiguous shards),
# and prints value, ms/step and the efficiency against N = 1 (weak: value_N / (N value_1); strong: the same, total work fixed).
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=${OUT:-gpurun_out/scale}
mkdir -p "$OUT"
NG=$(python -c "import torch; print(torch.cuda.device_count())")
echo "# $(date -u +%FT%TZ)  GPUs visible: $NG"
for spec in "C3 weak" "C4 strong" "C5 strong"; do
  set -- $spec; W=$1; S=$2
  for N in 1 2 4 8; do
    if [ "$N" -gt "$NG" ]; then echo "$W $S N=$N: skipped ($NG GPUs visible)"; continue; fi
    python bench.py --gpus $N --steps ${STEPS:-10} --warmup 2 --workload $W --scaling $S --no-cpu --no-seeds > "$OUT/${W}_${S}_$N.json" 2> "$OUT/${W}_${S}_$N.err" \
      || { echo "$W $S N=$N: bench.py failed (see $OUT/${W}_${S}_$N.err)"; continue; }
  done
  python - "$OUT" "$W" "$S" <<'PY'
import json, os, sys
out, w, s = sys.argv[1:4]
rows = {}
for n in (1, 2, 4, 8):
    f = os.path.join(out, "%s_%s_%d.json" % (w, s, n))
    if not os.path.exists(f) or not os.path.getsize(f):
        continue
    lines = [ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "%s: %d JSON lines" % (f, len(lines))
    j = json.loads(lines[0])
    assert j["n_gpus"] == n and j["rccl_world"] == n, "%s: n_gpus %s, rccl_world %s" % (f, j["n_gpus"], j["rccl_world"])
    assert "bogp_exchange_argmax over RCCL" in j["exchange"], j["exchange"]
    rows[n] = j
if 1 in rows:
    for n, j in sorted(rows.items()):
        eff = j["value"] / (n * rows[1]["value"])
        print("%s %-6s N=%d  %12.0f candidates/s  %8.3f ms/step  efficiency %.3f  argmax %s" % (w, s, n, j["value"], j["ms_per_step"], eff, j["argmax"]))
    if s == "strong":
        assert all(j["argmax"] == rows[1]["argmax"] for j in rows.values()), "the global argmax differs between GPU counts"
        print("%s strong: the same global argmax for N in %s" % (w, sorted(rows)))
PY
done
