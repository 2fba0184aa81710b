"""profiles/c3_pmc.json from the PMC passes of tools/run_profiles.sh (gpurun_out/<round>/pmc_summary.txt): the committed
figure `bench.py` quotes as `roofline.traffic`, stamped with the hash of the kernel source it was measured on.
usage: python tools/make_pmc_json.py gpurun_out/r03/pmc_summary.txt r03 <commit>"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

path, rnd, commit = sys.argv[1], sys.argv[2], sys.argv[3]
vals, cur = {}, None
for line in open(path):
    if not line.startswith(" "):
        cur = line.strip()
        continue
    m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+mean=(\S+)", line)
    if m and cur and "k_contract16" in cur:
        vals[m.group(1)] = float(m.group(3))
need = ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")
missing = [k for k in need if k not in vals]
if missing:
    raise SystemExit("counters missing from %s: %s (found %s)" % (path, missing, sorted(vals)))
N, d, per_launch = 2048, 20, 65536
traffic = int(round(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024))
out = {
    "source": "rocprofv3 --pmc passes of round %s (tools/run_profiles.sh: separate runs, ONE counter group per run, kernel-trace only), one C3 sweep (tools/pmc_sweep.py); see %s_c3_pmc_summary.txt" % (rnd[1:], rnd),
    "kernel": "k_contract16<4> (v_mfma_f64_16x16x4_f64, VGPR accumulators, 2 waves/SIMD, two-deep r-tile prefetch)",
    "workload": "C3",
    "candidates_per_launch": per_launch,
    "FETCH_SIZE_KB": vals["FETCH_SIZE"],
    "WRITE_SIZE_KB": vals["WRITE_SIZE"],
    "fetch_correction": "x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B: MI355X_MICROARCH.md section HBM); WRITE_SIZE uncorrected",
    "traffic_bytes_per_launch": traffic,
    "algorithmic_bytes_per_launch": per_launch * N * 8 + 4 * N * N + per_launch * 8,
    "TCC_HIT_sum": vals.get("TCC_HIT_sum"),
    "TCC_MISS_sum": vals.get("TCC_MISS_sum"),
    "note": "traffic/algorithmic = the (nJ+1)/2 = 4.5 re-reads of the 1 GiB correlation chunk by the 8 column groups; the kernel is FP64-MFMA bound (DESIGN.md 5.2; EXPERIMENTS.md r01)",
    "SQ_VALU_MFMA_BUSY_CYCLES": vals["SQ_VALU_MFMA_BUSY_CYCLES"],
    "GRBM_GUI_ACTIVE": vals["GRBM_GUI_ACTIVE"],
    "mfma_pipe_busy_frac": round(vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (vals["GRBM_GUI_ACTIVE"] * 128.0), 4),
    "kernel_source_sha256": bench.kernel_source_hash(),
    "commit": "%s (round %s)" % (commit, rnd[1:]),
}
with open(os.path.join(ROOT, "profiles", "c3_pmc.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
