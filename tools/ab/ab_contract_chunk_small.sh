# r04 A/B asked for by the r03 review: correlation-chunk sizes that fit the 256-MiB Infinity Cache (BOGP_CHUNK_MB = 64 / 128 / 192)
# against the default 1 GiB, C3.  Per setting: ms/step + kernel times from bench.py (no profiler), then three PMC passes of ONE
# sweep (tools/pmc_sweep.py; separate runs, kernel-trace only): FETCH_SIZE, WRITE_SIZE, GRBM_GUI_ACTIVE (+ MFMA busy).
# usage (GPU box): bash tools/ab/ab_contract_chunk_small.sh > gpurun_out/r04_contract_chunk_small_ab.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/chunk_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mb in 1024 192 128 64; do
  echo "== BOGP_CHUNK_MB=$mb"
  BOGP_CHUNK_MB=$mb python $ROOT/bench.py --workload C3 --no-cpu --no-seeds --steps 10 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d['roofline']
print('  ms/step %.3f (median %.3f min %.3f max %.3f)  contract frac %.4f  launches/step %d  avg launch %.3f ms  kernels/step %s' % (d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'], d['step_ms']['max'], r['frac'], r['launches']//d['steps'], r['avg_launch_ms'], d['kernels_ms_per_step']))"
  for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
    tag=$(echo $grp | cut -d' ' -f1)
    rm -rf $OUT/p_$mb_$tag
    BOGP_CHUNK_MB=$mb timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/p_${mb}_$tag -o p -- python $ROOT/tools/pmc_sweep.py > $OUT/p_${mb}_$tag.log 2>&1
    python $ROOT/tools/pmc_summary.py $OUT/p_${mb}_$tag | grep -A6 "k_contract16" | head -8 | sed 's/^/  /'
    # kernel wall time of the same pass (kernel trace): sum over the k_contract16 dispatches of the sweep
    python - <<PY
import csv,glob
tot=0.0;n=0
for f in glob.glob("$OUT/p_${mb}_$tag/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_contract16" in r["Kernel_Name"]:
            tot+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-6; n+=1
print("  [%s pass] k_contract16: %d dispatches, %.3f ms in total" % ("$tag", n, tot))
PY
  done
done
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
