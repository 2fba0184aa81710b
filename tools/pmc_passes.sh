# The PMC passes of ONE sweep per workload (tools/pmc_sweep.py) for profiles/<workload>_pmc.json: each counter group its own rocprofv3 run, kernel-trace only
# (never --pmc together with the hip / hsa / memory-copy trace domains).
# usage (GPU box): bash tools/pmc_passes.sh rNN [C2 C3 C4 C5] ; then in the build container: python tools/make_pmc_json.py rNN <commit> [C2 C3 C4 C5]
R=${1:-r05}; shift
WL=${@:-C3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in $WL; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/pmc_${W}_a -o a -- python $ROOT/tools/pmc_sweep.py $W > $OUT/pmc_${W}_a.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_${W}_b -o b -- python $ROOT/tools/pmc_sweep.py $W > $OUT/pmc_${W}_b.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_${W}_c -o c -- python $ROOT/tools/pmc_sweep.py $W > $OUT/pmc_${W}_c.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_${W}_d -o d -- python $ROOT/tools/pmc_sweep.py $W > $OUT/pmc_${W}_d.log 2>&1
  for p in a b c d; do echo "== pass $p"; python $ROOT/tools/pmc_summary.py $OUT/pmc_${W}_$p; done > $OUT/pmc_summary_$W.txt
done
cd $ROOT
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
head -50 $OUT/pmc_summary_*.txt
