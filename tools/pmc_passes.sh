# The PMC passes of ONE C3 sweep (tools/pmc_sweep.py) for profiles/c3_pmc.json: each counter group its own rocprofv3 run, kernel-trace only.
# usage (GPU box): bash tools/pmc_passes.sh rNN ; then here: python tools/make_pmc_json.py gpurun_out/rNN/pmc_summary.txt rNN <commit>
R=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/pmc_a -o a -- python $ROOT/tools/pmc_sweep.py > $OUT/pmc_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_b -o b -- python $ROOT/tools/pmc_sweep.py > $OUT/pmc_b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_c -o c -- python $ROOT/tools/pmc_sweep.py > $OUT/pmc_c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_d -o d -- python $ROOT/tools/pmc_sweep.py > $OUT/pmc_d.log 2>&1
cd $ROOT
for p in a b c d; do echo "== pass $p"; python tools/pmc_summary.py $OUT/pmc_$p; done > $OUT/pmc_summary.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/pmc_summary.txt | head -40
