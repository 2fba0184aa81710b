# Round-end profile collection on the GPU box (gpurun -- 'bash tools/run_profiles.sh rNN'): the rocprofv3 kernel-trace summary of
# the default bench.py command and of C2, then the PMC passes of ONE C3 sweep (tools/pmc_sweep.py), each counter group its
# own run, kernel-trace only (never with sys/hip/hsa traces).  Every rocprofv3 call sits under its own `timeout`: a counter
# group the hardware cannot collect in one pass makes rocprofv3 abort and then HANG (FETCH_SIZE + TCC_HIT_sum + TCC_MISS_sum did).
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_stats -o c3 -- python $ROOT/bench.py --no-cpu > $OUT/c3_bench_under_rocprof.json 2> $OUT/c3_stats.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_stats -o c2 -- python $ROOT/bench.py --workload C2 --no-cpu > $OUT/c2_bench_under_rocprof.json 2> $OUT/c2_stats.err
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/pmc_a -o a -- python $ROOT/tools/pmc_sweep.py > $OUT/pmc_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_b -o b -- python $ROOT/tools/pmc_sweep.py > $OUT/pmc_b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_c -o c -- python $ROOT/tools/pmc_sweep.py > $OUT/pmc_c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_d -o d -- python $ROOT/tools/pmc_sweep.py > $OUT/pmc_d.log 2>&1
cd $ROOT
for p in a b c d; do echo "== pass $p"; python tools/pmc_summary.py $OUT/pmc_$p; done > $OUT/pmc_summary.txt
find $OUT -name "*kernel_stats.csv" | head
# keep only the summaries (the traces are large)
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
