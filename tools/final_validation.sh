# Round-end validation on the GPU box: bash tools/final_validation.sh rNN  (gpurun -- 'bash tools/final_validation.sh r04')
# the whole -m gpu suite, smoke(), the default bench line (C3) and C2 / C4 / C5, then the rocprofv3 kernel statistics of the same bench commands.
R=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/${R}_gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/${R}_gpu_tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/${R}_c3_bench_final.json 2> gpurun_out/${R}_bench_err.log; tail -c 400 gpurun_out/${R}_c3_bench_final.json
for w in C2 C4 C5; do python bench.py --workload $w --no-cpu 2>/dev/null | tail -1; done > gpurun_out/${R}_c2_c4_c5_bench.jsonl
python bench.py --workload C2 --no-cpu 2>/dev/null | tail -1 > gpurun_out/${R}_c2_bench.json
OUT=$ROOT/gpurun_out/${R}_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_stats -o c3 -- python $ROOT/bench.py --no-cpu > $OUT/c3_bench_under_rocprof.json 2> $OUT/c3_stats.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_stats -o c2 -- python $ROOT/bench.py --workload C2 --no-cpu > $OUT/c2_bench_under_rocprof.json 2> $OUT/c2_stats.err
cd $ROOT
find $OUT -name "*kernel_stats.csv" | head
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT
