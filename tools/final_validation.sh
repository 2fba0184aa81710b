python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r02_c3_bench_final.json 2> gpurun_out/bench_err.log; tail -c 300 gpurun_out/r02_c3_bench_final.json
for w in C2 C4 C5; do python bench.py --workload $w --no-cpu 2>/dev/null | tail -1; done > gpurun_out/r02_c2_c4_c5_bench_final.jsonl
