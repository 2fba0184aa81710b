# kernel timelines of 3 likelihood evaluations at N = 8192: one-level, two-level (+ CU-masked look-ahead stream)
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/trace_big_one -o t -- python $ROOT/tools/prof_nll_big.py > $ROOT/gpurun_out/trace_big_one.log 2>&1
BOGP_BIG_CHOL=1 BOGP_CHOL_PANEL=8 BOGP_CHOL_RESERVE_CU=32 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/trace_big_two -o t -- python $ROOT/tools/prof_nll_big.py > $ROOT/gpurun_out/trace_big_two.log 2>&1
