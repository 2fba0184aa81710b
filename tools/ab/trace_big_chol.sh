# kernel timeline of 3 likelihood evaluations at N = 8192 under the env given on the command line
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/trace_big -o t -- python $ROOT/tools/prof_nll_big.py > $ROOT/gpurun_out/trace_big.log 2>&1
