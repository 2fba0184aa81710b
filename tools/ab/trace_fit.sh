# kernel timeline of likelihood + gradient evaluations (tools/prof_nll.py: N = 2048, d = 20) under the env given on the command line
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/trace_fit -o t -- python $ROOT/tools/prof_nll.py > $ROOT/gpurun_out/trace_fit.log 2>&1
