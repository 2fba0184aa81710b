# r06: the kernel's own phase trace (gpurun -- 'bash tools/ab/r06_contract_trace.sh [tag]'): variants/libbogp_<tag>.so (default: trace) replaces the
# scratch tree's library, tools/contract_trace.py writes the attribution.  The product library's timing of the same sweep goes first.
TAG=${1:-trace}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_contract_trace
mkdir -p $OUT
cd $ROOT
python tools/pmc_sweep.py C3 > $OUT/product_timing.txt 2>&1
cp bayesian-optimization_amd/libbogp.so /tmp/libbogp_product.so
cp variants/libbogp_$TAG.so bayesian-optimization_amd/libbogp.so
timeout 900 python tools/contract_trace.py C3 > $OUT/trace_$TAG.txt 2> $OUT/trace_$TAG.err
echo "rc=$?" >> $OUT/trace_$TAG.err
cp /tmp/libbogp_product.so bayesian-optimization_amd/libbogp.so
tail -5 $OUT/trace_$TAG.err
