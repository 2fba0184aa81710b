# r06: the phase trace of k_contract16d (gpurun -- 'bash tools/ab/r06_contract_d_trace.sh'): variants/libbogp_dtrace.so replaces the scratch tree's library
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_contract_d_trace
mkdir -p $OUT
cd $ROOT
python tools/pmc_sweep.py C3 2>&1 | grep -v amdgpu > $OUT/product_timing.txt
cp bayesian-optimization_amd/libbogp.so /tmp/libbogp_product.so
cp variants/libbogp_dtrace.so bayesian-optimization_amd/libbogp.so
timeout 900 python tools/contract_d_trace.py C3 > $OUT/trace.txt 2> $OUT/trace.err
echo "rc=$?" >> $OUT/trace.err
cp /tmp/libbogp_product.so bayesian-optimization_amd/libbogp.so
tail -3 $OUT/trace.err; cat $OUT/product_timing.txt; cat $OUT/trace.txt
