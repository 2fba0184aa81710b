# r06: the phase trace of k_contract16d (gpurun -- 'bash tools/ab/r06_contract_d_trace.sh [tag ...]'): variants/libbogp_<tag>.so (default: dtrace) replaces the scratch tree's library.
# dtrace1 = the FIRST version of the kernel (one load in front of every group of four MFMAs, 64-bit VALU address adds) under the same stamps.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_contract_d_trace
mkdir -p $OUT
cd $ROOT
python tools/pmc_sweep.py C3 2>&1 | grep -v amdgpu > $OUT/product_timing.txt
cp bayesian-optimization_amd/libbogp.so /tmp/libbogp_product.so
for TAG in ${@:-dtrace}; do
  cp variants/libbogp_$TAG.so bayesian-optimization_amd/libbogp.so
  timeout 900 python tools/contract_d_trace.py C3 > $OUT/trace_$TAG.txt 2> $OUT/trace_$TAG.err
  echo "rc=$?" >> $OUT/trace_$TAG.err
done
cp /tmp/libbogp_product.so bayesian-optimization_amd/libbogp.so
tail -2 $OUT/*.err; cat $OUT/product_timing.txt
