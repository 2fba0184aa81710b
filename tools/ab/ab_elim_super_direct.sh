R=${GRAFT_REPO_ROOT:-$(pwd)}
cp $R/bayesian-optimization_amd/libbogp.so /tmp/libbogp_product.so
echo "== bit-identity of the batch knobs on the product build (direct super-tile update)"
timeout 600 python -m pytest $R/tests/test_gpu_nll_batch_variants.py $R/tests/test_gpu_nll_batch.py -m gpu -q -x 2>&1 | tail -3
for v in superdirect superlds superdirect superlds; do
  cp $R/variants/libbogp_$v.so $R/bayesian-optimization_amd/libbogp.so
  echo "== libbogp_$v"
  BOGP_TIME_P=4,8,10,16 python $R/tools/time_nll_batch.py 1536 2048 3072 2>&1 | grep -v amdgpu.ids
done
cp /tmp/libbogp_product.so $R/bayesian-optimization_amd/libbogp.so
