# r06: library variants of k_sweep_small over the small training-set sizes (gpurun -- 'bash tools/ab/r06_small_sizes_ab.sh product <tag> ...'; 1e6 candidates, tools/sweep_scaling.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
cp bayesian-optimization_amd/libbogp.so /tmp/libbogp_product.so
for TAG in "$@"; do
  if [ "$TAG" = "product" ]; then cp /tmp/libbogp_product.so bayesian-optimization_amd/libbogp.so; else cp variants/libbogp_$TAG.so bayesian-optimization_amd/libbogp.so; fi
  echo "== $TAG"; python tools/sweep_scaling.py ${SIZES:-} 2>&1 | grep "1000000" | head -${ROWS:-4}
done
cp /tmp/libbogp_product.so bayesian-optimization_amd/libbogp.so
