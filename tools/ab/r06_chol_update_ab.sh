# r06: library variants against the large-N fit (tools/time_fit_big.py): gpurun -- 'bash tools/ab/r06_chol_update_ab.sh product cu3 ...'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
cp bayesian-optimization_amd/libbogp.so /tmp/libbogp_product.so
for TAG in "$@"; do
  if [ "$TAG" = "product" ]; then cp /tmp/libbogp_product.so bayesian-optimization_amd/libbogp.so; else cp variants/libbogp_$TAG.so bayesian-optimization_amd/libbogp.so; fi
  echo "== $TAG"
  python tools/time_fit_big.py ${SIZES:-4096 8192} 2>&1 | grep -v amdgpu.ids | grep "128-tile"
done
cp /tmp/libbogp_product.so bayesian-optimization_amd/libbogp.so
