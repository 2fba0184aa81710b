ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
cp bayesian-optimization_amd/libbogp.so /tmp/libbogp_product.so
for TAG in product smallold product smallold; do
  if [ "$TAG" = "product" ]; then cp /tmp/libbogp_product.so bayesian-optimization_amd/libbogp.so; else cp variants/libbogp_$TAG.so bayesian-optimization_amd/libbogp.so; fi
  echo "== $TAG"; python tools/sweep_scaling.py 2>&1 | grep "1000000" | head -4
done
cp /tmp/libbogp_product.so bayesian-optimization_amd/libbogp.so
