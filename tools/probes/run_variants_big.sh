# as run_variants.sh, for the large-N fit path: llf / llf + gradient / commit per library variant (variants/libbogp_*.so, tools/build_variant.sh), with the
# sha1 of (llf, gradient, committed factor) so that bit-identity of two variants shows in the output.  usage: bash tools/probes/run_variants_big.sh [sizes]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cp $R/bayesian-optimization_amd/libbogp.so /tmp/libbogp_product.so
for f in $R/variants/libbogp_*.so; do
  cp $f $R/bayesian-optimization_amd/libbogp.so
  echo "== $(basename $f .so)"
  python $R/tools/time_fit_big.py --big-only --sha ${@:-6144 8192} 2>&1 | grep 128-tile
done
cp /tmp/libbogp_product.so $R/bayesian-optimization_amd/libbogp.so
