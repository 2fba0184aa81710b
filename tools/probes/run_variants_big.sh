# as run_variants.sh, for the Cholesky path: llf / llf + gradient at N = 4096 and 8192 per library variant
R=${GRAFT_REPO_ROOT:-$(pwd)}
for f in $R/variants/libbogp_*.so; do
  cp $f $R/bayesian-optimization_amd/libbogp.so
  echo "== $(basename $f .so)"
  python $R/tools/time_fit_big.py 4096 8192 2>&1 | grep 128-tile
done
