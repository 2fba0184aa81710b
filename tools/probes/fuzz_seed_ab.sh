# one fuzz seed on two builds of the library (variants/libbogp_*.so): is a fuzz difference older than a change?
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for f in $R/variants/libbogp_*.so; do
  cp $f $R/bayesian-optimization_amd/libbogp.so
  echo "== $(basename $f .so)"
  python tools/fuzz_parity.py ${MODE:---wide} 1 ${SEED:-3060382} 2>&1 | grep -v amdgpu | tail -4
done
