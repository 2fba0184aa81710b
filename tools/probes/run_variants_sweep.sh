# as run_variants.sh, for the sweep: per-kernel milliseconds of a C3 step (bench.py --no-cpu) per library variant (results of experiment variants are wrong on purpose)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for f in $R/variants/libbogp_*.so; do
  cp $f $R/bayesian-optimization_amd/libbogp.so
  echo -n "== $(basename $f .so): "
  python bench.py --no-cpu --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels_ms_per_step'])" 2>&1 | tail -1
done
