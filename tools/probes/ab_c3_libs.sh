R=${GRAFT_REPO_ROOT:-$(pwd)}
cp $R/bayesian-optimization_amd/libbogp.so /tmp/libbogp_product.so
for v in r06start head r06start head; do
  cp $R/variants/libbogp_$v.so $R/bayesian-optimization_amd/libbogp.so
  printf "%-9s " $v
  python -c "
import sys, runpy
sys.path.insert(0, '$R')
from bogp import _lib
_lib.SIGNATURES.pop('bogp_chol_wide_panels', None)
sys.argv = ['bench.py', '--no-cpu', '--no-seeds', '--steps', '10', '--warmup', '3']
runpy.run_path('$R/bench.py', run_name='__main__')
" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['kernels_ms_per_step'])"
done
cp /tmp/libbogp_product.so $R/bayesian-optimization_amd/libbogp.so
