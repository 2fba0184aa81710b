# grouped one-level Cholesky for large N: group size x SYRK threshold (llf / llf+grad / commit, tools/time_fit_big.py)
for cfg in "0 2" "24 2" "40 2" "56 2" "24 4" "40 4"; do set -- $cfg
echo "== BOGP_CHOL_SYRK_MIN=$1 BOGP_CHOL_GROUP=$2"; BOGP_CHOL_SYRK_MIN=$1 BOGP_CHOL_GROUP=$2 python tools/time_fit_big.py 6144 8192 2>&1 | grep "128-tile"; done
echo "== GPU tests of the large-fit path"; python -m pytest tests/test_gpu_driver.py -x -q -m gpu -k "large or big" 2>&1 | tail -3
