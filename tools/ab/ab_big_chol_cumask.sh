# two-level Cholesky (BOGP_BIG_CHOL=1) with its look-ahead update on a CU-masked stream (BOGP_CHOL_RESERVE_CU=n CUs left to
# the panel chain), against the one-level chain, N = 8192 (r03)
echo "== one-level"; python tools/time_fit_big.py 8192 2>&1 | grep "128-tile"
for pw in 4 8; do for n in 0 16 32 64 96; do echo "== BOGP_BIG_CHOL=1 BOGP_CHOL_PANEL=$pw BOGP_CHOL_RESERVE_CU=$n"; BOGP_BIG_CHOL=1 BOGP_CHOL_PANEL=$pw BOGP_CHOL_RESERVE_CU=$n python tools/time_fit_big.py 8192 2>&1 | grep "128-tile"; done; done
