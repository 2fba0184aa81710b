# two-level Cholesky (BOGP_BIG_CHOL=1) x panel width, against the one-level chain, N = 8192 / 6144
echo "== one-level"; python tools/time_fit_big.py 6144 8192 2>&1 | grep "128-tile"
for pw in 2 4 8; do echo "== BOGP_BIG_CHOL=1 BOGP_CHOL_PANEL=$pw"; BOGP_BIG_CHOL=1 BOGP_CHOL_PANEL=$pw python tools/time_fit_big.py 6144 8192 2>&1 | grep "128-tile"; done
