# Cholesky trailing updates on triangular grids (live tiles only, BOGP_CHOL_TRI_GRID=1) vs m x m grids with an empty upper half (0)
for t in 0 1 0 1; do echo "== BOGP_CHOL_TRI_GRID=$t"; BOGP_CHOL_TRI_GRID=$t python tools/time_fit.py 2>&1 | tail -8; BOGP_CHOL_TRI_GRID=$t python tools/time_fit_big.py 4096 8192 2>&1 | grep -E "128-tile|64-block"; done
