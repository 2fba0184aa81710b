# fused block-column steps (k_chol_step) x threshold: llf / llf+grad / commit at N = 512 .. 8192 (tools/time_fit_big.py)
for f in 0 16 32 48 64; do echo "== BOGP_CHOL_FUSE_MAX=$f"; BOGP_CHOL_FUSE_MAX=$f python tools/time_fit_big.py 512 1024 2048 4096 8192 2>&1 | grep "128-tile"; done
