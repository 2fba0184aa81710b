# k_mm128 tile order: XCD-aware super tiles (BOGP_MM128_ORDER=0, r02) vs live tiles longest-K-first (1, default since r03)
for o in 0 1; do echo "== BOGP_MM128_ORDER=$o"; BOGP_MM128_ORDER=$o python tools/time_fit_big.py 6144 7040 8192 2>&1 | grep "128-tile"; done
