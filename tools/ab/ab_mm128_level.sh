for l in 1 3 4 5; do echo "== BOGP_MM128_MIN_LEVEL=$l"; BOGP_MM128_MIN_LEVEL=$l python tools/time_fit_big.py 6144 8192 2>&1 | grep "128-tile"; done
python -m pytest tests/test_gpu_driver.py -x -q -m gpu -k "large" 2>&1 | tail -2
