python -m pytest tests/test_gpu_parity.py tests/test_gpu_driver.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not full_size_argmax" > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
echo "== one synchronisation"; python tools/time_small_fit.py 2>&1 | grep "N="
echo "== BOGP_NLL_TWO_SYNCS=1"; BOGP_NLL_TWO_SYNCS=1 python tools/time_small_fit.py 2>&1 | grep "N="
python tools/time_fit_big.py 2048 2>&1 | grep 128-tile; BOGP_NLL_TWO_SYNCS=1 python tools/time_fit_big.py 2048 2>&1 | grep 128-tile
