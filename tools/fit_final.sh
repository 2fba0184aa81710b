./tools/probes/ubench_diag > gpurun_out/r02_ubench_diag.txt 2>&1
python tools/time_fit_big.py 512 1024 2048 4096 6144 8192 2>&1 | grep "128-tile" > gpurun_out/r02_fit_timing_final.txt
cat gpurun_out/r02_ubench_diag.txt gpurun_out/r02_fit_timing_final.txt
