# PMC passes (each counter group its own run, kernel-trace only) over three llf + gradient evaluations at N = 8192 (tools/prof_nll_big.py): matrix-pipe busy
# cycles and HBM traffic of the fit kernels -- k_mm128 (inverse, U U^T, the wide panels' products), k_chol_update, k_chol_step
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmc_big_fit; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/a -o a -- python $R/tools/prof_nll_big.py > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/b -o b -- python $R/tools/prof_nll_big.py > $O/b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/c -o c -- python $R/tools/prof_nll_big.py > $O/c.log 2>&1
cd $R
for p in a b c; do echo "== pass $p"; python tools/pmc_summary.py $O/$p | grep -A5 "k_mm128\|k_chol_update\|k_chol_step\|k_chol_panel"; done
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
