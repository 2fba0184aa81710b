# per-kernel statistics of llf + gradient evaluations at N = 8192 (the C5 fit path): rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/bigfit
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o b -- python $R/tools/time_fit_big.py 8192 > $O/run.log 2>&1
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python $R/tools/trace_elim_steps.py $O/st k_chol_update > $O/chol_update_steps.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cut -d, -f1-4 $O/kernel_stats.csv | head -25 | cut -c1-160
tail -3 $O/run.log
