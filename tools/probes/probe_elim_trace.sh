cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/p1
mkdir -p $O
python $R/tools/time_elim_sizes.py > $O/elim_sizes.txt 2>&1
python $R/tools/probes/time_batch_p1.py > $O/batch_p1.txt 2>&1
BOGP_ELIM_SPLIT_BLOCKS=1 python $R/tools/probes/time_batch_p1.py >> $O/batch_p1.txt 2>&1
BOGP_ELIM_SPLIT_BLOCKS=1 BOGP_ELIM_GROUP=2 python $R/tools/probes/time_batch_p1.py >> $O/batch_p1.txt 2>&1
BOGP_ELIM_SPLIT_BLOCKS=1 BOGP_ELIM_SUBSTEP=0 python $R/tools/probes/time_batch_p1.py >> $O/batch_p1.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python $R/tools/prof_nll.py > $O/tr.log 2>&1
python $R/tools/trace_elim_steps.py $O/tr k_elim_step > $O/steps.txt 2>&1
find $O -name "*.db" -delete
cat $O/elim_sizes.txt $O/batch_p1.txt $O/steps.txt
