# PMC passes of a full batch of likelihood evaluations (tools/time_nll_batch.py 2048 at P = 16) for profiles/r04_elim_batch_pmc.txt:
# each counter group its own rocprofv3 run, kernel-trace only.  usage (GPU box): bash tools/pmc_elim_batch.sh r04
R=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${R}_elim_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BOGP_TIME_P=16
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/pmc_a -o a -- python $ROOT/tools/time_nll_batch.py 2048 > $OUT/pmc_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_b -o b -- python $ROOT/tools/time_nll_batch.py 2048 > $OUT/pmc_b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_c -o c -- python $ROOT/tools/time_nll_batch.py 2048 > $OUT/pmc_c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_d -o d -- python $ROOT/tools/time_nll_batch.py 2048 > $OUT/pmc_d.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc_e -o e -- python $ROOT/tools/time_nll_batch.py 2048 > $OUT/pmc_e.log 2>&1
cd $ROOT
for p in a b c d e; do echo "== pass $p"; python tools/pmc_summary.py $OUT/pmc_$p | grep -A8 "k_elim_updateS_b\|k_elim_update_b\|k_elim_panel_b\|k_grad_contract_b"; done > $OUT/pmc_summary.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/pmc_summary.txt | head -120
