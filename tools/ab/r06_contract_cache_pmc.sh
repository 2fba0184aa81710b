# r06: cache counters of the contraction kernels at C3 (gpurun -- 'bash tools/ab/r06_contract_cache_pmc.sh'): L1 (TCP) and L2 (TCC) traffic of
# k_contract16d and of k_contract16<4> (BOGP_CONTRACT_DIRECT=0), each counter group its own rocprofv3 run with kernel-trace only.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_cache_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3-avail list 2>/dev/null | grep -o "\(TCP\|TCC\|TA\|TD\)_[A-Z0-9_a-z]*" | sort -u > $OUT/avail_counters.txt
W=${WL:-C3}
for DIRECT in 1 0; do
  export BOGP_CONTRACT_DIRECT=$DIRECT
  i=0
  for GRP in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TA_BUSY_avr TA_BUFFER_WAVEFRONTS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $GRP -d $OUT/d${DIRECT}_$i -o p -- python $ROOT/tools/pmc_sweep.py $W > $OUT/d${DIRECT}_$i.log 2>&1
    echo "== direct=$DIRECT pass $i: $GRP (rc=$?)" >> $OUT/summary.txt
    python $ROOT/tools/pmc_summary.py $OUT/d${DIRECT}_$i 2>&1 | awk '/^[^ ]/{f=($0 ~ /k_contract16/)} f' >> $OUT/summary.txt; tail -2 $OUT/d${DIRECT}_$i.log | grep -i "error\|invalid\|not" >> $OUT/summary.txt
  done
done
cd $ROOT
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/summary.txt
