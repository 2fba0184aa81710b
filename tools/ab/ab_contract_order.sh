# A/B of the workgroup order of k_contract16 x the chunk size, C3 (profiles/r02_contract_order_chunk_ab.txt).
# BOGP_CONTRACT_ORDER was an experimental switch of r02 (xcd: the nJ column groups of one candidate tile back to back on one
# XCD); it lost by 39 % and was removed again -- the script is kept as the record of what was run.
for ord in group xcd; do for mb in 1024 4096 16384; do
echo "== order $ord chunk $mb"
BOGP_CONTRACT_ORDER=$ord BOGP_CHUNK_MB=$mb python bench.py --workload C3 --no-cpu 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read());print(d['ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_ms'],d['kernels_ms_per_step'],d['argmax'])"
done; done
