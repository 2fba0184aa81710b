# A/B of the stage depth of the triangular contraction at C3 and C5-size (r03; the k_contract16w variant it switched on was removed after the measurement: profiles/r03_contract_stage_ab.txt, git babf07a):
# 2 (64-row stages, pitch 80, k_contract16w), 3 (64-row stages, pitch 72).
for st in 1 2 3 1 2 3; do
echo "== stage $st"
BOGP_CONTRACT_STAGE=$st python bench.py --workload C3 --no-cpu 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read());print('C3', d['ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_ms'],d['kernels_ms_per_step'],d['argmax'])"
done
for st in 1 2 3; do
echo "== stage $st"
BOGP_CONTRACT_STAGE=$st python bench.py --workload C5 --no-cpu 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read());print('C5', d['ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_ms'],d['kernels_ms_per_step'],d['argmax'])"
done
