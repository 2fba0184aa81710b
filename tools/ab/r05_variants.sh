#!/bin/bash
# r05 A/B harness: for every variant library bayesian-optimization_amd/libbogp.so.<v> (built in the build container from -D switches, see
# profiles/r05_profile_diet_ab.txt / r05_corr_mfma_ab.txt) run bench.py on WORKLOADS and tools/sweep_scaling.py on SIZES; prints the producer / contraction / step times.
# usage (GPU box): VARIANTS="orig a b" WORKLOADS="C3 C5" SIZES="768 1024" bash tools/ab/r05_variants.sh
P=bayesian-optimization_amd
cp $P/libbogp.so /tmp/libbogp.orig
for rep in $(seq 1 ${REPS:-1}); do
for v in ${VARIANTS}; do
  if [ $v = orig ]; then cp /tmp/libbogp.orig $P/libbogp.so; else cp $P/libbogp.so.$v $P/libbogp.so; fi
  for w in ${WORKLOADS:-C3}; do
    python bench.py --workload $w --steps 10 --warmup 2 --no-cpu --no-seeds 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-18s %s  corr_ms %.3f  contract_ms %.3f  ms_per_step %.3f' % ('$v', '$w', j['kernels_ms_per_step']['corr_ms'], j['kernels_ms_per_step']['contract_ms'], j['ms_per_step']))"
  done
  if [ -n "${SIZES:-}" ]; then python tools/sweep_scaling.py $SIZES 2>/dev/null | grep " 1000000 " | sed "s/^/$v  /"; fi
done
done
cp /tmp/libbogp.orig $P/libbogp.so
