#!/bin/bash
# r05 A/B: the producer's radial profile with the library's sqrt / exp against the domain-restricted pos_sqrt / neg_exp (csrc/bogp_device.h),
# in kernel A (k_corr_chunk, BOGP_CORR_MFMA=0) and kernel A' (k_corr_mfma).  Variant libraries are built in the build container:
#   hipcc -DBOGP_LIB_SQRT / -DBOGP_LIB_EXP ... -c kernels_posterior.hip ; link -> bayesian-optimization_amd/libbogp.so.lib{sqrt,exp,both}
# Run on the GPU box (scratch copy): bash tools/ab/r05_profile_diet.sh > gpurun_out/r05_profile_diet_ab.txt
P=bayesian-optimization_amd
cp $P/libbogp.so /tmp/libbogp.orig
for v in orig libsqrt libexp libboth; do
  if [ $v = orig ]; then cp /tmp/libbogp.orig $P/libbogp.so; else cp $P/libbogp.so.$v $P/libbogp.so; fi
  for m in 0 1; do
    for w in ${WORKLOADS:-C3}; do
      BOGP_CORR_MFMA=$m python bench.py --workload $w --steps 10 --warmup 2 --no-cpu --no-seeds 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %s  producer %s  corr_ms %.3f  contract_ms %.3f  ms_per_step %.3f' % ('$v', '$w', 'k_corr_mfma ' if $m else 'k_corr_chunk', j['kernels_ms_per_step']['corr_ms'], j['kernels_ms_per_step']['contract_ms'], j['ms_per_step']))"
    done
  done
done
cp /tmp/libbogp.orig $P/libbogp.so
