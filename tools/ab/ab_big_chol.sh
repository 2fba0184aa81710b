# wide first panels of the large-N Cholesky (BOGP_BIG_CHOL = list of block columns per wide panel; "0" = the one-level chain alone)
# usage: bash tools/ab/ab_big_chol.sh  ["N:sched;sched;..." ...]   e.g.  "8192:0;24,24;32"  "6144:0;16"
for spec in "${@:-6144:0;16 7040:0;24,24 8192:0;32;24,24;16,16,16}"; do
  N=${spec%%:*}; IFS=';' read -ra SCHEDS <<< "${spec#*:}"
  for sched in "${SCHEDS[@]}"; do
    printf "BOGP_BIG_CHOL=%-12s " "$sched"; BOGP_BIG_CHOL=$sched python tools/time_fit_big.py --big-only $N 2>&1 | grep "128-tile"
  done
done
