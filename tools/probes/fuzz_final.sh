# differential fuzz of the round's last build: three modes + the row-pair steps forced at every elimination size + block steps only
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/fuzz
(echo "standard:"; python tools/fuzz_parity.py 240 3010000 2>&1 | tail -4
 echo "--wide:"; python tools/fuzz_parity.py --wide 300 3020000 2>&1 | tail -4
 echo "--trend:"; python tools/fuzz_parity.py --trend 150 3030000 2>&1 | tail -4
 echo "BOGP_ELIM_STEP_PAIR_MIN=1 BOGP_ELIM_STEP_PAIR_MAX=1000000 --wide (row-pair steps at every size of the elimination path):"
 BOGP_ELIM_STEP_PAIR_MIN=1 BOGP_ELIM_STEP_PAIR_MAX=1000000 python tools/fuzz_parity.py --wide 240 3040000 2>&1 | tail -4
 echo "BOGP_ELIM_STEP_PAIR_MIN=0 --wide (block steps only):"
 BOGP_ELIM_STEP_PAIR_MIN=0 python tools/fuzz_parity.py --wide 150 3050000 2>&1 | tail -4
 echo "soak:"; python tools/soak.py 2>&1 | tail -2) | grep -v amdgpu.ids | tee gpurun_out/fuzz/final.txt
