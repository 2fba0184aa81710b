# the lock-step MLE figure of the C3 bench line (10 restarts, 400-evaluation budget, N = 2048) under the launch-plan switches of the batched elimination
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { echo -n "$* : "; env "$@" python bench.py --no-cpu --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['fit']; print('mle %.1f ms, slot of ten %.3f ms' % (f['mle_10_restarts_in_lock_step_ms'], f['llf_grad_ms_per_evaluation_in_a_batch_of_10']))"; }
run X=0
run BOGP_ELIM_SUBSTEP=2000
run BOGP_ELIM_SUBSTEP=100000
run BOGP_ELIM_SUBSTEP=0
run BOGP_ELIM_SUPER=1000
run BOGP_ELIM_SUPER=700
run BOGP_ELIM_SUPER=0
run BOGP_ELIM_GROUP=2
run BOGP_ELIM_XCD=0
