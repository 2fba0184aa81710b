"""bench.py -- candidates/sec of the GP posterior + acquisition + argmax sweep on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

With N > 1 and no torch.distributed.run environment (RANK / WORLD_SIZE unset) the script launches its own N ranks
(`launch_command`: one process per GPU on 127.0.0.1, a free port) and rank 0's JSON line is the last line of stdout,
so `python bench.py --gpus 8` works the way `--gpus 1` does.

A "step" is one pass of the hot path over one batch of synthetic candidates already resident in HBM: posterior
(mu, MSE) of M candidates per GPU + q = 2 criteria (MGFI t=2, EI) + argmax, then the one cross-rank exchange of the
per-shard winners -- `bogp_exchange_argmax`: the winners are packed on the device and gathered with ONE ncclAllGather
on the library's own RCCL communicator; it runs at N = 1 too (a one-rank communicator), so the timed step always
contains the collective.  `--scaling strong` keeps the TOTAL candidate count fixed (ragged contiguous shards).  Workload = BASELINE.json configs[2] ("C3", the configuration the metric is quoted on):
N = 2048 training points, d = 20, Matern-5/2, M = 1e6 candidates per GPU (weak scaling: configs[3] is 8 x 1e6).
Hyper-parameters are pinned (theta = 0.01, sigma2 = 0.9, nugget 1e-6, simple kriging), not fitted, as SURVEY.md
section 8d prescribes; X ~ U[-5,5], y = sum x^2 standardised; seeds fixed.

Prints ONE JSON line (rank 0) with the driver's fields + "roofline" (dominant kernel k_contract: algorithmic FP64
flops / HIP-event duration on the library's stream) + "cpu_baseline" (the NumPy oracle timed on this box's host
cores on a bounded sample of the same workload; also used as a parity check of the timed run).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK_FP64_TFLOPS = 78.6  # MI355X FP64 vector = matrix peak (AMD spec; 256 CU x 128 flop/clk x 2.4 GHz)
WORKLOADS = {
    "C2": dict(N=512, d=10, M=100_000, kernel=0, theta=0.02, acq=[(0, 0.0)], name="C2: N=512 d=10 M=1e5 SE EI"),
    "C3": dict(N=2048, d=20, M=1_000_000, kernel=3, theta=0.01, acq=[(3, 2.0), (0, 0.0)],
               name="C3: N=2048 d=20 M=1e6/GPU Matern-5/2 MGFI(t=2)+EI"),  # fmt: skip
    # configs[3]: ParallelBO batch q = 8, t_i = exp(log 2 + 0.5 z_i) (bayes_opt.py:84-86), z from a seeded rng; 1e6 per GPU
    "C4": dict(N=2048, d=20, M=1_000_000, kernel=3, theta=0.01,
               acq=[(3, float(t)) for t in np.exp(np.log(2.0) + 0.5 * np.random.default_rng(4).standard_normal(8))],
               name="C4: N=2048 d=20 M=1e6/GPU Matern-5/2 q=8 MGFI(t_i ~ logN(log 2, 0.5))"),
    "C5": dict(N=8192, d=50, M=500_000, kernel=0, theta=0.004, acq=[(2, 0.5)], name="C5: N=8192 d=50 M=5e5/GPU SE UCB"),
}


def dominant_kernel(workload):
    """(name pattern, source file, first line marker, last line marker) of the kernel `roofline` is about: the one-launch sweep at N <= 512,
    the triangular contraction otherwise."""
    if WORKLOADS[workload]["N"] <= 512:
        return "k_sweep_small", "kernels_small.hip", None, None
    return "k_contract16", "kernels_posterior.hip", "// Kernel B: triangular contraction", "// host-side launchers"


def candidates_per_launch(workload):
    """Candidates one launch of the dominant kernel processes: the whole sweep for k_sweep_small, one 1-GiB chunk of r otherwise."""
    w = WORKLOADS[workload]
    if w["N"] <= 512:
        return w["M"]
    return ((1 << 30) // (((w["N"] + 31) // 32 * 32) * 8)) // 64 * 64


def kernel_source_hash(workload="C3"):
    """sha256 of the CODE of the dominant kernel (its region of the translation unit; comments and blank lines stripped: neither a reworded
    comment nor an edit to ANOTHER kernel of the file invalidates a measurement): a committed PMC figure is only quoted for the source it
    was measured on."""
    import hashlib
    import re

    _, fname, first, last = dominant_kernel(workload)
    with open(os.path.join(ROOT, "bayesian-optimization_amd", "csrc", fname)) as f:
        src = f.read()
    if first is not None:
        src = src[src.index(first) : src.index(last)]
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    code = [re.sub(r"\s+", " ", re.sub(r"//.*$", "", ln)).strip() for ln in src.split("\n")]
    return hashlib.sha256("\n".join(c for c in code if c).encode()).hexdigest()[:16]


def measured_traffic(workload, n_per_launch):
    """HBM bytes per launch of the dominant kernel.  rocprofv3 counters cannot be collected from inside this process, so
    this is a COMMITTED figure: the last PMC passes (tools/pmc_passes.sh -> profiles/<workload>_pmc.json: FETCH_SIZE x2 gfx950
    correction + WRITE_SIZE, separate --pmc runs), quoted only when chunk size AND the kernel source hash recorded with it match the
    running code; otherwise (None, reason) -- never a stale number."""
    pmc_file = os.path.join("profiles", "%s_pmc.json" % workload.lower())
    try:
        with open(os.path.join(ROOT, pmc_file)) as f:
            p = json.load(f)
    except Exception:
        return None, "no committed PMC file (%s)" % pmc_file
    if os.environ.get("BOGP_CHUNK_MB"):
        return None, "committed PMC passes are for the default chunk size"
    if int(p["candidates_per_launch"]) != int(n_per_launch):
        return None, "committed PMC passes used %s candidates per launch" % p["candidates_per_launch"]
    if p.get("kernel_source_sha256") != kernel_source_hash(workload):
        return None, "kernel source changed since the committed PMC passes (%s, commit %s)" % (pmc_file, p.get("commit", "?"))
    return float(p["traffic_bytes_per_launch"]), "committed PMC passes of commit %s (%s); not collected in this run" % (p.get("commit", "?"), pmc_file)


def launch_command(n_gpus, argv, port=None):
    """argv of the self-launch: `python -m torch.distributed.run` with one rank per GPU of this node, rendezvous on
    127.0.0.1 (the container hostname may not resolve) at a free port, and bench.py's own arguments forwarded untouched."""
    if port is None:
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_gpus)), "--master-addr", "127.0.0.1",
            "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)  # fmt: skip


def plumbing_check(args):
    """`--plumbing-check` (CPU, gloo): what a rank sees after the (self-)launch -- its rank / world / local rank, that a
    collective over the N ranks works, and the arguments as forwarded.  Rank 0 prints one JSON line.  Exercised by
    tests/test_dist_gloo.py so that the first multi-GPU driver run cannot die in the launch plumbing."""
    import torch
    import torch.distributed as dist

    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" in os.environ:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        got = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(got, torch.tensor([rank, local], dtype=torch.int64))
        ranks = [[int(v) for v in t] for t in got]
        dist.destroy_process_group()
    else:
        ranks = [[0, 0]]
    if rank == 0:
        print(json.dumps({"plumbing": True, "n_gpus": world, "gpus_arg": args.gpus, "ranks": ranks, "steps": args.steps, "warmup": args.warmup,
                          "workload": args.workload, "scaling": args.scaling, "master_addr": os.environ.get("MASTER_ADDR")}), flush=True)  # fmt: skip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-sample", type=int, default=40960, help="candidates timed through the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-seeds", action="store_true", help="skip the seed-1 / seed-2 repeats of the workload (SURVEY.md 8d)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: M candidates per GPU (default); strong: the workload's total (C3 1e6, C4 8e6, C5 4e6) split over the ranks")
    ap.add_argument("--plumbing-check", action="store_true", help=argparse.SUPPRESS)
    # CPU dry run of the WHOLE step on N gloo ranks (tests/test_dist_gloo.py): the engine is a stand-in class named here
    # (tests/support/oracle_engine.py:OracleEngine -- test infrastructure, loaded only under this flag), sizes are cut down to "N,d,M";
    # --dry-fail-comm-rank R makes rank R fail to build its communicator, which must send EVERY rank down the fallback exchange
    ap.add_argument("--dry-run-engine", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--dry-size", default="96,4,4001", help=argparse.SUPPRESS)
    ap.add_argument("--dry-fail-comm-rank", type=int, default=-1, help=argparse.SUPPRESS)
    args = ap.parse_args()
    dry = args.dry_run_engine is not None

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        import subprocess

        sys.stdout.flush()
        raise SystemExit(subprocess.call(launch_command(args.gpus, sys.argv[1:])))
    if args.plumbing_check:
        return plumbing_check(args)

    import torch
    import torch.distributed as dist

    from bogp import _lib, distributed

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs a %d-rank torch.distributed.run launch (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
    dev = torch.device("cpu") if dry else torch.device("cuda", local)
    if not dry:
        torch.cuda.set_device(local)
    use_dist = world > 1 or "RANK" in os.environ  # launched by torch.distributed.run (also with 1 rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    w = dict(WORKLOADS[args.workload])
    total_strong = {"C2": 100_000, "C3": 1_000_000, "C4": 8_000_000, "C5": 4_000_000}[args.workload]
    if dry:  # the same criteria and kernel, a problem the CPU stand-in sweeps in a second
        w["N"], w["d"], w["M"] = (int(v) for v in args.dry_size.split(","))
        w["theta"] = 0.3 / w["d"]
        total_strong = w["M"] * world + 3  # ragged for every world size > 1
    N, d, M = w["N"], w["d"], w["M"]
    offset = rank * M
    if args.scaling == "strong":  # fixed total, contiguous ragged shards (optim.shard_bounds)
        from bogp.optim import shard_bounds

        a_, b_ = shard_bounds(total_strong, rank, world)
        M, offset = b_ - a_, a_
    rng = np.random.default_rng(0)  # the model is replicated: every rank builds and factorises the same one
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(X**2, axis=1)
    y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    par = np.r_[np.full(d, w["theta"]), 0.9]
    plugin = float(y.min())

    if dry:
        import importlib

        mod, cls = args.dry_run_engine.split(":")
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        eng = getattr(importlib.import_module(mod), cls)()
    else:
        eng = _lib.Engine(local)
    eng.set_train(X, y)
    t0 = time.perf_counter()
    llf = eng.commit(w["kernel"], _lib.MODE_NOISY, par, 1e-6, False, 0.0)
    commit_s = time.perf_counter() - t0  # first call in the process: includes library initialisation
    # fit is reported separately: one likelihood (+ gradient) evaluation at the pinned parameters = what the
    # MLE loop of GaussianProcess.fit pays per L-BFGS-B evaluation; the model is re-committed afterwards
    fit_ms = {}
    if not dry:
        for name, eg in (("llf_ms", False), ("llf_grad_ms", True)):
            eng.nll(w["kernel"], _lib.MODE_NOISY, par, 1e-6, False, 0.0, eval_grad=eg)
            t0 = time.perf_counter()
            for _ in range(3):
                eng.nll(w["kernel"], _lib.MODE_NOISY, par, 1e-6, False, 0.0, eval_grad=eg)
            fit_ms[name] = (time.perf_counter() - t0) / 3 * 1e3
        # the same evaluation as one slot of a batch of 10 (bogp_nll_batch: what a lock-step MLE round of 10 restarts pays per
        # restart, gpr.py:1127-1162 with GaussianProcess(restart_batch=10)); parameter vectors spread around the pinned one
        pars10 = np.tile(par, (10, 1)) * 10.0 ** np.random.default_rng(5).uniform(-0.2, 0.2, size=(10, len(par)))
        eng.nll_batch(w["kernel"], _lib.MODE_NOISY, pars10, 1e-6, False, 0.0, eval_grad=True)
        t0 = time.perf_counter()
        for _ in range(3):
            eng.nll_batch(w["kernel"], _lib.MODE_NOISY, pars10, 1e-6, False, 0.0, eval_grad=True)
        fit_ms["llf_grad_ms_per_evaluation_in_a_batch_of_10"] = (time.perf_counter() - t0) / 3 / 10 * 1e3
        # a whole MLE the lock-step way (bogp_mle_batch = GaussianProcess(restart_batch=10)): 10 restarts around the pinned parameters,
        # a shared budget of 400 likelihood + gradient evaluations, log10 search box of 2.5 decades per parameter
        x0 = np.log10(pars10)
        t0 = time.perf_counter()
        _, _, nev, _, rounds = eng.mle_batch(w["kernel"], _lib.MODE_NOISY, x0, np.log10(par) - 1.5, np.log10(par) + 1.0, 1e-6, False, 0.0,
                                             eval_budget=400)
        fit_ms["mle_10_restarts_in_lock_step_ms"] = (time.perf_counter() - t0) * 1e3
        fit_ms["mle_evaluations"], fit_ms["mle_device_rounds"] = int(np.sum(nev)), int(rounds)
        t0 = time.perf_counter()
        eng.commit(w["kernel"], _lib.MODE_NOISY, par, 1e-6, False, 0.0)
        fit_ms["commit_ms"] = (time.perf_counter() - t0) * 1e3

    # this rank's candidate shard, generated on the device and adopted without a copy
    if dry:
        # rows [offset, offset + M) of ONE global table that every rank (and the checking test) can rebuild: row i is drawn from its own
        # seed, so a shard does not depend on how the table is cut
        Xs = torch.from_numpy(dry_candidates(offset, M, d))
        eng.upload_candidates(Xs.numpy())
    else:
        g = torch.Generator(device="cuda")
        g.manual_seed(1234 + rank)
        Xs = (torch.rand((M, d), dtype=torch.float64, device="cuda", generator=g) * 10.0 - 5.0).contiguous()
        torch.cuda.synchronize()
        eng.bind_candidates(Xs.data_ptr(), M, owner=Xs)
    # the library's own RCCL communicator over the ranks (a one-rank communicator at N = 1: the collective still runs)
    # (RCCL announces itself on C stdout: send that to stderr so that stdout carries the JSON line and nothing else)
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    comm_error = None
    try:
        if rank == args.dry_fail_comm_rank:  # (fails where ncclCommInitRank would: after the id has been broadcast)

            def _fail(*_a):
                raise RuntimeError("communicator creation failed on rank %d (forced by --dry-fail-comm-rank)" % rank)

            eng.comm_init = _fail
        distributed.init_engine_comm(eng)
    except Exception as e:  # noqa: BLE001 -- reported in the JSON line, never silent
        comm_error = "%s: %s" % (type(e).__name__, e)
    finally:
        ctypes.CDLL(None).fflush(None)
        os.dup2(saved, 1)
        os.close(saved)
    if use_dist:  # every rank takes the same path: the library exchange only if EVERY rank has its communicator
        ok = torch.tensor([0.0 if comm_error else 1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) == 0.0 and comm_error is None:
            comm_error = "another rank could not create the library communicator"
    if comm_error and eng.comm_world:
        eng.comm_destroy()
    q = len(w["acq"])

    def step():
        if comm_error is None:
            eng.sweep(w["acq"], plugin, True, local_result=False)  # queued; the exchange below is the step's one host wait
            return eng.exchange_argmax(q, offset, True)  # (values, GLOBAL indices, points), identical on every rank
        # the library communicator could not be built (reported as `exchange` in the JSON line): same sweep, the q winners
        # exchanged through torch.distributed instead (one all-gather of the same records)
        bv, bi = eng.sweep(w["acq"], plugin, True)
        return distributed.exchange_argmax(bv, bi + offset, eng.read_candidates(bi))

    def fence():
        if use_dist:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    tim = dict(corr_ms=0.0, contract_ms=0.0, acquisition_ms=0.0, n_chunks=0)
    step_ms = []  # wall time of each step (a step ends in the exchange's host wait, so these need no extra synchronisation)
    for _ in range(args.steps):
        ts = time.perf_counter()
        out = step()
        step_ms.append((time.perf_counter() - ts) * 1e3)
        lt = eng.last_timing()
        for k in tim:
            tim[k] += lt[k]
    fence()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [elapsed / args.steps * 1e3]
    if use_dist:
        each = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(each, torch.tensor([elapsed], dtype=torch.float64, device=dev))
        per_rank_ms = [float(e.item()) / args.steps * 1e3 for e in each]
        elapsed = max(float(e.item()) for e in each)  # MAX over ranks

    # PCIe-inclusive ask(): H2D of the shard + one step (noted, never `value`)
    h2d_ms = gen_ms = full_ms = None
    if rank == 0 and args.scaling == "weak" and not dry:
        Xh = Xs.cpu().numpy()
        h2d = []
        for i in range(4):  # (the first pass allocates the library's candidate buffer and creates the copy stream: not timed)
            t1 = time.perf_counter()
            eng.upload_candidates(Xh, lazy=True)  # (what optim.sweep_argmax does: the copy of chunk c + 1 rides beside the kernels of chunk c)
            eng.sweep(w["acq"], plugin, True)
            if i:
                h2d.append((time.perf_counter() - t1) * 1e3)
        h2d_ms = float(np.median(h2d))
        # fully on-device ask(): candidates drawn by the library's Philox kernel (no host sampling, no H2D) + sweep
        eng.generate_candidates([-5.0] * d, [5.0] * d, M, seed=99)
        t1 = time.perf_counter()
        eng.generate_candidates([-5.0] * d, [5.0] * d, M, seed=100)
        gb, gi = eng.sweep(w["acq"], plugin, True)
        eng.read_candidates(gi)
        gen_ms = (time.perf_counter() - t1) * 1e3
        # a FULL ask() of the fused proposal (what integration.fused_batch_arg_max_acquisition does per ParallelBO.ask):
        # candidates drawn on the device -> one posterior pass for the q criteria -> top-16 per criterion (de-duplication
        # fall-backs) -> read-back of the 16 q winning points; measured on rank 0's shard, exchange not included
        t1 = time.perf_counter()
        eng.generate_candidates([-5.0] * d, [5.0] * d, M, seed=101)
        tv, ti = eng.sweep_topk(w["acq"], plugin, True, 16)
        eng.read_candidates(ti.ravel())
        full_ms = (time.perf_counter() - t1) * 1e3
        eng.bind_candidates(Xs.data_ptr(), M, owner=Xs)

    # seeds 0..2.  The timed region above is seed 0 (the driver's number); on one GPU the same workload is repeated
    # for seeds 1 and 2 -- another training set, another candidate shard, the model re-committed outside the timed steps -- and the
    # median / min / max step of each seed is reported beside it.  The committed model and candidates are seed 0's again afterwards.
    seed_stats = None
    if rank == 0 and world == 1 and args.scaling == "weak" and not args.no_seeds and not dry:
        seed_stats = {"0": {"median_ms": float(np.median(step_ms)), "min_ms": float(np.min(step_ms)), "max_ms": float(np.max(step_ms)), "steps": len(step_ms)}}
        for sd in (1, 2):
            r2 = np.random.default_rng(sd)
            X2 = r2.uniform(-5, 5, size=(N, d))
            y2 = np.sum(X2**2, axis=1)
            y2 = ((y2 - y2.mean()) / y2.std()).reshape(-1, 1)
            eng.set_train(X2, y2)
            eng.commit(w["kernel"], _lib.MODE_NOISY, par, 1e-6, False, 0.0)
            g2 = torch.Generator(device="cuda")
            g2.manual_seed(1234 + 1000 * sd)
            Xs2 = (torch.rand((M, d), dtype=torch.float64, device="cuda", generator=g2) * 10.0 - 5.0).contiguous()
            torch.cuda.synchronize()
            eng.bind_candidates(Xs2.data_ptr(), M, owner=Xs2)
            pl2 = float(y2.min())
            ms = []
            for i in range(2 + max(3, min(args.steps, 5))):
                ts = time.perf_counter()
                eng.sweep(w["acq"], pl2, True)
                if i >= 2:
                    ms.append((time.perf_counter() - ts) * 1e3)
            seed_stats[str(sd)] = {"median_ms": float(np.median(ms)), "min_ms": float(np.min(ms)), "max_ms": float(np.max(ms)), "steps": len(ms)}
        eng.set_train(X, y)
        eng.commit(w["kernel"], _lib.MODE_NOISY, par, 1e-6, False, 0.0)
        eng.bind_candidates(Xs.data_ptr(), M, owner=Xs)

    if use_dist:  # ragged shards under --scaling strong: the job's candidate count is the sum over ranks
        tm = torch.tensor([float(M)], dtype=torch.float64, device=dev)
        dist.all_reduce(tm, op=dist.ReduceOp.SUM)
        M_total = float(tm.item())
    else:
        M_total = float(M)
    if dry:
        # every rank must hold the same (values, GLOBAL indices, points): gathered and compared here, reported by rank 0
        mine = [np.asarray(out[0]).tolist(), [int(i) for i in out[1]], np.asarray(out[2]).tolist()]
        everyone = [None] * world
        if use_dist:
            dist.all_gather_object(everyone, mine)
        else:
            everyone = [mine]
        if rank == 0:
            ctypes.CDLL(None).fflush(None)
            print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "scaling": args.scaling, "M_total": M_total, "shard": [offset, M],
                              "values": mine[0], "argmax": mine[1], "points": mine[2], "ranks_identical": all(e == mine for e in everyone),
                              "exchange": "engine" if comm_error is None else "fallback (%s)" % comm_error,
                              "ms_per_step": elapsed / args.steps * 1e3}), flush=True)  # fmt: skip
        if use_dist:
            dist.destroy_process_group()
        return
    if rank == 0:
        total = M_total * args.steps
        value = total / elapsed
        traffic, traffic_source = measured_traffic(args.workload, min(candidates_per_launch(args.workload), M) if N <= 512 else candidates_per_launch(args.workload))
        flops_contract = (float(N) * N + 3.0 * N) * M * args.steps  # k_contract: forward substitution + sum of squares
        achieved = flops_contract / (tim["contract_ms"] * 1e-3) / 1e12
        if tim["corr_ms"] == 0.0 and tim["acquisition_ms"] == 0.0:  # the fused small-N sweep: ONE kernel, its whole time
            kernel_label = ("k_sweep_small (N <= 512: correlation producer + v_mfma_f64_16x16x4_f64 contraction + acquisition + "
                            "argmax in ONE kernel; `achieved` counts the contraction's flops over the whole kernel's time)")
        else:
            kernel_label = "k_contract16d (v_mfma_f64_16x16x4_f64, VGPR accumulators, operands straight from global memory: no LDS, no barrier in the main loop)"
        res = {
            "metric": "candidates/sec (GP posterior+EI) at N=2048,d=20 and ask() wall-time, 1/2/4/8 GPU",
            "value": value,
            "unit": "candidates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_per_rank": per_rank_ms,
            # spread over the K timed steps of rank 0 (`ms_per_step` above is the mean the contract asks for: K steps / wall time)
            "step_ms": {"median": float(np.median(step_ms)), "min": float(np.min(step_ms)), "max": float(np.max(step_ms))},
            "seeds": seed_stats,
            "rccl_world": int(eng.comm_world) if comm_error is None else (world if use_dist else 0),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": w["name"], "N": N, "d": d, "M_per_gpu": M, "M_total": M_total, "q": len(w["acq"]),
                       "parallelism": "candidate shards x%d, 1 ncclAllGather of q*(val,idx,x) per step (bogp_exchange_argmax)" % world},
            "roofline": (lambda r: dict(r, hbm_GBps=(r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9) if r["traffic"] else None))({
                "bound": "mfma", "kernel": kernel_label, "achieved": achieved, "peak": PEAK_FP64_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_TFLOPS,
                "traffic": traffic, "traffic_source": traffic_source,
                "avg_launch_ms": tim["contract_ms"] / max(1, tim["n_chunks"]), "launches": tim["n_chunks"],
                "flops_per_candidate": float(N) * N + 3.0 * N,
                # what the kernel executes: 16 x 16 tiles on and below the diagonal of the (padded) triangle
                "executed_flops_per_candidate": 256.0 * ((N + 31) // 32 * 2) * ((N + 31) // 32 * 2 + 1),
            }),
            "kernels_ms_per_step": {k: tim[k] / args.steps for k in ("corr_ms", "contract_ms", "acquisition_ms")},
            "whole_step_tflops": eng.flops_per_candidate() * M / (elapsed / args.steps) / 1e12,
            "ask_ms": elapsed / args.steps * 1e3,
            "ask_ms_with_h2d": h2d_ms,
            "ask_ms_device_generated": gen_ms,
            "ask_ms_full": full_ms,
            "commit_s": commit_s,
            "fit": fit_ms,
            "llf": llf,
            "argmax": [int(i) for i in out[1]],
            "exchange": ("bogp_exchange_argmax over RCCL, world %d (executed inside every timed step)" % eng.comm_world) if comm_error is None
            else "torch.distributed all-gather (library communicator failed: %s)" % comm_error,
        }
        if world == 1 and not args.no_cpu:
            res["cpu_baseline"] = cpu_baseline(w, X, y, par, plugin, Xs[: args.cpu_sample].cpu().numpy(), args.cpu_sample, eng)
        # RCCL prints a banner through C stdio, whose buffer would otherwise drain AFTER this line at exit: flush it first so
        # the JSON line is the LAST line of stdout
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


def dry_candidates(first_row, n_rows, d):
    """Rows [first_row, first_row + n_rows) of the dry run's global candidate table: row i = default_rng(10_000 + i).uniform(-5, 5, d)."""
    return np.array([np.random.default_rng(10_000 + i).uniform(-5.0, 5.0, size=d) for i in range(first_row, first_row + n_rows)]).reshape(n_rows, d)


def cpu_baseline(w, X, y, par, plugin, Xh, n_sample, eng):
    """The oracle ('port' of gpr.py:486-510 + the vectorised acquisition closed forms) on a bounded sample of the same candidates, 1024-row
    chunks, BLAS threads = all host cores: 2 warm-up passes, then the MEDIAN of 5 timed passes over disjoint fifths of the sample
    (BASELINE.md section 3.3).  Doubles as a parity check of the GPU run.  Second figure, `as_is`: how the unmodified
    reference consumes the path -- one acquisition call per point (acquisition_fun.py:153-176 through gpr.py:486-510 on a (1, d) array) --
    beside the device's one-point call (bogp_point_eval)."""
    from oracle import gp_oracle as O

    pools = []
    try:
        from threadpoolctl import threadpool_info

        pools = [{k: p.get(k) for k in ("user_api", "internal_api", "num_threads", "version")} for p in threadpool_info()]
        threads = max([p.get("num_threads") or 1 for p in pools] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    n = min(n_sample, len(Xh))
    n_pass = 5
    per = (n // n_pass) // 1024 * 1024 or n // n_pass
    st = O.make_state(par, X, y, w["kernel"], O.MODE_NOISY, 1e-6)
    for _ in range(2):
        O.sweep(st, Xh[:1024], w["acq"], plugin, True)  # warm-up
    secs, omu, omse, ovals = [], [], [], []
    for i in range(n_pass):
        t0 = time.perf_counter()
        _, _, v, m, s = O.sweep(st, Xh[i * per : (i + 1) * per], w["acq"], plugin, True, return_values=True)
        secs.append(time.perf_counter() - t0)
        omu.append(m.ravel()), omse.append(s.ravel()), ovals.append(v)
    nt = n_pass * per
    omu, omse, ovals = np.concatenate(omu), np.concatenate(omse), np.concatenate(ovals, axis=1)
    oidx = np.array([O.nan_first_argmax(v) for v in ovals])
    # parity of the very rows that were timed: the same rows through the GPU path
    eng.upload_candidates(Xh[:nt])
    mu, mse = eng.predict()
    best, idx = eng.sweep(w["acq"], plugin, True)
    np.testing.assert_allclose(mu, omu, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(mse, omse, rtol=1e-6, atol=1e-12)
    np.testing.assert_array_equal(idx, oidx)
    med = float(np.median(secs))
    # as-is: single-row calls
    n_one = 200 if w["N"] <= 2048 else 40
    t0 = time.perf_counter()
    for i in range(n_one):
        m1, s1 = O.predict(st, Xh[i : i + 1])
        O.acquisition(w["acq"][0][0], w["acq"][0][1], m1.ravel(), s1.ravel(), plugin, float(st.sigma2[0]), True)
    one_cpu = (time.perf_counter() - t0) / n_one
    dev_us = None
    try:
        eng.point_eval(Xh[0], w["acq"][:1], plugin, True)
        t0 = time.perf_counter()
        for i in range(n_one):
            eng.point_eval(Xh[i], w["acq"][:1], plugin, True)
        dev_us = (time.perf_counter() - t0) / n_one * 1e6
    except Exception:  # a model the one-point path does not serve
        pass
    return {"value": per / med, "unit": "candidates/s", "cores": threads, "kind": "port",
            "sample": "2 warm-ups, then the median of %d timed passes of %d rows each (disjoint slices of the timed run's candidates, %d rows in all), 1024-row chunks, "
                      "NumPy/SciPy oracle; parity vs GPU checked on those rows (1e-6, argmax exact)" % (n_pass, per, nt),
            "seconds_per_pass": [round(t, 4) for t in secs], "threadpool_info": pools, "os_cpu_count": os.cpu_count(),
            "as_is": {"value": 1.0 / one_cpu, "unit": "single-row acquisition evaluations/s (value only)", "calls": n_one,
                      "what": "one predict(eval_MSE=True) + one criterion per (1, d) row through the oracle: how the unmodified reference consumes the path (BASELINE.md 3.5)",
                      "device_point_eval_us": dev_us, "device_evaluations_per_s": (1e6 / dev_us) if dev_us else None}}  # fmt: skip


if __name__ == "__main__":
    main()
