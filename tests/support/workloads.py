"""BASELINE.json's configurations at their full per-GPU sizes, as seeded workloads shared by the GPU parity tests and
by `oracle/make_fullsize_golden.py` (which pushes the very same candidates through the imported reference).
Pure data: no oracle, no product imports."""
import numpy as np

# kernel / acquisition ids of include/bogp.h
_SE, _M52 = 0, 3
_EI, _UCB, _MGFI = 0, 2, 3

FULL_SIZE = {
    "C2": dict(N=512, d=10, M=100_000, kernel=_SE, theta=0.02, acq=[(_EI, 0.0)]),
    "C3": dict(N=2048, d=20, M=1_000_000, kernel=_M52, theta=0.01, acq=[(_MGFI, 2.0), (_EI, 0.0)]),
    # configs[3]: ParallelBO, q = 8, t_i = exp(log 2 + 0.5 z_i) (bayes_opt.py:84-86); one rank's shard of 1e6
    "C4": dict(N=2048, d=20, M=1_000_000, kernel=_M52, theta=0.01,
               acq=[(_MGFI, float(t)) for t in np.exp(np.log(2.0) + 0.5 * np.random.default_rng(4).standard_normal(8))]),
    "C5": dict(N=8192, d=50, M=500_000, kernel=_SE, theta=0.004, acq=[(_UCB, 0.5)]),
}  # fmt: skip


def full_size_problem(cfg):
    """(X, y, par, Xs): X ~ U[-5,5]^(N x d), y = sum x^2 standardised, par = [theta]*d + [sigma2 = 0.9] (noisy mode,
    nugget 1e-6, simple kriging), Xs ~ U[-5,5]^(M x d) from the same generator stream (seed 0)."""
    w = FULL_SIZE[cfg]
    rng = np.random.default_rng(0)
    X = rng.uniform(-5.0, 5.0, size=(w["N"], w["d"]))
    y = np.sum(X**2, axis=1)
    y = (y - y.mean()) / y.std() + 0.0 * rng.standard_normal(w["N"])  # the draw keeps the stream of tests' _problem()
    par = np.r_[np.full(w["d"], w["theta"]), 0.9]
    Xs = rng.uniform(-5.0, 5.0, size=(w["M"], w["d"]))
    return X, y.reshape(-1, 1), par, Xs


# configs[3] / configs[4] WHOLE: 8 contiguous shards of the candidate grid, rank r owning rows [r * M, (r + 1) * M).
# Shard 0 is the single-rank workload above (so G22 / G23 stay its fixtures); shards 1..7 come from their own seeded streams.
SHARDED = {"C4": dict(R=8, M_total=8_000_000), "C5": dict(R=8, M_total=4_000_000)}


def shard_candidates(cfg, r):
    """Rank r's (M x d) block of the 8-shard candidate grid of `cfg`; its global row offset is r * M."""
    w = FULL_SIZE[cfg]
    if r == 0:
        return full_size_problem(cfg)[3]
    return np.random.default_rng(10_000 + 100 * int(cfg[1:]) + r).uniform(-5.0, 5.0, size=(w["M"], w["d"]))
