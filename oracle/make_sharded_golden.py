"""configs[3] and configs[4] WHOLE: the reference's answer for the 8-shard candidate grids ("missing" item 1).

Run (build container only; hours of CPU, resumable):
    python oracle/make_sharded_golden.py posterior C4 1 2 3     # the reference's predict over every row of shards 1, 2, 3
    python oracle/make_sharded_golden.py screen    C5 0 1 ... 7  # the BLAS screen over every row of those shards
    python oracle/make_sharded_golden.py assemble  C4            # -> tests/golden/G34_c4_sharded.npz
    python oracle/make_sharded_golden.py assemble  C5            # -> tests/golden/G35_c5_sharded.npz

What `ParallelBO._batch_arg_max_acquisition` (`bayes_opt.py:100-115`) consumes is, per criterion, the best candidate of the
WHOLE grid.  §8(e) shards that grid into 8 contiguous blocks; the fixtures hold the reference's answer per shard and after
the global reduce (max value, ties -> lowest global row, i.e. `np.argmax` over the concatenation).

C4 (N = 2048, d = 20, Matern-5/2, q = 8 MGFI, 8 x 1e6 rows): EVERY row of every shard goes through the imported reference's
`GaussianProcess.predict(eval_MSE=True)` in 1024-row chunks, as `make_fullsize_golden.py` does for shard 0 (whose cached
posterior is reused).  The criteria are the oracle's vectorised closed forms, and the stored top rows once more through
the reference's own `MGFI` class row by row.

C5 (N = 8192, d = 50, SE, UCB, 8 x 5e5 rows): the reference needs 7 600 s per shard (measured on shard 0, G23), 15 h
for the other seven.  BASELINE.md §3.3 allows sub-sampled rows for the CPU side, so C5 is two-tiered:
  tier 1, the SCREEN, over every row: the same posterior (`gpr.py:486-510`) with the weighted squared distance formed as
      |a|^2 + |b|^2 - 2 a.b by one dgemm instead of the (M N, d) |dx| temporary; everything after the distance is the
      reference's operation order (exp, r.gamma, solve_triangular, 1 - sum rt^2).  It ranks the rows and gives the sums.
  tier 2, the REFERENCE itself, on the rows that matter: the 64 best rows of each shard by the screen and a fixed slice
      of 512 rows per shard.  The fixture's values, the per-shard top-16 and the global top-16 are the reference's numbers.
  The screen is certified inside this script: against the reference on every tier-2 row, and on ALL 5e5 rows of shard 0
  (whose full reference posterior G23 was made from); the deviations and the screen-value margin between rank 16 and rank
  64 (how far a row outside the tier-2 set is from entering the top-16) are stored in the fixture.
"""
import os
import sys
import time
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "shims"))

import numpy as np  # noqa: E402
import scipy  # noqa: E402
from scipy.linalg import solve_triangular  # noqa: E402

from oracle import gp_oracle as O  # noqa: E402
from oracle.make_fullsize_golden import reference_model, reference_rowwise  # noqa: E402  (imports the reference)
from oracle.make_golden import pin  # noqa: E402
from tests.support.workloads import FULL_SIZE, SHARDED, full_size_problem, shard_candidates  # noqa: E402

warnings.filterwarnings("ignore")
OUT = os.path.join(ROOT, "tests", "golden")
CACHE = "/tmp/bogp_fullsize"
FILES = {"C4": "G34_c4_sharded", "C5": "G35_c5_sharded"}
TOPK, SCREEN_KEEP, SLICE = 16, 64, 512


def _cache(cfg, r, kind):
    w = FULL_SIZE[cfg]
    if r == 0 and kind == "ref":  # make_fullsize_golden.py's cache of the single-rank workload
        key = "N%d_d%d_M%d_k%d" % (w["N"], w["d"], w["M"], w["kernel"])
    else:
        key = "%s_shard%d_%s" % (cfg, r, kind)
    return tuple(os.path.join(CACHE, "%s_%s.npy" % (key, k)) for k in ("mu", "mse", "pos"))


def fitted(cfg):
    w = FULL_SIZE[cfg]
    X, y, par, _ = full_size_problem(cfg)
    gp = reference_model(w["kernel"], w["d"])
    llf = pin(gp, X, y, par)
    return gp, X, y, par, llf


def reference_posterior(cfg, r, gp):
    """All rows of shard r through the reference's predict; resumable."""
    os.makedirs(CACHE, exist_ok=True)
    fmu, fmse, fpos = _cache(cfg, r, "ref")
    Xs = shard_candidates(cfg, r)
    M = len(Xs)
    if os.path.exists(fpos):
        mu, mse, pos = np.load(fmu), np.load(fmse), int(np.load(fpos))
    else:
        mu, mse, pos = np.empty(M), np.empty(M), 0
    chunk = 1024 if gp.X.shape[0] <= 2048 else 256
    t0 = time.time()
    while pos < M:
        b = min(M, pos + chunk)
        m, s = gp.predict(Xs[pos:b], eval_MSE=True)
        mu[pos:b], mse[pos:b] = m[:, 0], s[:, 0]
        pos = b
        if (pos // chunk) % 64 == 0 or pos == M:
            np.save(fmu, mu), np.save(fmse, mse), np.save(fpos, np.array(pos))
            print("%s shard %d: %d / %d rows, %.0f s" % (cfg, r, pos, M, time.time() - t0), flush=True)
    return mu, mse


def screen_posterior(cfg, r, gp, chunk=4096):
    """Tier 1 of C5: SE posterior of all rows of shard r with the distance by dgemm.  gpr.py:486-510 after the distance."""
    assert FULL_SIZE[cfg]["kernel"] == O.KERNEL_SE and not gp.estimate_trend
    os.makedirs(CACHE, exist_ok=True)
    fmu, fmse, fpos = _cache(cfg, r, "screen")
    Xs = shard_candidates(cfg, r)
    M = len(Xs)
    if os.path.exists(fpos) and int(np.load(fpos)) == M:
        return np.load(fmu), np.load(fmse)
    mu, mse = np.empty(M), np.empty(M)
    sq = np.sqrt(gp.theta_)
    B = gp.X * sq
    b2 = np.sum(B * B, axis=1)
    beta = float(np.ravel(gp.mean.beta)[0]) if np.size(gp.mean.beta) else 0.0
    t0 = time.time()
    for a in range(0, M, chunk):
        A = Xs[a : a + chunk] * sq
        D2 = np.sum(A * A, axis=1)[:, None] + b2[None, :] - 2.0 * A.dot(B.T)
        np.maximum(D2, 0.0, out=D2)
        rr = np.exp(-D2)
        mu[a : a + chunk] = beta + rr.dot(gp.gamma)[:, 0]
        rt = solve_triangular(gp.C, rr.T, lower=True)
        v = (1.0 - (rt**2.0).sum(axis=0)) * float(gp.sigma2[0])
        mse[a : a + chunk] = np.where(v < 0.0, 0.0, v)
        if (a // chunk) % 8 == 0:
            print("%s shard %d screen: %d / %d rows, %.0f s" % (cfg, r, a + chunk, M, time.time() - t0), flush=True)
    np.save(fmu, mu), np.save(fmse, mse), np.save(fpos, np.array(M))
    return mu, mse


def stable_top(v, k):
    """The k best rows: value descending, ties -> lowest index (repeated np.argmax)."""
    order = np.argsort(-v, kind="stable")[:k]
    assert order[0] == int(np.argmax(v))
    return order


def global_reduce(top_idx, top_val, Mshard, k):
    """Per criterion: merge the shards' (value, global row) records as SURVEY §8(e) prescribes."""
    R, q, _ = top_idx.shape
    gi, gv = np.empty((q, k), np.int64), np.empty((q, k))
    for c in range(q):
        rows = np.concatenate([top_idx[r, c] + r * Mshard for r in range(R)])
        vals = np.concatenate([top_val[r, c] for r in range(R)])
        order = np.lexsort((rows, -vals))[:k]
        gi[c], gv[c] = rows[order], vals[order]
    return gi, gv


def assemble_c4():
    cfg, w, R = "C4", FULL_SIZE["C4"], SHARDED["C4"]["R"]
    gp, X, y, par, llf = fitted(cfg)
    M, q = w["M"], len(w["acq"])
    plugin, sigma2 = O.plugin_value(y, True), float(gp.sigma2[0])
    top_idx = np.empty((R, q, TOPK), np.int64)
    top_val, top_mu, top_mse = (np.empty((R, q, TOPK)) for _ in range(3))
    sum_mu, sum_mse, xsum = np.empty(R), np.empty(R), np.empty(R)
    count_pos = np.empty((R, q), np.int64)
    slice_rows = np.arange(0, M, M // SLICE)[:SLICE]
    slice_mu, slice_mse = np.empty((R, SLICE)), np.empty((R, SLICE))
    ref_rows = np.empty((R, q, TOPK))
    for r in range(R):
        fpos = _cache(cfg, r, "ref")[2]
        assert os.path.exists(fpos) and int(np.load(fpos)) == M, "posterior of shard %d is not complete" % r
        mu, mse = reference_posterior(cfg, r, gp)
        Xs = shard_candidates(cfg, r)
        xsum[r] = float(np.sum(Xs[::997]))
        for c, (a, p) in enumerate(w["acq"]):
            v = O.acquisition(a, p, mu, mse, plugin, sigma2, True)
            assert not np.isnan(v).any()
            o = stable_top(v, TOPK)
            top_idx[r, c], top_val[r, c], top_mu[r, c], top_mse[r, c] = o, v[o], mu[o], mse[o]
            count_pos[r, c] = int(np.count_nonzero(v > 0))
        sum_mu[r], sum_mse[r] = float(np.sum(mu)), float(np.sum(mse))
        slice_mu[r], slice_mse[r] = mu[slice_rows], mse[slice_rows]
        ref_rows[r] = reference_rowwise(gp, w["acq"], plugin, Xs[top_idx[r]])
        np.testing.assert_allclose(ref_rows[r], top_val[r], rtol=1e-8, atol=1e-300)
        print("C4 shard %d: argmax %s" % (r, top_idx[r, :, 0].tolist()), flush=True)
    gi, gv = global_reduce(top_idx, top_val, M, TOPK)
    out = dict(cfg=cfg, R=R, N=w["N"], d=w["d"], M=M, kernel=w["kernel"], par=par, llf=float(llf), plugin=plugin, sigma2=sigma2,
               acq=np.array(w["acq"], float), top_idx=top_idx, top_val=top_val, top_mu=top_mu, top_mse=top_mse, ref_rowwise=ref_rows,
               count_pos=count_pos, sum_mu=sum_mu, sum_mse=sum_mse, x_checksum=xsum, slice_rows=slice_rows, slice_mu=slice_mu,
               slice_mse=slice_mse, global_idx=gi, global_val=gv, numpy=np.__version__, scipy=scipy.__version__)  # fmt: skip
    np.savez_compressed(os.path.join(OUT, FILES[cfg] + ".npz"), **out)
    print("C4 whole: global argmax", gi[:, 0].tolist(), "in shards", (gi[:, 0] // M).tolist(), flush=True)


def assemble_c5():
    cfg, w, R = "C5", FULL_SIZE["C5"], SHARDED["C5"]["R"]
    gp, X, y, par, llf = fitted(cfg)
    M = w["M"]
    (a, p), plugin, sigma2 = w["acq"][0], O.plugin_value(y, True), float(gp.sigma2[0])
    keep_idx = np.empty((R, SCREEN_KEEP), np.int64)
    keep_mu, keep_mse, keep_val, keep_screen_val = (np.empty((R, SCREEN_KEEP)) for _ in range(4))
    top_idx = np.empty((R, 1, TOPK), np.int64)
    top_val, top_mu, top_mse = (np.empty((R, 1, TOPK)) for _ in range(3))
    slice_rows = np.arange(0, M, M // SLICE)[:SLICE]
    slice_mu, slice_mse = np.empty((R, SLICE)), np.empty((R, SLICE))
    sum_mu, sum_mse, xsum, margin = np.empty(R), np.empty(R), np.empty(R), np.empty(R)
    dev_mu, dev_mse = np.empty(R), np.empty(R)
    for r in range(R):
        smu, smse = screen_posterior(cfg, r, gp)
        Xs = shard_candidates(cfg, r)
        xsum[r] = float(np.sum(Xs[::997]))
        sv = O.acquisition(a, p, smu, smse, plugin, sigma2, True)
        keep = stable_top(sv, SCREEN_KEEP)
        rows = np.concatenate([keep, slice_rows])
        m, s = [], []
        for b in range(0, len(rows), 64):  # tier 2: the reference itself
            mm, ss = gp.predict(Xs[rows[b : b + 64]], eval_MSE=True)
            m.append(mm[:, 0]), s.append(ss[:, 0])
        m, s = np.concatenate(m), np.concatenate(s)
        dev_mu[r] = float(np.max(np.abs(m - smu[rows])))
        dev_mse[r] = float(np.max(np.abs(s - smse[rows]) / sigma2))
        rv = O.acquisition(a, p, m[:SCREEN_KEEP], s[:SCREEN_KEEP], plugin, sigma2, True)
        order = np.lexsort((keep, -rv))[:TOPK]
        top_idx[r, 0], top_val[r, 0], top_mu[r, 0], top_mse[r, 0] = keep[order], rv[order], m[:SCREEN_KEEP][order], s[:SCREEN_KEEP][order]
        keep_idx[r], keep_mu[r], keep_mse[r], keep_val[r], keep_screen_val[r] = keep, m[:SCREEN_KEEP], s[:SCREEN_KEEP], rv, sv[keep]
        slice_mu[r], slice_mse[r] = m[SCREEN_KEEP:], s[SCREEN_KEEP:]
        sum_mu[r], sum_mse[r] = float(np.sum(smu)), float(np.sum(smse))
        margin[r] = float(top_val[r, 0, TOPK - 1] - sv[keep[-1]])  # rank 16 (reference) above rank 64 (screen)
        assert margin[r] > 1e3 * float(np.max(np.abs(rv - sv[keep]))), "the screen cannot separate the top-16 of shard %d" % r
        print("C5 shard %d: argmax %d, screen deviation mu %.2e mse/s2 %.2e, margin %.2e" % (r, top_idx[r, 0, 0], dev_mu[r], dev_mse[r], margin[r]), flush=True)
    # shard 0, every row: the screen against the full reference posterior G23 was made from
    f0 = _cache(cfg, 0, "ref")
    full_dev = np.array([np.nan, np.nan])
    if os.path.exists(f0[2]) and int(np.load(f0[2])) == M:
        rmu, rmse = np.load(f0[0]), np.load(f0[1])
        smu, smse = screen_posterior(cfg, 0, gp)
        full_dev = np.array([np.max(np.abs(rmu - smu)), np.max(np.abs(rmse - smse)) / sigma2])
        rvf = O.acquisition(a, p, rmu, rmse, plugin, sigma2, True)
        np.testing.assert_array_equal(stable_top(rvf, TOPK), top_idx[0, 0])  # the two-tier top-16 IS the all-rows reference top-16
        print("C5 shard 0, all rows: screen deviation mu %.2e mse/s2 %.2e; two-tier top-16 == reference top-16" % tuple(full_dev), flush=True)
    ref_rows = np.empty((R, 1, TOPK))
    for r in range(R):
        ref_rows[r] = reference_rowwise(gp, w["acq"], plugin, shard_candidates(cfg, r)[top_idx[r]])
        np.testing.assert_allclose(ref_rows[r], top_val[r], rtol=1e-8)
    gi, gv = global_reduce(top_idx, top_val, M, TOPK)
    out = dict(cfg=cfg, R=R, N=w["N"], d=w["d"], M=M, kernel=w["kernel"], par=par, llf=float(llf), plugin=plugin, sigma2=sigma2,
               acq=np.array(w["acq"], float), top_idx=top_idx, top_val=top_val, top_mu=top_mu, top_mse=top_mse, ref_rowwise=ref_rows,
               keep_idx=keep_idx, keep_mu=keep_mu, keep_mse=keep_mse, keep_val=keep_val, keep_screen_val=keep_screen_val,
               screen_sum_mu=sum_mu, screen_sum_mse=sum_mse, screen_dev_mu=dev_mu, screen_dev_mse=dev_mse, screen_dev_all_rows_shard0=full_dev,
               margin_rank16_vs_rank64=margin, x_checksum=xsum, slice_rows=slice_rows, slice_mu=slice_mu, slice_mse=slice_mse,
               global_idx=gi, global_val=gv, numpy=np.__version__, scipy=scipy.__version__)  # fmt: skip
    np.savez_compressed(os.path.join(OUT, FILES[cfg] + ".npz"), **out)
    print("C5 whole: global argmax", gi[:, 0].tolist(), "in shard", (gi[:, 0] // M).tolist(), flush=True)


if __name__ == "__main__":
    what, cfg = sys.argv[1], sys.argv[2]
    if what == "assemble":
        {"C4": assemble_c4, "C5": assemble_c5}[cfg]()
    else:
        gp = fitted(cfg)[0]
        for r in map(int, sys.argv[3:]):
            (reference_posterior if what == "posterior" else screen_posterior)(cfg, r, gp)
