"""One exchange step per sweep: all-gather of q x (value, global index [, point]) + identical deterministic reduce.

RCCL has no MAXLOC and an (f64, i64) pair does not pack into one max-reducible 64-bit key, hence gather-then-reduce
(SURVEY.md section 8e).  Over xGMI this is latency-bound (q * 16 B per rank); it is issued once per `ask()`, never
per tile.

Two transports, one rule:
  * the library's own (`init_engine_comm` -> `bogp_comm_init`; `Engine.exchange_argmax / exchange_topk`): the winners are
    packed on the device from the sweep's result buffers and gathered with ONE ncclAllGather (RCCL over xGMI), no host
    bounce; a plain-C client of include/bogp.h shards the same way.  `optim.sweep_*` use it whenever the model's engine
    has a communicator;
  * `exchange_argmax / exchange_topk` below on host arrays through any initialised `torch.distributed` backend ("gloo"
    in the CPU tests, where no device exists).  Without a process group they are the identity (single GPU).
Both apply np.argmax over the concatenation of the shards: largest value, a NaN beats every number, ties -> lowest
global index (`reduce_pairs` here, `bogp_reduce_pairs` in the library; checked against each other in tests/test_abi.py).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np


def _dist():
    try:
        import torch.distributed as dist
    except Exception:  # torch absent: single-process use only
        return None
    return dist if dist.is_available() and dist.is_initialized() else None


def rank_world(group=None):
    """(rank, world size) of the initialised process group, (0, 1) without one."""
    dist = _dist()
    return (0, 1) if dist is None else (dist.get_rank(group), dist.get_world_size(group))


def init_engine_comm(engine, group=None):
    """Give `engine` (one per process / GPU) the library's RCCL communicator spanning the ranks of the initialised
    torch.distributed group: rank 0 creates the ncclUniqueId, the 128 bytes travel through the process group (any
    backend), every rank joins with `bogp_comm_init`.  Without a process group: a one-rank communicator (the exchange
    then still runs -- a device-side all-gather of one shard).  Returns (rank, world)."""
    from . import _lib

    if getattr(engine, "comm_world", 0):
        return engine.comm_rank, engine.comm_world
    dist = _dist()
    make_id = getattr(engine, "comm_unique_id", _lib.comm_unique_id)  # (an engine may bring its own transport's id: CPU stand-ins in tests)
    if dist is None:
        engine.comm_init(make_id(), 0, 1)
        return 0, 1
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    # (rank 0 may fail to create the id -- librccl missing, say: the failure travels in the broadcast, so that every rank raises
    # instead of waiting for an id that never comes)
    box = [None]
    if rank == 0:
        try:
            box = [make_id()]
        except Exception as e:  # noqa: BLE001
            box = [RuntimeError("rank 0 could not create the communicator id: %s" % e)]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if isinstance(box[0], Exception):
        raise box[0]
    engine.comm_init(box[0], rank, world)
    return rank, world


def reduce_pairs(vals: np.ndarray, idxs: np.ndarray) -> np.ndarray:
    """vals, idxs: (R, q).  Per criterion pick the winning rank with np.argmax semantics over the concatenated
    global array: the maximum value, a NaN (if any) beats every number, ties -> lowest global index."""
    vals = np.asarray(vals, dtype=np.float64)
    idxs = np.asarray(idxs, dtype=np.int64)
    R, q = vals.shape
    win = np.zeros(q, dtype=np.int64)
    for c in range(q):
        b = 0
        for r in range(1, R):
            av, bv = vals[r, c], vals[b, c]
            an, bn = np.isnan(av), np.isnan(bv)
            if an != bn:
                better = an
            elif not an and av != bv:
                better = av > bv
            else:
                better = idxs[r, c] < idxs[b, c]
            if better:
                b = r
        win[c] = b
    return win


def exchange_argmax(best_val: np.ndarray, best_gidx: np.ndarray, best_x: Optional[np.ndarray] = None, group=None):
    """All ranks call this with their local winners; all ranks return the same global winners
    (best_val (q,), best_gidx (q,), best_x (q, d) or None)."""
    best_val = np.ascontiguousarray(best_val, dtype=np.float64)
    best_gidx = np.ascontiguousarray(best_gidx, dtype=np.int64)
    dist = _dist()
    if dist is None or (dist.get_world_size(group) == 1 and not os.environ.get("BOGP_FORCE_EXCHANGE")):
        return best_val, best_gidx, best_x
    import torch

    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    q = len(best_val)
    d = 0 if best_x is None else best_x.shape[1]
    # one buffer: [val | idx (bit pattern) | x] per criterion, all float64-sized words -> ONE collective
    pack = np.empty((q, 2 + d), dtype=np.float64)
    pack[:, 0] = best_val
    pack[:, 1] = best_gidx.view(np.float64)
    if d:
        pack[:, 2:] = best_x
    mine = torch.from_numpy(pack).to(dev)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    allp = torch.stack(gathered).cpu().numpy()  # (R, q, 2 + d)
    vals = allp[:, :, 0]
    idxs = np.ascontiguousarray(allp[:, :, 1]).view(np.int64)
    win = reduce_pairs(vals, idxs)
    ar = np.arange(q)
    out_x = allp[win, ar, 2:] if d else None
    return vals[win, ar].copy(), idxs[win, ar].copy(), out_x


def merge_topk(vals: np.ndarray, idxs: np.ndarray, k: int):
    """vals, idxs: (R, k_local) winners of R shards for ONE criterion (each row sorted best-first, padded with
    (-inf, -1)).  Returns the global k best as (values (k,), indices (k,), source rank (k,), source slot (k,)) in
    np.argmax order: larger value first, NaN before every number, ties -> lower global index."""
    ent = []
    R, kl = vals.shape
    for r in range(R):
        for s in range(kl):
            if idxs[r, s] >= 0:
                v = float(vals[r, s])
                ent.append((0 if np.isnan(v) else 1, -v if not np.isnan(v) else 0.0, int(idxs[r, s]), r, s))
    ent.sort()
    ent = ent[:k]
    ov = np.full(k, -np.inf)
    oi = np.full(k, -1, dtype=np.int64)
    orank = np.full(k, -1, dtype=np.int64)
    oslot = np.full(k, -1, dtype=np.int64)
    for j, (_, _, gi, r, s) in enumerate(ent):
        ov[j], oi[j], orank[j], oslot[j] = vals[r, s], gi, r, s
    return ov, oi, orank, oslot


def exchange_topk(best_val: np.ndarray, best_gidx: np.ndarray, best_x: Optional[np.ndarray], k: int, group=None):
    """Top-k flavour of `exchange_argmax`: inputs (q, k) values / global indices (+ (q, k, d) points) of this rank;
    every rank returns the same global (q, k) winners.  Still ONE all-gather."""
    best_val = np.ascontiguousarray(best_val, dtype=np.float64)
    best_gidx = np.ascontiguousarray(best_gidx, dtype=np.int64)
    q = best_val.shape[0]
    d = 0 if best_x is None else best_x.shape[-1]
    dist = _dist()
    if dist is None or (dist.get_world_size(group) == 1 and not os.environ.get("BOGP_FORCE_EXCHANGE")):
        allp = None
        vals, idxs = best_val[None], best_gidx[None]
        xs = None if best_x is None else best_x[None]
    else:
        import torch

        world = dist.get_world_size(group)
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        pack = np.empty((q, k, 2 + d), dtype=np.float64)
        pack[..., 0] = best_val
        pack[..., 1] = best_gidx.view(np.float64)
        if d:
            pack[..., 2:] = best_x
        mine = torch.from_numpy(pack).to(dev)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine, group=group)
        allp = torch.stack(gathered).cpu().numpy()  # (R, q, k, 2 + d)
        vals = allp[..., 0]
        idxs = np.ascontiguousarray(allp[..., 1]).view(np.int64)
        xs = allp[..., 2:] if d else None
    out_v = np.empty((q, k))
    out_i = np.empty((q, k), dtype=np.int64)
    out_x = np.full((q, k, d), np.nan) if d else None
    for c in range(q):
        v, i, rr, ss = merge_topk(vals[:, c, :], idxs[:, c, :], k)
        out_v[c], out_i[c] = v, i
        if d:
            for j in range(k):
                if rr[j] >= 0:
                    out_x[c, j] = xs[rr[j], c, ss[j]]
    return out_v, out_i, out_x


def exchange_best_parameters(param: np.ndarray, neg_llf: float, group=None):
    """MLE restarts spread over ranks: all-gather (objective, parameter vector) and return the
    pair with the smallest objective (ties -> lowest rank), identical on every rank."""
    dist = _dist()
    if dist is None or dist.get_world_size(group) == 1:
        return param, neg_llf
    import torch

    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    v = neg_llf if np.isfinite(neg_llf) else np.inf
    mine = torch.from_numpy(np.r_[v, np.asarray(param, dtype=np.float64)]).to(dev)
    gathered = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(gathered, mine, group=group)
    allp = torch.stack(gathered).cpu().numpy()
    obj = np.where(np.isnan(allp[:, 0]), np.inf, allp[:, 0])
    w = int(np.argmin(obj))  # first minimum = lowest rank
    return allp[w, 1:].copy(), float(allp[w, 0])
