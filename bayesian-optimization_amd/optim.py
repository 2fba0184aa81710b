"""Inner maximiser of the acquisition function: the M-candidate GPU sweep behind `argmax_restart`'s signature.

The reference maximises one criterion with multi-restart L-BFGS-B / (1+1)-CMA-ES / MIES, ONE point per call
(`bayes_optim/acquisition/optim/__init__.py:55-153`; <= 100 d points per `ask()`).  The sweep is a new option
behind the same signature (`optimizer="sweep"`): sample `eval_budget` candidates in the box, evaluate posterior +
criterion for all of them in one device pass, return the argmax as `(xopt: list, fopt: float)`.
q criteria that differ only in their parameter (ParallelBO's t / alpha draws, `bayes_opt.py:100-115`) share one
posterior pass: `sweep_argmax(criteria=[...])`.

Candidates shard across ranks (one process per GPU): every rank sweeps its own block and ONE exchange of
q x (value, global index) decides the winner (`distributed.exchange_argmax`).
"""
from __future__ import annotations

import functools
import inspect
import warnings
from typing import Callable, List, Optional, Sequence

import numpy as np
from scipy.optimize import fmin_l_bfgs_b

from . import distributed


_TRANS = {  # Real.scale -> (forward, inverse) (variable.py:22-55)
    "linear": (lambda v: v, lambda v: v),
    "log": (np.log, np.exp),
    "log10": (np.log10, lambda v: np.power(10, v)),
    "logit": (lambda v: np.log(v / (1 - v)), lambda v: 1 / (1 + np.exp(-v))),
    "bilog": (lambda v: np.sign(v) * np.log(1 + np.abs(v)), lambda v: np.sign(v) * (np.exp(np.abs(v)) - 1)),
}


class Box:
    """Minimal continuous search space with the members of `RealSpace` the maximiser touches
    (`search_space.py:724-769`: `bounds`, `dim`, `sample(N, method)`), incl. the per-variable `scale` and `precision` of
    `Real` (variable.py:165-257): sampling is uniform on the transformed scale, then mapped back, rounded and clipped."""

    def __init__(self, bounds, random_seed=None, precision=None, scale=None):
        self.bounds = [tuple(map(float, b)) for b in bounds]
        self.dim = len(self.bounds)
        self._rng = np.random.default_rng(random_seed)
        self.precision = list(precision) if np.ndim(precision) else [precision] * self.dim
        self.scale = [(s or "linear") for s in (list(scale) if np.ndim(scale) else [scale] * self.dim)]

    def sample(self, N=1, method="uniform"):
        lo_t, hi_t, scales, precs, lo, hi = design_of(self)
        X = self._rng.uniform(lo_t, hi_t, size=(int(N), self.dim))
        for k in range(self.dim):
            X[:, k] = _TRANS[scales[k]][1](X[:, k])
            if precs[k] is not None:
                X[:, k] = np.clip(np.round(X[:, k], precs[k]), lo[k], hi[k])
        return X


def design_of(space):
    """(lo_t, hi_t, scales, precisions, lo, hi) of a continuous search space: the box the design is DRAWN in (transformed
    bounds, `Real._bounds_transformed`), the per-variable scale names and precisions, and the variables' own bounds.
    Accepts a plain list of (lo, hi) pairs, a `Box`, or the reference's `RealSpace` (its `data` list of `Real`)."""
    if isinstance(space, Box):
        lo = np.array([b[0] for b in space.bounds], dtype=float)
        hi = np.array([b[1] for b in space.bounds], dtype=float)
        scales, precs = list(space.scale), list(space.precision)
    elif hasattr(space, "data") and hasattr(space, "bounds"):  # bayes_optim.RealSpace
        lo = np.array([v.bounds[0] for v in space.data], dtype=float)
        hi = np.array([v.bounds[1] for v in space.data], dtype=float)
        scales = [getattr(v, "scale", "linear") or "linear" for v in space.data]
        precs = [getattr(v, "precision", None) for v in space.data]
    else:
        b = list(getattr(space, "bounds", space))
        lo = np.array([x[0] for x in b], dtype=float)
        hi = np.array([x[1] for x in b], dtype=float)
        scales, precs = ["linear"] * len(b), [None] * len(b)
    lo_t = np.array([_TRANS[s][0](np.float64(a)) for s, a in zip(scales, lo)], dtype=float)
    hi_t = np.array([_TRANS[s][0](np.float64(a)) for s, a in zip(scales, hi)], dtype=float)
    return lo_t, hi_t, scales, precs, lo, hi


def _generate(eng, space, count, seed, first_row, method, n_total):
    """Draw `count` rows of the design of `space` on the device, with RealSpace._sample's post-processing
    (search_space.py:754: to_linear_scale, then round) applied there too."""
    lo_t, hi_t, scales, precs, lo, hi = design_of(space)
    plain = all(s == "linear" for s in scales) and all(p is None for p in precs)
    eng.set_candidate_transform(None if plain else scales, None if plain else precs, lo, hi)
    eng.generate_candidates(lo_t, hi_t, count, seed, first_row=first_row, method=method, n_total=n_total)


def device_sample(engine, space, N: int, method: str = "LHS-maximin", seed: int = 0, maximin: int = 5) -> np.ndarray:
    """`space.sample(N, method)` of the reference (search_space.py:742-754) drawn on the device and read back: (N, d).
    `method`: "uniform" | "LHS" | "LHS-maximin" | "sobol" -- the reference's own "LHS" is pyDOE's maximin criterion over
    five hypercubes, i.e. "LHS-maximin" here (the plain "LHS" is one hypercube, the only flavour that can be drawn in
    row shards).  The engine must know d (a training set has been set); its candidate buffer is overwritten."""
    lo_t, hi_t, scales, precs, lo, hi = design_of(space)
    plain = all(s == "linear" for s in scales) and all(p is None for p in precs)
    engine.set_candidate_transform(None if plain else scales, None if plain else precs, lo, hi)
    engine.generate_candidates(lo_t, hi_t, int(N), seed, method=method, maximin=maximin)
    return engine.read_candidates(np.arange(int(N)))


def shard_bounds(M: int, rank: int, world: int):
    """Contiguous block of rank r: rows [r M / R, (r+1) M / R)."""
    return (rank * M) // world, ((rank + 1) * M) // world


def is_continuous(space) -> bool:
    """True for the spaces the sweep and L-BFGS-B serve: a `Box`, the reference's `RealSpace` (recognised by class name
    through the MRO -- `isinstance(search_space, RealSpace)` is the reference's own test, optim/__init__.py:71), or a
    plain sequence of (lo, hi) pairs."""
    if space is None:
        return False
    if isinstance(space, Box) or "RealSpace" in [c.__name__ for c in type(space).__mro__]:
        return True
    if any(c.__name__ == "SearchSpace" for c in type(space).__mro__):
        return False
    try:
        return all(len(b) == 2 for b in space)
    except TypeError:
        return False


def engine_rank_world(eng, group=None):
    """(rank, world) a sweep shards by: the engine's own communicator when it has one (`bogp_comm_init` with an id that
    travelled out of band, or `bogp_comm_attach`: no torch process group need exist), else the torch process group."""
    if getattr(eng, "comm_world", 0):
        return int(eng.comm_rank), int(eng.comm_world)
    return distributed.rank_world(group)


def feasible_rows(X: np.ndarray, h: Optional[Callable], g: Optional[Callable]) -> np.ndarray:
    """Boolean mask of the rows of X the reference would ACCEPT as the outcome of a restart (optim/__init__.py:125-127):
    |h(x)| within 1e-1 of zero for every equality constraint and g(x) <= 0 for every inequality constraint.  `h` / `g`
    take one point as a list, like the wrappers `BaseBO` binds (base.py:224-229, 236-237)."""
    keep = np.ones(len(X), dtype=bool)
    if len(X) > 100_000 and (h is not None or g is not None):  #  one Python call per candidate and constraint
        warnings.warn("a constrained sweep calls the constraint functions once per candidate (%d Python calls): lower eval_budget "
                      "or keep the problem on the reference's optimiser" % len(X), RuntimeWarning, stacklevel=3)
    for i in range(len(X)):
        x = X[i].tolist()
        if h is not None:
            keep[i] = bool(np.all(np.isclose(np.abs(h(x)), 0, atol=1e-1)))
        if keep[i] and g is not None:
            keep[i] = bool(np.all(np.asarray(g(x)) <= 0))
    return keep


def candidate_block(bounds, M: int, seed: int, rank: int = 0, world: int = 1) -> np.ndarray:
    """This rank's block of M uniform candidates in the box: a deterministic function of (seed, rank, world)."""
    lo = np.array([b[0] for b in bounds], dtype=float)
    hi = np.array([b[1] for b in bounds], dtype=float)
    a, b_ = shard_bounds(M, rank, world)
    rng = np.random.default_rng([int(seed), int(rank), int(world)])
    return rng.uniform(lo, hi, size=(b_ - a, len(lo)))


def sweep_argmax(criteria: Sequence, Xs: np.ndarray, index_offset: int = 0, group=None, return_points: bool = True):
    """Evaluate q criteria (same model, same minimize / plugin) on this rank's candidates `Xs` and reduce across
    ranks.  Returns (best_val (q,), best_global_idx (q,), best_x (q, d) or None)."""
    c0 = criteria[0]
    model = c0.model
    eng = model.engine
    if getattr(model, "_committed_par", None) is None:
        raise Exception("The model is not fitted yet!")
    for c in criteria[1:]:
        if c.model is not model or c.minimize != c0.minimize or c.effective_plugin() != c0.effective_plugin():
            raise ValueError("criteria sharing one sweep must share model, minimize and plugin")
    Xs = model._check_X(Xs)
    eng.upload_candidates(Xs, lazy=True)  # (the sweep right below overlaps the copy with its first chunks)
    acq = [(c.acq_id, c.acq_par()) for c in criteria]
    best, idx = eng.sweep(acq, c0.effective_plugin(), c0.minimize)
    if getattr(eng, "comm_world", 0):  # the library's own exchange: device records, ONE ncclAllGather, no host bounce
        return eng.exchange_argmax(len(acq), int(index_offset), return_points)
    gidx = idx + int(index_offset)
    xbest = Xs[idx] if return_points else None
    return distributed.exchange_argmax(best, gidx, xbest, group=group)


def sweep_generated(criteria: Sequence, bounds, M: int, seed: int, rank: int = 0, world: int = 1, group=None,
                    method: str = "uniform"):
    """`sweep_argmax` over M candidates in `bounds` (a list of (lo, hi) pairs, a `Box`, or the reference's `RealSpace` -- scales
    and precisions of its variables are honoured on the device) that never touch the host: rank r draws rows
    [r M / R, (r+1) M / R) of the design on its GPU, sweeps them, and the ranks exchange their winners (value, global
    row, point).  `method` names the design like `RealSpace._sample` does (search_space.py:742-754): "uniform" (Philox
    stream `seed`), "LHS" (an M-point Latin hypercube of stream `seed`; "LHS-maximin": pyDOE's criterion, one rank only),
    "sobol" (points 1..M of the unscrambled
    sequence; `seed` unused).  The union of the shards is the same M-point set for every world size."""
    c0 = criteria[0]
    model = c0.model
    if getattr(model, "_committed_par", None) is None:
        raise Exception("The model is not fitted yet!")
    a, b_ = shard_bounds(int(M), rank, world)
    eng = model.engine
    _generate(eng, bounds, b_ - a, seed, a, method, int(M))
    best, idx = eng.sweep([(c.acq_id, c.acq_par()) for c in criteria], c0.effective_plugin(), c0.minimize)
    if getattr(eng, "comm_world", 0):
        return eng.exchange_argmax(len(criteria), a, True)
    return distributed.exchange_argmax(best, idx + a, eng.read_candidates(idx), group=group)


def sweep_topk(criteria: Sequence, Xs: np.ndarray, k: int, index_offset: int = 0, group=None):
    """As `sweep_argmax`, returning the k best candidates per criterion:
    (values (q, k), global indices (q, k), points (q, k, d)), identical on every rank."""
    c0 = criteria[0]
    model = c0.model
    if getattr(model, "_committed_par", None) is None:
        raise Exception("The model is not fitted yet!")
    Xs = model._check_X(Xs)
    eng = model.engine
    eng.upload_candidates(Xs, lazy=True)  # (the sweep right below overlaps the copy with its first chunks)
    best, idx = eng.sweep_topk([(c.acq_id, c.acq_par()) for c in criteria], c0.effective_plugin(), c0.minimize, k)
    if getattr(eng, "comm_world", 0):
        return eng.exchange_topk(len(criteria), k, int(index_offset), True)
    xb = np.where((idx >= 0)[..., None], Xs[np.clip(idx, 0, len(Xs) - 1)], np.nan)
    gidx = np.where(idx >= 0, idx + int(index_offset), -1)
    return distributed.exchange_topk(best, gidx, xb, k, group=group)


def sweep_topk_generated(criteria: Sequence, bounds, M: int, k: int, seed: int, rank: int = 0, world: int = 1, group=None,
                         method: str = "uniform"):
    """`sweep_topk` over M candidates generated on the device (`method` as in `sweep_generated`): rank r draws and sweeps
    its block of the design; (values (q, k), global rows (q, k), points (q, k, d)) are identical on every rank."""
    c0 = criteria[0]
    model = c0.model
    if getattr(model, "_committed_par", None) is None:
        raise Exception("The model is not fitted yet!")
    a, b_ = shard_bounds(int(M), rank, world)
    eng = model.engine
    _generate(eng, bounds, b_ - a, seed, a, method, int(M))
    best, idx = eng.sweep_topk([(c.acq_id, c.acq_par()) for c in criteria], c0.effective_plugin(), c0.minimize, k)
    if getattr(eng, "comm_world", 0):
        return eng.exchange_topk(len(criteria), k, a, True)
    flat = np.clip(idx, 0, b_ - a - 1).ravel()
    xb = np.where((idx >= 0)[..., None], eng.read_candidates(flat).reshape(idx.shape + (eng.d,)), np.nan)
    gidx = np.where(idx >= 0, idx + a, -1)
    return distributed.exchange_topk(best, gidx, xb, k, group=group)


def batch_argmax(criteria: Sequence, search_space, eval_budget: int, history: Optional[np.ndarray] = None, k: int = 8,
                 index_offset: Optional[int] = None, group=None, Xs: Optional[np.ndarray] = None, design: Optional[str] = None,
                 seed: Optional[int] = None, rank: Optional[int] = None, world: Optional[int] = None, masks=None, values=None,
                 h: Optional[Callable] = None, g: Optional[Callable] = None):
    """The q-point proposal of `ParallelBO._batch_arg_max_acquisition` (bayes_opt.py:100-115) in ONE posterior pass:
    q criteria (same model; they differ only in t / alpha) share (mu, MSE); each takes its best candidate that is
    neither already taken by an earlier criterion nor `np.isclose` to an evaluated point in `history`
    (BO.pre_eval_check, bayes_opt.py:27-55) -- falling back through its top-k instead of the reference's random
    padding (base.py:282-289).  Returns (xopt: tuple of q lists, fopt: tuple of q floats) like the reference.
    `design` = "uniform" | "LHS" | "sobol" draws the `eval_budget` candidates on the device (rank r of `world` its block
    of ONE design); otherwise every rank sweeps its own host sample (`search_space.sample`, or `Xs`) and the union of the
    world x eval_budget points competes.  `masks` / `values` (ask(fixed=...), utils.py:184-213): `search_space` spans the
    free variables only, the fixed columns are filled in for the model, `history` holds full points, and the returned
    points hold the free variables (the caller's `fillin_fixed_value` completes them, base.py:476).  `h` / `g` (constraints
    over the free variables, one point as a list): only host-sampled candidates the reference would accept as a restart's
    outcome (`feasible_rows`) enter the sweep; with none, `((), ())` -- the reference's "no feasible restart" answer."""
    if rank is None or world is None:
        rank, world = engine_rank_world(criteria[0].model.engine, group)
    if design is not None:  # "uniform" | "LHS" | "sobol": the candidates are drawn on the GPU(s) and never touch the host
        if int(eval_budget) < world:  # (only ONE design is sharded; on the host-sampled path every rank draws its own eval_budget rows)
            raise ValueError("%d candidates cannot be sharded over %d ranks (every rank must own at least one)" % (eval_budget, world))
        if masks is not None or h is not None or g is not None:
            raise NotImplementedError("device-generated designs take neither fixed variables nor constraints")
        seed = int(np.random.randint(0, 2**62)) if seed is None else int(seed)
        vals, gidx, pts = sweep_topk_generated(criteria, search_space, int(eval_budget), k, seed, rank, world, group, design)
    else:
        if Xs is None:
            Xs = np.asarray(search_space.sample(int(eval_budget), method="uniform"), dtype=float)
        if h is not None or g is not None:
            if world > 1:
                raise NotImplementedError("a constrained sweep runs on one rank (a shard without feasible rows cannot join the exchange)")
            Xs = np.asarray(Xs, dtype=float)[feasible_rows(np.asarray(Xs, dtype=float), h, g)]
            if len(Xs) == 0:
                return (), ()
        full = np.asarray(Xs, dtype=float)
        if masks is not None:
            full = np.empty((len(Xs), len(masks)))
            full[:, ~masks] = Xs
            full[:, masks] = np.asarray(values, dtype=float)
        off = rank * len(full) if index_offset is None else int(index_offset)
        vals, gidx, pts = sweep_topk(criteria, full, k, index_offset=off, group=group)
    chosen_x, chosen_f, taken = [], [], set()
    hist = None if history is None or len(history) == 0 else np.asarray(history, dtype=float)
    for c in range(len(criteria)):
        pick = None
        for r in range(k):
            gi = int(gidx[c, r])
            if gi < 0 or gi in taken:
                continue
            if hist is not None and np.any(np.all(np.isclose(hist, pts[c, r]), axis=1)):
                continue
            pick = r
            break
        if pick is None:  # every fall-back exhausted: keep the argmax (the caller's duplicate check will pad)
            pick = 0
        taken.add(int(gidx[c, pick]))
        x = np.asarray(pts[c, pick], dtype=float)
        chosen_x.append((x[~masks] if masks is not None else x).tolist())
        chosen_f.append(float(vals[c, pick]))
    return tuple(chosen_x), tuple(chosen_f)


def unwrap_criterion(obj):
    """Find the acquisition object behind what `BaseBO._create_acquisition` hands to `argmax_restart`
    (base.py:482-494): `partial_argument(functools.partial(criterion, return_dx=...), var_name, fixed)` -- a
    `functools.wraps` wrapper (utils.py:184-213).  Returns (criterion, masks, values): `masks` marks the variables
    the wrapper fills in with the fixed `values` (None when nothing is fixed or `obj` is the criterion itself)."""
    masks = values = None
    for _ in range(8):
        if hasattr(obj, "acq_id") and hasattr(obj, "acq_par"):
            return obj, masks, values
        if isinstance(obj, functools.partial):
            obj = obj.func
        elif hasattr(obj, "__wrapped__"):
            try:
                nl = inspect.getclosurevars(obj).nonlocals
                if "masks" in nl and np.any(nl["masks"]):
                    masks, values = np.asarray(nl["masks"], dtype=bool), list(nl["values"])
            except TypeError:
                pass
            obj = obj.__wrapped__
        else:
            break
    return None, None, None


# optimizer name -> sampling method of RealSpace._sample (search_space.py:742-754) realised on the device
DEVICE_DESIGNS = {"sweep-device": "uniform", "sweep-device-lhs": "LHS", "sweep-device-sobol": "sobol"}


def _reference_argmax_restart():
    """The reference's own `argmax_restart` when `bayes_optim` is importable in this process (it is whenever a reference
    driver is the caller), read from its DEFINING module -- which no integration route re-points -- else None."""
    try:
        import importlib

        return importlib.import_module("bayes_optim.acquisition.optim").argmax_restart
    except Exception:
        return None


def argmax_restart(
    obj_func: Callable,
    search_space,
    h: Callable = None,
    g: Callable = None,
    eval_budget: int = 100,
    n_restart: int = 10,
    wait_iter: int = 3,
    optimizer: str = "BFGS",
    logger=None,
):
    """Same signature and return convention as the reference's `argmax_restart` (optim/__init__.py:55-153).

    optimizer="sweep": `obj_func` must be one of this package's acquisition objects; `eval_budget` candidates are
    drawn with `search_space.sample(N, "uniform")` and swept on the GPU.  With constraints `h` / `g`, only the sampled
    candidates the reference would accept as a restart's outcome (`feasible_rows`) are swept; `([], [])` when there is none.
    optimizer="sweep-device" / "sweep-device-lhs" / "sweep-device-sobol": the candidates (uniform / Latin hypercube /
    Sobol') are generated on the GPU and never touch the host.
    optimizer="sweep-BFGS" / "sweep-device-BFGS": the sweep's best `n_restart` candidates (host-sampled / drawn on the GPU)
    are polished together on the device (`polish_topk`: lock-step projected L-BFGS, one batched value + gradient call
    per iteration).
    optimizer="BFGS": the reference's multi-restart L-BFGS-B loop on `obj_func(x) -> (value, dx)` (host; every
    evaluation is one device call through the acquisition object).
    Anything else ("MIES", "OnePlusOne_Cholesky_CMA", constraints under "BFGS", a non-continuous space) is the
    reference's own business: the call is handed to its `argmax_restart` when `bayes_optim` is importable, for which a
    bogp criterion is an ordinary callable; without the reference, NotImplementedError.
    """
    ours = optimizer in DEVICE_DESIGNS or optimizer in ("sweep", "sweep-BFGS", "sweep-device-BFGS")
    if not ours and (optimizer != "BFGS" or h is not None or g is not None or not is_continuous(search_space)):
        ref = _reference_argmax_restart()
        if ref is None:
            raise NotImplementedError(
                "optimizer %r / constraints / non-continuous spaces are served by the reference's own argmax_restart "
                "(bayes_optim is not importable here); this package serves 'BFGS', 'sweep', 'sweep-device[-lhs|-sobol]' "
                "and 'sweep-BFGS' on continuous spaces" % optimizer)  # fmt: skip
        return ref(obj_func, search_space, h=h, g=g, eval_budget=eval_budget, n_restart=n_restart, wait_iter=wait_iter,
                   optimizer=optimizer, logger=logger)  # fmt: skip
    if optimizer == "sweep-device-BFGS":  # as "sweep-BFGS" with the candidates drawn on the GPU: nothing of size M on the host
        crit, masks, _ = unwrap_criterion(obj_func)
        if crit is None or masks is not None or h is not None or g is not None:
            raise NotImplementedError("optimizer=%r takes an unconstrained bogp criterion without fixed variables" % optimizer)
        rank, world = engine_rank_world(crit.model.engine)
        if int(eval_budget) < world:
            raise ValueError("%d candidates cannot be sharded over %d ranks" % (eval_budget, world))
        k = int(max(1, min(n_restart, 32, int(eval_budget))))
        tv, _, tx = sweep_topk_generated([crit], search_space, int(eval_budget), k, int(np.random.randint(0, 2**62)), rank, world)
        ok = np.isfinite(tv[0])
        if not ok.any():
            return [], []
        xp, fp = polish_topk(crit, tx[0][ok], np.array(search_space.bounds, dtype=float))
        j = int(np.argmax(fp))
        return xp[j].tolist(), float(fp[j])
    if optimizer in DEVICE_DESIGNS:  # candidates drawn on the GPU; the stream is seeded from the global np.random
        crit, masks, _ = unwrap_criterion(obj_func)
        if crit is None or masks is not None or h is not None or g is not None:
            raise NotImplementedError("optimizer=%r takes an unconstrained bogp criterion without fixed variables" % optimizer)
        rank, world = engine_rank_world(crit.model.engine)
        if int(eval_budget) < world:
            raise ValueError("%d candidates cannot be sharded over %d ranks" % (eval_budget, world))
        best, _, xb = sweep_generated([crit], search_space, int(eval_budget), int(np.random.randint(0, 2**62)), rank, world,
                                      method=DEVICE_DESIGNS[optimizer])  # fmt: skip
        return xb[0].tolist(), float(best[0])
    if ours:  # "sweep" | "sweep-BFGS": host-sampled candidates
        crit, masks, values = unwrap_criterion(obj_func)
        if crit is None:
            raise TypeError("optimizer=%r needs a bogp acquisition object (or the reference's wrapper around one)" % optimizer)
        Xs = np.asarray(search_space.sample(int(eval_budget), method="uniform"), dtype=float)
        # Under a communicator / process group every rank sweeps ITS OWN draw of `eval_budget` candidates (the union is
        # world x eval_budget points; global row = rank x eval_budget + local row) and the winner's POINT comes out of the
        # exchange, so that every rank returns the same (xopt, fopt) whatever its local sample was.
        rank, world = engine_rank_world(crit.model.engine)
        if h is not None or g is not None:
            if world > 1:
                raise NotImplementedError("a constrained sweep runs on one rank (a shard without feasible rows cannot join the exchange)")
            Xs = Xs[feasible_rows(Xs, h, g)]
            if len(Xs) == 0:
                return [], []  # what the reference returns when no restart ends feasible (optim/__init__.py:148-149)
        full = Xs
        if masks is not None:  # ask(fixed=...): the free columns are swept, the fixed ones are filled in
            full = np.empty((len(Xs), len(masks)))
            full[:, ~masks] = Xs
            full[:, masks] = np.asarray(values, dtype=float)
        if optimizer == "sweep":
            best, _, xb = sweep_argmax([crit], full, index_offset=rank * int(eval_budget), return_points=True)
            x = np.asarray(xb[0], dtype=float)
            return (x[~masks] if masks is not None else x).tolist(), float(best[0])
        # hybrid: the sweep picks the n_restart most promising candidates, which are then polished
        if masks is not None or h is not None or g is not None:
            raise NotImplementedError("optimizer='sweep-BFGS' takes an unconstrained bogp criterion without fixed variables")
        k = int(max(1, min(n_restart, 32, len(Xs))))
        tv, _, tx = sweep_topk([crit], Xs, k, index_offset=rank * int(eval_budget))
        ok = np.isfinite(tv[0])
        if not ok.any():
            return [], []
        xp, fp = polish_topk(crit, tx[0][ok], np.array(search_space.bounds, dtype=float))
        j = int(np.argmax(fp))
        return xp[j].tolist(), float(fp[j])

    xopt, fopt = [], []
    best = -np.inf
    wait_count = 0
    bounds = np.array(search_space.bounds)

    direct = hasattr(obj_func, "acq_id")  # our criterion object itself; otherwise the reference's one-point wrapper

    def neg(x):  # Penalized without constraints (optim/__init__.py:45-52)
        x = np.asarray(x, dtype=float)
        f, fg = obj_func(x.reshape(1, -1), return_dx=True) if direct else obj_func(x)
        return -1.0 * float(np.asarray(f, float).ravel()[0]), -1.0 * np.asarray(fg, float).ravel()

    for iteration in range(n_restart):
        x0 = np.asarray(search_space.sample(N=1, method="uniform")[0], dtype=float)
        xopt_, fopt_, stop_dict = fmin_l_bfgs_b(neg, x0, pgtol=1e-8, factr=1e6, bounds=bounds, maxfun=eval_budget)
        xopt_ = xopt_.flatten().tolist()
        fopt_ = -float(fopt_)
        if fopt_ > best:
            best = fopt_
            wait_count = 0
        else:
            wait_count += 1
        eval_budget -= stop_dict["funcalls"]
        xopt.append(xopt_)
        fopt.append(fopt_)
        if eval_budget <= 0 or wait_count >= wait_iter:
            break
    if len(xopt) == 0:
        return [], []
    idx = np.argsort(fopt)[::-1]
    return xopt[idx[0]], fopt[idx[0]]


def polish_topk(crit, starts: np.ndarray, bounds: np.ndarray, max_iter: int = 50):
    """Local refinement of `starts` (k, d) inside the box `bounds` (d, 2); returns (points (k, d), values (k,)) with
    values[i] >= the criterion at starts[i].  On the device engine all k starts advance in lock step (`bogp_polish`:
    one batched value + gradient evaluation per iteration, the optimiser state never leaves the GPU) -- the reference
    runs its restarts one after the other, one point per call (optim/__init__.py:74-153).  An engine without `polish`
    (the test stand-ins) or a model the fused call does not serve (polynomial trend basis) gets the reference-style
    sequential L-BFGS-B through the one-point call."""
    starts = np.atleast_2d(np.asarray(starts, dtype=float))
    bounds = np.asarray(bounds, dtype=float)
    model = crit.model
    eng = getattr(model, "engine", None)
    fused = getattr(model, "_fused_point_ok", None)
    if eng is not None and hasattr(eng, "polish") and fused is not None and fused():  # (r05: any d up to BOGP_MAX_DIM; r03-r04 stopped at 64)
        if getattr(model, "_committed_par", None) is None:
            raise Exception("The model is not fitted yet!")
        xs, fs, _ = eng.polish(model._check_X(starts), bounds[:, 0], bounds[:, 1], (crit.acq_id, crit.acq_par()), crit.effective_plugin(),
                               crit.minimize, max_evals=int(max_iter))  # fmt: skip
        return xs, fs
    xs, fs = [], []
    for x0 in starts:
        def neg(x):
            f, fg = crit(np.asarray(x, dtype=float).reshape(1, -1), return_dx=True)
            return -1.0 * float(np.asarray(f, float).ravel()[0]), -1.0 * np.asarray(fg, float).ravel()

        f0 = -neg(x0)[0]
        x1, f1, _ = fmin_l_bfgs_b(neg, x0, pgtol=1e-8, factr=1e6, bounds=bounds, maxfun=max_iter)
        if -float(f1) >= f0:
            xs.append(np.asarray(x1, dtype=float).ravel()), fs.append(-float(f1))
        else:
            xs.append(x0), fs.append(f0)
    return np.array(xs), np.array(fs)
