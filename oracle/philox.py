"""NumPy restatement of libbogp's on-device candidate generator (TEST INFRASTRUCTURE, see gp_oracle.py header).

Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11): multipliers
0xD2511F53 / 0xCD9E8D57, key increments 0x9E3779B9 / 0xBB67AE85, 10 rounds.  Pinned against the paper's
known-answer vectors in tests/test_oracle_golden.py.  Element E = row * d + k of the candidate array uses counter
(E >> 1, 0, 0, 0), key (seed_lo, seed_hi), words (0,1) for even E and (2,3) for odd E;
u = ((a >> 5) * 2**26 + (b >> 6)) * 2**-53, x = lo + (hi - lo) * u  -- the arithmetic of csrc/kernels_acq.hip.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0 = np.asarray(k0, dtype=np.uint32)
    k1 = np.asarray(k1, dtype=np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = (p1 & MASK).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = (p0 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return c0, c1, c2, c3


def uniform_box(lo, hi, M, seed, first_row=0):
    """The M x d block [first_row, first_row + M) of the candidate stream `seed` in the box [lo, hi]."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    d = len(lo)
    E = np.arange(first_row * d, (first_row + M) * d, dtype=np.uint64)
    P = E >> np.uint64(1)
    z = np.zeros(len(E), dtype=np.uint32)
    w = philox4x32_10((P & MASK).astype(np.uint32), (P >> np.uint64(32)).astype(np.uint32), z, z,
                      np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))  # fmt: skip
    odd = (E & np.uint64(1)).astype(bool)
    a = np.where(odd, w[2], w[0]).astype(np.uint64)
    b = np.where(odd, w[3], w[1]).astype(np.uint64)
    u = ((a >> np.uint64(5)).astype(np.float64) * 67108864.0 + (b >> np.uint64(6)).astype(np.float64)) * (1.0 / 9007199254740992.0)
    k = (E % np.uint64(d)).astype(np.int64)
    return (lo[k] + (hi[k] - lo[k]) * u).reshape(M, d)


def _unit_words(E, seed):
    """u in [0, 1) for the global element indices E (uint64 array) of stream `seed`: the words `uniform_box` uses."""
    P = E >> np.uint64(1)
    z = np.zeros(len(E), dtype=np.uint32)
    w = philox4x32_10((P & MASK).astype(np.uint32), (P >> np.uint64(32)).astype(np.uint32), z, z,
                      np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))  # fmt: skip
    odd = (E & np.uint64(1)).astype(bool)
    a = np.where(odd, w[2], w[0]).astype(np.uint64)
    b = np.where(odd, w[3], w[1]).astype(np.uint64)
    return ((a >> np.uint64(5)).astype(np.float64) * 67108864.0 + (b >> np.uint64(6)).astype(np.float64)) * (1.0 / 9007199254740992.0)


def fmix32(h):
    """MurmurHash3's 32-bit finaliser (Appleby, public domain)."""
    h = np.asarray(h, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x85EBCA6B)
        h ^= h >> np.uint32(13)
        h *= np.uint32(0xC2B2AE35)
        h ^= h >> np.uint32(16)
    return h


def lhs_permutation(i, k, seed, n):
    """pi_k(i): the keyed permutation of [0, n) libbogp's Latin hypercube uses for dimension k (arrays i, k of equal
    length).  6-round balanced Feistel network on 2*hb bits + cycle walking; round keys = the 8 words of
    Philox(counter (k, 0x4C4853, 0|1, 0), key seed), the first six used."""
    i = np.asarray(i, dtype=np.uint64).copy()
    k = np.asarray(k, dtype=np.uint32)
    if n <= 1:
        return np.zeros_like(i)
    bits = int(n - 1).bit_length()
    hb = 1 if bits < 2 else (bits + 1) // 2
    mask = np.uint32((1 << hb) - 1)
    z = np.zeros(len(i), dtype=np.uint32)
    tag = np.full(len(i), 0x4C4853, dtype=np.uint32)
    k0, k1 = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    rk = list(philox4x32_10(k, tag, z, z, k0, k1)) + list(philox4x32_10(k, tag, z + np.uint32(1), z, k0, k1))
    todo = np.ones(len(i), dtype=bool)
    while todo.any():
        v = i[todo]
        L = (v >> np.uint64(hb)).astype(np.uint32)
        R = v.astype(np.uint32) & mask
        for r in range(6):
            F = fmix32(R ^ rk[r][todo]) & mask
            L, R = R, L ^ F
        v = (L.astype(np.uint64) << np.uint64(hb)) | R.astype(np.uint64)
        i[todo] = v
        todo[todo] = v >= np.uint64(n)
    return i


def lhs_box(lo, hi, M, seed, first_row=0, n_strata=None):
    """Rows [first_row, first_row + M) of the n_strata-point Latin hypercube of stream `seed` in the box [lo, hi]
    (csrc/kernels_acq.hip k_generate_lhs; follows pyDOE's classic design: one jittered point per stratum and
    dimension, strata shuffled per dimension)."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    d = len(lo)
    n = int(first_row + M if n_strata is None else n_strata)
    E = np.arange(first_row * d, (first_row + M) * d, dtype=np.uint64)
    k = (E % np.uint64(d)).astype(np.int64)
    pi = lhs_permutation(E // np.uint64(d), k.astype(np.uint32), seed, n)
    t = (pi.astype(np.float64) + _unit_words(E, seed)) / float(n)
    return (lo[k] + (hi[k] - lo[k]) * t).reshape(M, d)


def min_pdist(X):
    """scipy.spatial.distance.pdist(X).min() with the arithmetic spelled out: per pair the squared differences are added
    one dimension after the other (no FMA), the minimum taken over the squares and rooted once (sqrt is monotone)."""
    X = np.asarray(X, dtype=np.float64)
    n, d = X.shape
    best = np.inf
    for i in range(1, n):
        s = np.zeros(i)
        for k in range(d):
            diff = X[i, k] - X[:i, k]
            s = s + diff * diff
        best = min(best, float(s.min()))
    return float(np.sqrt(best))


def lhs_maximin_box(lo, hi, M, seed, iterations=5):
    """pyDOE's _lhsmaximin on libbogp's hypercubes (bogp_candidates_generate_lhs_maximin): trial t = stream seed +
    0x9E3779B97F4A7C15 t in the unit cube, keep the first trial with the largest minimum pairwise distance, return that
    design in the box [lo, hi] with (distance, trial)."""
    d = len(lo)
    best, best_t = -1.0, 0
    for t in range(iterations):
        s_t = (seed + 0x9E3779B97F4A7C15 * t) & 0xFFFFFFFFFFFFFFFF
        dist = min_pdist(lhs_box(np.zeros(d), np.ones(d), M, s_t))
        if best < dist:
            best, best_t = dist, t
    s_b = (seed + 0x9E3779B97F4A7C15 * best_t) & 0xFFFFFFFFFFFFFFFF
    return lhs_box(lo, hi, M, s_b), best, best_t


def sobol_box(lo, hi, M, sv, first_index=1):
    """Points [first_index, first_index + M) of the unscrambled Sobol' sequence with direction numbers sv (d, bits) in
    the box [lo, hi] (k_generate_sobol): XOR of sv[:, b] over the set bits b of the index's Gray code."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    sv = np.asarray(sv, dtype=np.uint64)
    d, bits = sv.shape
    n = np.arange(first_index, first_index + M, dtype=np.uint64)
    g = n ^ (n >> np.uint64(1))
    v = np.zeros((M, d), dtype=np.uint64)
    for b in range(bits):
        sel = ((g >> np.uint64(b)) & np.uint64(1)).astype(bool)
        v[sel] ^= sv[:, b]
    return lo + (hi - lo) * (v.astype(np.float64) * 2.0**-bits)


def transform(X, scales=None, precisions=None, lo=None, hi=None):
    """RealSpace.round(RealSpace.to_linear_scale(X)) (search_space.py:754-770) for a design X drawn in the transformed box:
    per column the inverse of the variable's scale (variable.py:40-55), then np.round to its precision and the clip to its
    bounds (variable.py:250-257).  NumPy's own exp / power: the device's libm may differ from them in the last place."""
    X = np.array(X, dtype=np.float64)
    d = X.shape[1]
    inv = {None: lambda v: v, "linear": lambda v: v, "log": np.exp, "log10": lambda v: np.power(10, v),
           "logit": lambda v: 1 / (1 + np.exp(-v)), "bilog": lambda v: np.sign(v) * (np.exp(np.abs(v)) - 1)}  # fmt: skip
    for k in range(d):
        X[:, k] = inv[scales[k] if scales is not None else None](X[:, k])
        p = precisions[k] if precisions is not None else None
        if p is not None:
            X[:, k] = np.clip(np.round(X[:, k], p), lo[k], hi[k])
    return X
