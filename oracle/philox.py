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
