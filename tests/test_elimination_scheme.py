"""The in-place elimination of DESIGN.md section 5.12 restated in NumPy -- executable documentation of what k_nll_small (4 x 4 blocks in
registers, one thread a block) and k_elim_step (64 x 64 blocks in global memory, one workgroup a block) do, checked against
numpy.linalg on the CPU.  No device, no library call: the GPU tests compare the kernels with the oracle; this file pins the SCHEME.

State: one b x b block T(bi, bj) per pair bi >= bj of block rows 0 .. nb (block row nb carries the right-hand sides [y; 1]).
T(bi, bj) is a block of R until block column bj is eliminated (step bj), then X(bj, bi)^T with X = L^-T until step bi, then block
(bi, bj) of -R^-1.  A step k: every block row i publishes a panel block P[i] = M_i L_kk^-T with M_i = the finished block of column k
(i > k), the TRANSPOSE of the finished block of row k (i < k), or the identity (i = k); then EVERY block does the same update,
T(bi, bj) <- (bi == k or bj == k ? 0 : T(bi, bj)) - P[bi] P[bj]^T.  gpr.py:795-808 (L, Yt, Ft) and :996-997 (R^-1, gamma) are
what falls out."""
import numpy as np
import pytest


def eliminate(R, rhs, b, pipelined=False):
    """R: (N, N) SPD, rhs: (m, N) rows to solve against (m <= b).  Returns L blocks' by-products and the final state:
    Rinv (N, N), X = rhs R^-1 (m, N), Z = rhs L^-T (m, N) (the forward-substituted rows), logdet = sum(log diag L)."""
    N = R.shape[0]
    nb = -(-N // b)
    Np = nb * b
    A = np.eye(Np)
    A[:N, :N] = R  # identity padding
    B = np.zeros((b, Np))
    B[: rhs.shape[0], :N] = rhs
    T = {}
    for bi in range(nb):
        for bj in range(bi + 1):
            T[bi, bj] = A[bi * b : (bi + 1) * b, bj * b : (bj + 1) * b].copy()
    for bj in range(nb):
        T[nb, bj] = B[:, bj * b : (bj + 1) * b].copy()
    Z = np.zeros((b, Np))
    logdet = 0.0

    def raw_of(state, k):
        """the blocks step k's panel is built from: column k as they are, row k transposed, the identity for block row k"""
        raw = {}
        for i in range(nb + 1):
            if i > k:
                raw[i] = state[i, k].copy()
            elif i < k:
                raw[i] = state[k, i].T.copy()
        return raw

    def panel_of(raw, D, k):
        L = np.linalg.cholesky(D)
        Linv_T = np.linalg.inv(L).T
        P = {i: m @ Linv_T for i, m in raw.items()}
        P[k] = Linv_T  # I L^-T
        return P, L

    if not pipelined:
        for k in range(nb):
            P, L = panel_of(raw_of(T, k), T[k, k], k)
            logdet += np.log(np.diag(L)).sum()
            Z[:, k * b : (k + 1) * b] = P[nb]
            for (bi, bj), t in T.items():
                base = 0.0 if (bi == k or bj == k) else t
                T[bi, bj] = base - P[bi] @ P[bj].T
    else:
        # software pipeline (k_nll_small): the blocks of column / row p + 1 are copied out in their state after update p - 1; the panel
        # builder applies update p to its copies itself, M_i -= P_p[i] P_p[p + 1]^T, while the owners apply update p to their own blocks
        raw = raw_of(T, 0)
        D = T[0, 0].copy()
        P, L = panel_of(raw, D, 0)
        for (bi, bj) in T:  # restart of column 0 (its first X-phase update is step 0 itself) ...
            if bj == 0:
                T[bi, bj] = np.zeros_like(T[bi, bj])
        raw_next = raw_of(T, 1) if nb > 1 else None  # ... BEFORE column / row 1 are copied out: block (1, 0) goes out as the zero it now is
        D_next = T[1, 1].copy() if nb > 1 else None
        for p in range(nb):
            logdet += np.log(np.diag(L)).sum()
            Z[:, p * b : (p + 1) * b] = P[nb]
            # the panel builder, one step ahead
            if p + 1 < nb:
                kn = p + 1
                for i in raw_next:
                    raw_next[i] = raw_next[i] - P[i] @ P[kn].T
                D_upd = D_next - P[kn] @ P[kn].T
                P_new, L_new = panel_of(raw_next, D_upd, kn)
            # the owners: update p, then restart column / row p + 1, then copy out column / row p + 2
            for (bi, bj), t in T.items():
                T[bi, bj] = t - P[bi] @ P[bj].T
            if p + 1 < nb:
                for (bi, bj) in T:
                    if bj == p + 1 or bi == p + 1:
                        T[bi, bj] = np.zeros_like(T[bi, bj])
                if p + 2 < nb:
                    raw_next = raw_of(T, p + 2)
                    D_next = T[p + 2, p + 2].copy()
                P, L = P_new, L_new
    Rinv = np.zeros((Np, Np))
    for bi in range(nb):
        for bj in range(bi + 1):
            blk = -T[bi, bj]
            Rinv[bi * b : (bi + 1) * b, bj * b : (bj + 1) * b] = blk
            Rinv[bj * b : (bj + 1) * b, bi * b : (bi + 1) * b] = blk.T
    X = np.hstack([-T[nb, bj] for bj in range(nb)])
    m = rhs.shape[0]
    return Rinv[:N, :N], X[:m, :N], Z[:m, :N], logdet


def spd(N, seed):
    rng = np.random.default_rng(seed)
    Xp = rng.uniform(-2, 2, size=(N, 3))
    D2 = ((Xp[:, None, :] - Xp[None, :, :]) ** 2).sum(-1)
    return np.exp(-0.7 * D2) + 1e-3 * np.eye(N), rng.standard_normal(N)


@pytest.mark.parametrize("pipelined", [False, True])
@pytest.mark.parametrize("N,b", [(1, 4), (3, 4), (4, 4), (5, 4), (16, 4), (37, 4), (64, 4), (64, 64), (65, 64), (130, 64), (200, 64), (50, 8)])
def test_in_place_elimination_gives_inverse_solves_and_logdet(N, b, pipelined):
    R, y = spd(N, 1000 * N + b)
    rhs = np.vstack([y, np.ones(N)])
    Rinv, X, Z, logdet = eliminate(R, rhs, b, pipelined)
    L = np.linalg.cholesky(R)
    scale = np.abs(np.linalg.inv(R)).max()
    assert np.abs(Rinv - np.linalg.inv(R)).max() <= 1e-9 * scale
    assert np.abs(X - np.linalg.solve(R, rhs.T).T).max() <= 1e-9 * max(1.0, np.abs(X).max())      # rows: (R^-1 y)^T, (R^-1 1)^T
    assert np.abs(Z - np.linalg.solve(L, rhs.T).T).max() <= 1e-10 * max(1.0, np.abs(Z).max())      # rows: Yt^T, Ft^T (gpr.py:799, 803)
    assert logdet == pytest.approx(np.log(np.diag(L)).sum(), rel=1e-12, abs=1e-12)


def test_likelihood_scalars_from_the_by_products():
    """sum(log diag L), |Ft|, Ft.Yt, rho.rho and gamma -- the five things the kernels hand to the host formulas (gpr.py:931-977, 996)."""
    N = 45
    R, y = spd(N, 7)
    rhs = np.vstack([y, np.ones(N)])
    Rinv, X, Z, logdet = eliminate(R, rhs, 4, pipelined=True)
    Yt, Ft = Z[0], Z[1]
    beta = (Ft @ Yt) / (Ft @ Ft)
    rho = Yt - Ft * beta
    gamma = X[0] - beta * X[1]
    L = np.linalg.cholesky(R)
    yt = np.linalg.solve(L, y)
    ft = np.linalg.solve(L, np.ones(N))
    b_ref = (ft @ yt) / (ft @ ft)
    assert beta == pytest.approx(b_ref, rel=1e-10)
    assert rho @ rho == pytest.approx(((yt - ft * b_ref) ** 2).sum(), rel=1e-9)
    assert np.abs(gamma - np.linalg.solve(R, y - b_ref)).max() <= 1e-8 * np.abs(gamma).max()
