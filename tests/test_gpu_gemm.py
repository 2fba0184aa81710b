"""The in-tree dense product (csrc/kernels_gemm.hip: k_gemm64) that replaced every rocBLAS call of r01-r02 -- the
polynomial-trend algebra of gpr.py:799-808 (`Ft = solve_triangular(C, F)`, `linalg.qr(Ft)`, `dot(Q, dot(Q.T, Yt))`), the
REML trend terms (:850-918), the small-batch `solve_triangular(C, r.T)` of :494 and the trend part of `gradient()` -- against
NumPy on the shapes those call sites produce: odd sizes on every side of the 64 x 64 tile and the 32-deep k-block, all four
transpositions, alpha / beta, triangular operands (k range shortened per row tile), and the deterministic split-K path.
Needs a real MI355X: `pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from bogp import _lib  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    e = _lib.Engine(0)
    yield e
    e.close()


def ref(A, B, C, ta, tb, alpha, beta):
    opA = A.T if ta else A
    opB = B.T if tb else B
    return alpha * (opA @ opB) + (beta * C if C is not None else 0.0)


def check(got, want, A, B):
    scale = np.abs(A).max() * np.abs(B).max() * max(A.shape) + 1e-300
    assert np.max(np.abs(got - want)) <= 1e-13 * scale


# (m, n, k): the call sites' shapes -- N x p x N, p x p x N (Gram matrices), N x p x p, p x 1 x N and N x 1 x p (matrix-vector),
# 1 x 1 x N (a dot product), N x B x N (small batches) -- at sizes straddling 64 / 32 and far from them
SHAPES = [(1, 1, 1), (1, 1, 777), (3, 5, 2), (21, 21, 2048), (231, 231, 700), (64, 64, 32), (65, 63, 33), (128, 1, 31), (200, 21, 200),
          (513, 7, 129), (21, 1, 1500), (1500, 1, 21), (300, 32, 300), (2048, 21, 21), (700, 231, 231)]


@pytest.mark.parametrize("m,n,k", SHAPES)
@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True), (True, True)])
def test_general_product_matches_numpy(eng, m, n, k, ta, tb):
    rng = np.random.default_rng(m * 1000003 + n * 1009 + k + 7 * ta + 13 * tb)
    A = rng.standard_normal((k, m) if ta else (m, k))
    B = rng.standard_normal((n, k) if tb else (k, n))
    C0 = rng.standard_normal((m, n))
    for split in (False, True):
        got = eng.selftest_gemm(A, B, ta=ta, tb=tb, split=split)
        check(got, ref(A, B, None, ta, tb, 1.0, 0.0), A, B)
        got = eng.selftest_gemm(A, B, C_in=C0, ta=ta, tb=tb, alpha=-0.75, beta=1.5, split=split)
        check(got - 1.5 * C0, ref(A, B, None, ta, tb, -0.75, 0.0), A, B)


def test_padded_leading_dimensions(eng):
    """lda / ldb / ldc above the stored rows (the trend buffers are views into N- or Np-row arrays)."""
    rng = np.random.default_rng(3)
    big_a, big_b, big_c = rng.standard_normal((300, 90)), rng.standard_normal((120, 40)), rng.standard_normal((350, 40))
    A, B = big_a[:257, :90], big_b[:90, :33]
    out = np.asfortranarray(big_c.copy())
    lib = _lib.load()
    fa, fb = np.asfortranarray(big_a), np.asfortranarray(big_b)
    rc = lib.bogp_selftest_gemm(eng._h, 0, 0, 257, 33, 90, 1.0, _lib._ptr(fa), 300, _lib._ptr(fb), 120, 0.0, _lib._ptr(out), 350, 0, 0)
    assert rc == 0
    check(out[:257, :33], A @ B, A, B)
    np.testing.assert_array_equal(out[257:], big_c[257:])        # rows below m are not touched
    np.testing.assert_array_equal(out[:, 33:], big_c[:, 33:])    # nor columns right of n


@pytest.mark.parametrize("n_rows", [1, 63, 64, 65, 200, 1000])
@pytest.mark.parametrize("nrhs", [1, 5, 32, 70])
def test_triangular_operand(eng, n_rows, nrhs):
    """V r with V = L^-1 lower triangular and zeros stored above (the small-batch posterior, gpr.py:494), and V^T z (upper)."""
    rng = np.random.default_rng(n_rows * 131 + nrhs)
    V = np.tril(rng.standard_normal((n_rows, n_rows)))
    R = rng.standard_normal((n_rows, nrhs))
    for split in (False, True):
        check(eng.selftest_gemm(V, R, tri=1, split=split), V @ R, V, R)
        check(eng.selftest_gemm(V, R, ta=True, tri=2, split=split), V.T @ R, V, R)


def test_split_k_is_deterministic_and_equals_the_sliced_sum(eng):
    """The split path adds the slices' partial tiles in slice order whichever workgroup arrives last: identical bits on every
    run, and within rounding of the one-pass product."""
    rng = np.random.default_rng(11)
    A, B = rng.standard_normal((4096, 21)), rng.standard_normal((4096, 21))
    runs = [eng.selftest_gemm(A, B, ta=True, split=True) for _ in range(6)]
    for r in runs[1:]:
        np.testing.assert_array_equal(r, runs[0])
    one_pass = eng.selftest_gemm(A, B, ta=True, split=False)
    check(runs[0], one_pass, A, B)
    check(runs[0], A.T @ B, A, B)


def test_bad_arguments_are_refused(eng):
    lib = _lib.load()
    a = np.zeros((4, 4), order="F")
    assert lib.bogp_selftest_gemm(eng._h, 0, 0, 4, 4, 4, 1.0, _lib._ptr(a), 3, _lib._ptr(a), 4, 0.0, _lib._ptr(a), 4, 0, 0) == _lib.ERR_INVALID
    assert lib.bogp_selftest_gemm(eng._h, 0, 0, 4, 4, 4, 1.0, None, 4, _lib._ptr(a), 4, 0.0, _lib._ptr(a), 4, 0, 0) == _lib.ERR_INVALID
    assert lib.bogp_selftest_gemm(eng._h, 0, 0, 4, 2, 3, 1.0, _lib._ptr(a), 4, _lib._ptr(a), 4, 0.0, _lib._ptr(a), 4, 1, 0) == _lib.ERR_INVALID
