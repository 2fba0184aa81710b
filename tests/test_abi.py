"""The C-ABI library loads and exports every symbol include/bogp.h declares (no compute calls: CPU-only test)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

from bogp import _lib

HEADER = os.path.join(ROOT, "include", "bogp.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bogp_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    assert len(syms) >= 15
    for must in ("bogp_create", "bogp_set_train", "bogp_nll", "bogp_commit", "bogp_predict", "bogp_sweep", "bogp_gradient"):
        assert must in syms


def test_library_is_built_and_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "libbogp.so missing: run `python __graft_entry__.py`"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), "libbogp.so does not export %s" % name


def test_ctypes_table_covers_the_header_exactly():
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_abi_version_and_constants_match_header():
    lib = _lib.load()
    src = open(HEADER).read()
    consts = dict(re.findall(r"#define\s+(BOGP_[A-Z_0-9]+)\s+\(?(-?\d+)\)?", src))
    assert lib.bogp_abi_version() == int(consts["BOGP_ABI_VERSION"]) == _lib.ABI_VERSION == 9
    assert int(consts["BOGP_KERNEL_MATERN52"]) == _lib.KERNEL_MATERN52 == 3
    assert int(consts["BOGP_MODE_NOISE_ESTIM"]) == _lib.MODE_NOISE_ESTIM == 2
    assert int(consts["BOGP_ACQ_MGFI"]) == _lib.ACQ_MGFI == 3
    assert int(consts["BOGP_ERR_NOT_POSDEF"]) == _lib.ERR_NOT_POSDEF == -3
    assert int(consts["BOGP_ERR_LLF_POSITIVE"]) == _lib.ERR_LLF_POSITIVE == -6
    assert int(consts["BOGP_MAX_Q"]) == _lib.MAX_Q
    assert int(consts["BOGP_TREND_QUADRATIC"]) == _lib.TREND_QUADRATIC == 2
    assert lib.bogp_trend_size(0, 7) == 1 and lib.bogp_trend_size(1, 7) == 8 and lib.bogp_trend_size(2, 7) == 36


def test_likelihood_path_by_size():
    """bogp_nll_path: which device path a likelihood evaluation takes is decided from the sizes alone (no handle, no device call):
    one launch of one workgroup up to N = 156 while X and the block image fit one CU's LDS, one launch per 64 columns up to
    N = 3072 (2048 until the end of r05), the multi-kernel path above, for polynomial bases and for several targets."""
    lib = _lib.load()
    consts = dict(re.findall(r"#define\s+(BOGP_[A-Z_0-9]+)\s+\(?(-?\d+)\)?", open(HEADER).read()))
    general, one, elim = (int(consts["BOGP_NLL_PATH_" + k]) for k in ("GENERAL", "ONE_LAUNCH", "ELIM"))
    p = lib.bogp_nll_path
    assert [p(n, 10, 0, 1) for n in (1, 16, 128, 129, 156)] == [one] * 5
    assert [p(n, 10, 0, 1) for n in (157, 192, 256, 1024, 2048, 2049, 3072)] == [elim] * 7
    assert p(3073, 10, 0, 1) == general and p(8192, 50, 0, 1) == general
    assert p(100, 64, 0, 1) == one and p(100, 65, 0, 1) == general      # theta travels as a kernel argument: d <= 64; ld = 128 < 192
    assert p(150, 40, 0, 1) == elim                                       # X + the block image exceed 160 KB of LDS: next path
    assert p(100, 5, 1, 1) == general and p(100, 5, 2, 1) == general      # linear / quadratic basis
    assert p(100, 5, 0, 2) == general                                     # several targets
    assert p(0, 5, 0, 1) == general and p(10, 0, 0, 1) == general
    os.environ["BOGP_NLL_FUSED"] = "0"
    try:
        assert p(64, 5, 0, 1) == general and p(300, 5, 0, 1) == general
    finally:
        del os.environ["BOGP_NLL_FUSED"]
    os.environ["BOGP_NLL_ELIM"] = "0"
    try:
        assert p(64, 5, 0, 1) == one and p(300, 5, 0, 1) == general
    finally:
        del os.environ["BOGP_NLL_ELIM"]


def test_wide_panel_schedule_of_the_large_cholesky_by_size():
    """bogp_chol_wide_panels: the first block columns of the general path's Cholesky (gpr.py:795) run as wide panels from ld = 6144 (N = 6017) on -- 24 block
    columns a panel while 72 stay behind them -- decided from the size and BOGP_BIG_CHOL alone (no handle, no device call); a list in the switch is
    taken as given as far as launch_chol_lower's own conditions allow (even column count, an unfused chain behind the last panel)."""
    import ctypes as C

    lib = _lib.load()

    def sched(N):
        w = (C.c_int * 16)()
        n = lib.bogp_chol_wide_panels(N, w, 16)
        return [int(w[i]) for i in range(n)]

    assert [sched(N) for N in (0, 100, 3073, 4096, 6016)] == [[]] * 5   # fewer than 96 block columns: no wide panels
    assert sched(6017) == sched(6144) == sched(7000) == sched(7552) == [24]  # (the leading dimension is N rounded up to 128 above N = 3072)
    assert sched(7553) == sched(8192) == sched(9088) == [24, 24]
    assert sched(10240) == [24, 24, 24] and sched(16384) == [24] * 7
    assert lib.bogp_chol_wide_panels(8192, None, 0) == 2                # count only
    for value, expect in (("0", []), ("32", [32]), ("16,10,6", [16, 10, 6]), ("15,16", []), ("16,15", [16]), ("64,40", [64])):
        os.environ["BOGP_BIG_CHOL"] = value
        try:
            assert sched(8192) == expect, value
        finally:
            del os.environ["BOGP_BIG_CHOL"]
    os.environ["BOGP_NO_BIG_FIT"] = "1"
    try:
        assert sched(8192) == []
    finally:
        del os.environ["BOGP_NO_BIG_FIT"]


def test_oracle_ids_match_library_ids():
    from oracle import gp_oracle as O

    assert (O.KERNEL_SE, O.KERNEL_MATERN12, O.KERNEL_MATERN32, O.KERNEL_MATERN52, O.KERNEL_ABSEXP) == (0, 1, 2, 3, 4)
    assert _lib.KERNEL_ABSEXP == 4
    assert (O.MODE_NOISELESS, O.MODE_NOISY, O.MODE_NOISE_ESTIM) == (_lib.MODE_NOISELESS, _lib.MODE_NOISY, _lib.MODE_NOISE_ESTIM)
    assert (O.ACQ_EI, O.ACQ_EPSILON_PI, O.ACQ_UCB, O.ACQ_MGFI) == (_lib.ACQ_EI, _lib.ACQ_EPSILON_PI, _lib.ACQ_UCB, _lib.ACQ_MGFI)
    assert (O.TREND_CONSTANT, O.TREND_LINEAR, O.TREND_QUADRATIC) == (_lib.TREND_CONSTANT, _lib.TREND_LINEAR, _lib.TREND_QUADRATIC)


def _has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.mark.skipif(_has_gpu(), reason="checks the loud failure on a box WITHOUT a GPU")
def test_no_gpu_means_loud_failure_not_fallback():
    with pytest.raises(_lib.BogpError) as e:
        _lib.Engine(0)
    assert e.value.code == _lib.ERR_NO_DEVICE


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "bayesian-optimization_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "gp_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f


def test_graft_entry_build_is_consistent_with_the_abi():
    """The driver's build check: `make` (a no-op when the library is current) + the ABI version assertion."""
    import __graft_entry__ as entry

    entry.build()


def _build_c_client(tmp_path):
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    exe = str(tmp_path / "abi_smoke")
    pkg = os.path.join(ROOT, "bayesian-optimization_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe, "-L", pkg, "-lbogp", "-Wl,-rpath," + pkg]  # fmt: skip
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    """include/bogp.h compiles as C99 (-pedantic -Werror) and a C program links against libbogp.so with no Python in
    the process.  Without a GPU the client must stop at bogp_create with BOGP_ERR_NO_DEVICE (exit code 3), loudly."""
    import subprocess

    exe = _build_c_client(tmp_path)
    res = subprocess.run([exe, "50", "3", "100", "7"], capture_output=True, text=True, timeout=120)
    if res.returncode == 0:  # a GPU is present: the numbers are checked by the -m gpu test
        assert "best" in res.stdout
    else:
        assert res.returncode == 3 and "bogp_create" in res.stderr, (res.returncode, res.stderr)


def test_c_reduce_rules_equal_the_python_ones():
    """bogp_reduce_pairs / bogp_merge_topk (the library's side of the exchange, host code: no GPU needed) against the
    Python rules of bogp.distributed on adversarial records: cross-rank ties, NaNs, -inf padding, empty slots."""
    import numpy as np

    from bogp import distributed

    rng = np.random.default_rng(0)
    for trial in range(200):
        R, q, d = int(rng.integers(1, 9)), int(rng.integers(1, 6)), int(rng.integers(0, 4))
        vals = rng.choice([0.0, 1.0, 2.5, -1.0, np.nan, np.inf, -np.inf], size=(R, q))  # many exact ties across ranks
        idxs = rng.permutation(R * q * 10)[: R * q].reshape(R, q).astype(np.int64)
        xs = rng.standard_normal((R, q, d))
        rec = np.empty((R, q, 2 + d))
        rec[..., 0], rec[..., 1], rec[..., 2:] = vals, idxs.view(np.float64), xs
        v, i, x = _lib.reduce_pairs_c(rec)
        win = distributed.reduce_pairs(vals, idxs)
        ar = np.arange(q)
        np.testing.assert_array_equal(v, vals[win, ar])
        np.testing.assert_array_equal(i, idxs[win, ar])
        if d:
            np.testing.assert_array_equal(x, xs[win, ar])
        # it is np.argmax over the concatenation ordered by global index
        for c in range(q):
            order = np.argsort(idxs[:, c])
            assert i[c] == idxs[order, c][int(np.argmax(vals[order, c]))]
    for trial in range(100):
        R, q, k, d = int(rng.integers(1, 6)), int(rng.integers(1, 4)), int(rng.integers(1, 7)), int(rng.integers(0, 3))
        vals = rng.choice([0.0, 1.0, 2.5, -1.0, np.nan, 7.0], size=(R, q, k))
        idxs = rng.permutation(R * q * k * 4)[: R * q * k].reshape(R, q, k).astype(np.int64)
        empty = rng.random((R, q, k)) < 0.2
        idxs[empty], vals[empty] = -1, -np.inf
        xs = rng.standard_normal((R, q, k, d))
        rec = np.empty((R, q, k, 2 + d))
        rec[..., 0], rec[..., 1], rec[..., 2:] = vals, idxs.view(np.float64), xs
        v, i, x = _lib.merge_topk_c(rec)
        for c in range(q):
            pv, pi, pr, ps = distributed.merge_topk(vals[:, c, :], idxs[:, c, :], k)
            np.testing.assert_array_equal(v[c], pv)
            np.testing.assert_array_equal(i[c], pi)
            for j in range(k):
                if d and pr[j] >= 0:
                    np.testing.assert_array_equal(x[c, j], xs[pr[j], c, ps[j]])
                elif d:
                    assert np.all(np.isnan(x[c, j]))


def test_switch_table_is_complete():
    """every BOGP_* variable the native library reads is in the table of tools/README.md ("Switches") with its default and the
    test that runs its non-default setting -- at most 20 of them -- and that test's source does mention the variable."""
    import glob
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    read = set()
    for f in glob.glob(os.path.join(root, "bayesian-optimization_amd", "csrc", "*.h*")):
        read |= set(re.findall(r'getenv\("(BOGP_[A-Z0-9_]+)"\)', open(f).read()))
    table = {}
    for line in open(os.path.join(root, "tools", "README.md")):
        m = re.match(r"\| `(BOGP_[A-Z0-9_]+)` \|", line)
        if m:
            table[m.group(1)] = re.findall(r"`(tests/[a-z_0-9]+\.py)", line)
    assert read == set(table), "library reads %s, table lists %s" % (sorted(read - set(table)), sorted(set(table) - read))
    assert len(read) <= 20, sorted(read)
    for var, files in table.items():
        assert files, var
        assert any(var in open(os.path.join(root, f)).read() for f in files), "%s: none of %s mentions it" % (var, files)
