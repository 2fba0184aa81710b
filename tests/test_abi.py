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
    assert lib.bogp_abi_version() == 2
    src = open(HEADER).read()
    consts = dict(re.findall(r"#define\s+(BOGP_[A-Z_0-9]+)\s+\(?(-?\d+)\)?", src))
    assert int(consts["BOGP_KERNEL_MATERN52"]) == _lib.KERNEL_MATERN52 == 3
    assert int(consts["BOGP_MODE_NOISE_ESTIM"]) == _lib.MODE_NOISE_ESTIM == 2
    assert int(consts["BOGP_ACQ_MGFI"]) == _lib.ACQ_MGFI == 3
    assert int(consts["BOGP_ERR_NOT_POSDEF"]) == _lib.ERR_NOT_POSDEF == -3
    assert int(consts["BOGP_ERR_LLF_POSITIVE"]) == _lib.ERR_LLF_POSITIVE == -6
    assert int(consts["BOGP_MAX_Q"]) == _lib.MAX_Q
    assert int(consts["BOGP_TREND_QUADRATIC"]) == _lib.TREND_QUADRATIC == 2
    assert lib.bogp_trend_size(0, 7) == 1 and lib.bogp_trend_size(1, 7) == 8 and lib.bogp_trend_size(2, 7) == 36


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
