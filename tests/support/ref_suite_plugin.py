"""pytest plugin (build container only): runs the REFERENCE's own unittest files, with or without `bogp.install()`.

    cd <scratch> && BOGP_REF_SUITE_INSTALL=1 PYTHONPATH=/root/repo:/root/repo/tests:/root/reference:/root/repo/oracle/shims \
        python -m pytest -p support.ref_suite_plugin /root/reference/unittest/test_BO.py ... -p no:cacheprovider

Before collection it (a) papers over one scikit-learn API change the reference predates (`OneHotEncoder(sparse=...)` became
`sparse_output=` in 1.2 and the old keyword was removed in 1.4: without this the RandomForest tests of the reference cannot
run in this image at all, install or not) and (b) with BOGP_REF_SUITE_INSTALL=1 installs the binding -- so that the test
modules' `from bayes_optim.surrogate import GaussianProcess` resolves to the dispatching class -- with the oracle-backed
engine stand-in under `bogp.GaussianProcess` (no GPU here).
tests/test_install_dropin.py::test_reference_suite_passes_under_install drives both runs and compares them test by test."""
import os
import warnings


def pytest_configure(config):
    warnings.filterwarnings("ignore")
    import sklearn.preprocessing as skp

    _orig = skp.OneHotEncoder

    def OneHotEncoder(*a, sparse=None, **kw):  # noqa: N802 -- same name: the reference imports it by name
        if sparse is not None:
            kw.setdefault("sparse_output", sparse)
        return _orig(*a, **kw)

    skp.OneHotEncoder = OneHotEncoder
    import bayes_optim.surrogate.random_forest as rf

    if hasattr(rf, "OneHotEncoder"):
        rf.OneHotEncoder = OneHotEncoder
    if os.environ.get("BOGP_REF_SUITE_INSTALL", "0") == "1":
        import bayes_optim

        import bogp
        from support.oracle_engine import OracleEngine

        bogp._lib.Engine = lambda device=0: OracleEngine(device)
        bogp.install(bayes_optim)
