"""Differential fuzz: the device engine against the oracle engine (tests/support/oracle_engine.py = oracle/gp_oracle.py
behind the same methods) on random problems for a given number of seconds -- sizes around every block boundary of the
kernels (N mod 16 / 32 / 64, the 512 limit of the fused sweep, M mod 64), every kernel / mode / trend the path builds,
random criteria.  Prints one line per failure with the seed that reproduces it; exits non-zero if any.
--trend (r03, after the library GEMMs were replaced by kernels_gemm.hip): every problem has a linear or quadratic basis with up
to 91 columns, N from 3 p to 1500 -- the shapes k_gemm64's bounds checks, triangular k ranges and split K have to get right --
plus the restricted likelihood with its three trend products; candidates up to 70 000 rows so that the k_mm128 chunk products
run on whole and on ragged chunks.
usage: python tools/fuzz_parity.py [--wide | --trend] [seconds] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bogp import _lib  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402
from support.oracle_engine import OracleEngine  # noqa: E402

KERNELS = [O.KERNEL_SE, O.KERNEL_MATERN12, O.KERNEL_MATERN32, O.KERNEL_MATERN52, O.KERNEL_ABSEXP, O.KERNEL_CUBIC, O.KERNEL_GENEXP, O.KERNEL_MATERN_NU]
NO_GRAD = (O.KERNEL_CUBIC, O.KERNEL_GENEXP, O.KERNEL_MATERN_NU)
EDGE_N = [2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 640, 1023, 1025]


# Every place a comparison is made at something wider than the bare 1e-6 rtol is COUNTED (README, "Tolerance ledger"): the summary line at the end
# says how many problems took each carve-out, so that a creeping tolerance shows up as a count going up between rounds.
CARVE = {}


def carve(name, n=1):
    CARVE[name] = CARVE.get(name, 0) + n


def close(a, b, rtol=1e-6, atol=0.0):
    return np.allclose(np.asarray(a, float), np.asarray(b, float), rtol=rtol, atol=atol, equal_nan=True)


TREND = False  # --trend: polynomial bases with many columns (see the module docstring)
WIDE = False  # --wide: d up to 60 (the fused sweep's limit), N up to 2200, + the restricted likelihood and input gradients


def one(seed, eng, orc):
    O.KV_OVERRIDE = None  # (a previous ill-conditioned general-nu problem may have set it, see below)
    rng = np.random.default_rng(seed)
    d = int(rng.integers(1, 13))
    N = int(rng.choice(EDGE_N)) if rng.random() < 0.6 else int(rng.integers(2, 700))
    if WIDE:
        d = int(rng.choice([1, 2, 3, 5, 10, 16, 17, 20, 33, 48, 59, 60]))
        if rng.random() < 0.25:
            N = int(rng.choice([1536, 2047, 2048, 2049, 2200]))
    N = max(N, 2)
    kernel = int(rng.choice(KERNELS))
    mode = int(rng.integers(0, 3))
    trend = int(rng.choice([0, 0, 0, 1, 2])) if d <= 4 and N > 40 else 0
    if TREND:
        d = int(rng.choice([1, 2, 3, 5, 8, 12]))
        trend = int(rng.choice([1, 2]))
        pcols = d + 1 if trend == 1 else (d + 1) * (d + 2) // 2
        N = int(rng.choice([3 * pcols, 3 * pcols + 1, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 1500])) if rng.random() < 0.7 else int(rng.integers(3 * pcols, 1500))
        N = max(N, 3 * pcols)
        kernel = int(rng.choice([O.KERNEL_SE, O.KERNEL_MATERN32, O.KERNEL_MATERN52, O.KERNEL_ABSEXP]))
    est = bool(rng.integers(0, 2))
    X = rng.uniform(-5, 5, (N, d))
    y = np.sum(np.sin(X), axis=1) + 0.3 * np.sum(X**2, axis=1) / d
    y = ((y - y.mean()) / (y.std() + 1e-12) + 0.05 * rng.standard_normal(N)).reshape(-1, 1)
    theta = np.exp(rng.uniform(np.log(0.02), np.log(0.5), d)) / d
    if kernel == O.KERNEL_GENEXP:
        theta = np.r_[theta, rng.uniform(1.0, 2.0)]
    if kernel == O.KERNEL_MATERN_NU:  # the order as the last theta entry (general-nu arm, kernel.py:201-207)
        theta = np.r_[theta, rng.choice([0.3, 0.8, 1.0, 2.0, 3.7, 4.5]) if rng.random() < 0.5 else rng.uniform(0.2, 5.0)]
    nv = 0.0
    if mode == O.MODE_NOISELESS:
        par = theta
        if kernel in (O.KERNEL_SE, O.KERNEL_MATERN52, O.KERNEL_CUBIC) and N > 64:  # smooth kernels without a nugget: singular
            mode, par, nv = O.MODE_NOISY, np.r_[theta, 0.9], 1e-4
    elif mode == O.MODE_NOISY:
        par, nv = np.r_[theta, rng.uniform(0.5, 1.5)], 10.0 ** rng.uniform(-6, -2)
    else:
        par = np.r_[theta, rng.uniform(0.7, 0.999)]
    p = 1 if trend == 0 else (d + 1 if trend == 1 else (d + 1) * (d + 2) // 2)
    beta = 0.0 if est else (0.1 if trend == 0 else np.linspace(-0.1, 0.1, p))
    tag = "seed %d: N=%d d=%d kernel=%d mode=%d trend=%d est=%d" % (seed, N, d, kernel, mode, trend, est)
    # both sides lose cond(R) eps of accuracy: compare only where that is below the tolerances (the estimate: squared ratio of
    # the extreme diagonal entries of the Cholesky factor, a lower bound of cond(R)); the rest still runs, for crashes
    Rm = O.correlation_matrix(kernel, np.asarray(theta), X)
    if mode == O.MODE_NOISY:
        Rm = (par[-1] * Rm + nv * np.eye(N)) / (par[-1] + nv)
    elif mode == O.MODE_NOISE_ESTIM:
        Rm = par[-1] * Rm + (1 - par[-1]) * np.eye(N)
    try:
        dl = np.diag(np.linalg.cholesky(Rm))
        cond = float((dl.max() / dl.min()) ** 2)
        if TREND or (d == 1 and cond > 1e8):  # (d == 1: seed 400300, the same family in the standard mode) the 2-norm condition number itself: the diagonal ratio is only a lower bound (seed 70229: a 1-D exponential
            w = np.linalg.eigvalsh(Rm)  # kernel on 1500 points, 5e7 by the ratio and 1.1e11 by the spectrum -- both builds,
            cond = max(cond, float(w[-1] / max(w[0], 1e-300)))  # with and without rocBLAS, sat 2.4e-5 from the oracle's gradient)
    except np.linalg.LinAlgError:
        cond = np.inf
    # r03: up to cond 1e12 the comparison is MADE, at the tolerance both sides can honour -- each carries
    # cond(R) eps of error (the LAPACK solve of the oracle included), so the posterior is compared at max(1e-6, 100 cond eps);
    # beyond 1e12 the problem still runs, for crashes
    compare = cond <= 1e12
    tol = max(1e-8, 100.0 * cond * 2.2e-16)  # likelihood tolerance: both sides carry cond(R) eps
    ptol = max(1e-6, 100.0 * cond * 2.2e-16)  # posterior / criterion tolerance
    if compare and ptol > 1e-6:
        carve("T5 posterior / criteria at 100 cond(R) eps instead of 1e-6 (cond > 4.5e7)")
    if compare and tol > 1e-8:
        carve("T6 likelihood at 100 cond(R) eps instead of 1e-8 (cond > 4.5e5)")
    if kernel == O.KERNEL_MATERN_NU and compare and cond > 1e4:
        carve("T7 general-nu oracle given the accurate K_nu (cond > 1e4)")
        # r05: the device's K_nu is within 4 eps of the TRUE value (tests/test_gpu_special.py), but scipy.special.kv -- the oracle's, the reference's --
        # is up to hundreds of eps off (tests/golden/G36_kv_table.npz): the reference's R is a perturbed matrix and cond(R) amplifies the
        # perturbation (seed 960921: cond 8e9, the two means 1e-4 apart).  On ill-conditioned problems of this kernel the oracle therefore gets an
        # ACCURATE kv (tests/support/bessel.py: the device's algorithm restated in Python, held to the mpmath table at 6 eps) and the comparison is
        # made at the generic tolerance; everything but the Bessel function -- distances, factorisation, solves, criteria -- is still the oracle's.
        from support.bessel import kv_accurate

        O.KV_OVERRIDE = kv_accurate
    eng.set_train(X, y)
    orc.set_train(X, y)
    fails = []
    grad = kernel not in NO_GRAD
    try:
        ref = orc.nll(kernel, mode, par, nv, est, beta, eval_grad=grad, trend=trend)
        ref_err = None
    except Exception as e:  # noqa: BLE001
        ref, ref_err = None, type(e).__name__
    try:
        got = eng.nll(kernel, mode, par, nv, est, beta, eval_grad=grad, trend=trend)
        got_err = None
    except _lib.BogpError as e:
        got, got_err = None, type(e).__name__
    if (ref_err is None) != (got_err is None):
        # a likelihood at the edge of positive definiteness may fail on one side only; count, do not fail
        carve("T8 likelihood failed on one side only (edge of positive definiteness): counted, not compared")
        return [], tag + " -- one-sided failure (%s / %s)" % (ref_err, got_err)
    if ref is None:
        return [], None
    if not compare:
        try:
            eng.commit(kernel, mode, par, nv, est, beta, trend=trend)
            eng.upload_candidates(rng.uniform(-5, 5, (100, d)))
            eng.sweep([(0, 0.0)], float(y.min()), True)
        except _lib.BogpError:
            pass
        carve("T9 cond(R) > 1e12: ran for crashes, not compared")
        return [], tag + " -- ill-conditioned (cond %.1e > 1e12): ran, not compared" % cond
    if grad:
        if not close(got[0], ref[0], tol, tol):
            fails.append("llf %r vs %r" % (got[0], ref[0]))
        gtol = 1e-5 * max(1.0, ptol / 1e-6)  # the gradient's traces go through R^-1: cond(R) eps on both sides, like the posterior
        if not close(np.ravel(got[1]), np.ravel(ref[1]), gtol, 0.1 * gtol * (1 + np.abs(ref[1]).max())):
            fails.append("grad max diff %g (|grad| %.3g, cond %.1e)" % (np.abs(np.ravel(got[1]) - np.ravel(ref[1])).max(), np.abs(ref[1]).max(), cond))
    elif not close(got, ref, tol, tol):
        fails.append("llf %r vs %r" % (got, ref))
    # r04: the batched likelihood must return the sequential call's BITS, slot by slot (bogp_nll_batch), whatever the conditioning
    if True:
        P = int(rng.integers(2, 7))
        pars = np.tile(par, (P, 1)) * 10.0 ** rng.uniform(-0.15, 0.15, size=(P, len(par)))
        if mode == 2:
            pars[:, -1] = np.clip(pars[:, -1], 1e-6, 1 - 1e-9)
        pars[0] = par
        try:
            bl, bg, bi = eng.nll_batch(kernel, mode, pars, nv, est, beta, eval_grad=grad, trend=trend)
        except _lib.BogpError as e:
            fails.append("nll_batch raised %s" % e)
            bl = None
        if bl is not None:
            for s_ in range(P):
                try:
                    one = eng.nll(kernel, mode, pars[s_], nv, est, beta, eval_grad=grad, trend=trend)
                    l1, g1 = (one[0], np.ravel(one[1])) if grad else (one, None)
                except _lib.NotPositiveDefinite:
                    l1, g1 = -np.inf, (np.zeros(len(par)) if grad else None)
                except _lib.BogpError:
                    continue
                if not (bl[s_] == l1 or (np.isnan(bl[s_]) and np.isnan(l1))) or (grad and not np.array_equal(bg[s_], g1)):
                    fails.append("nll_batch slot %d of %d differs from the sequential call: %r vs %r" % (s_, P, bl[s_], l1))
                    break
    try:
        orc.commit(kernel, mode, par, nv, est, beta, trend=trend)
        eng.commit(kernel, mode, par, nv, est, beta, trend=trend)
    except Exception as e:  # noqa: BLE001
        return fails, tag + " -- commit failed (%s)" % type(e).__name__
    M = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 100, 1000, 4097])) if rng.random() < 0.7 else int(rng.integers(1, 6000))
    if TREND and rng.random() < 0.15:
        M = int(rng.choice([16384, 20000, 70000]))
    Xs = rng.uniform(-5, 5, (M, d))
    if M > 3:
        Xs[1] = X[0]  # a training point among the candidates
    eng.upload_candidates(Xs)
    orc.upload_candidates(Xs)
    mu, mse = eng.predict()
    rmu, rmse = orc.predict()
    s2 = float(np.atleast_1d(orc.get_state(False)["sigma2"])[0]) if hasattr(orc, "get_state") else 1.0
    mscale = max(1.0, float(np.max(np.abs(rmu))))
    if not close(mu, rmu, ptol, 1e-7 * ptol / 1e-6 * mscale):
        fails.append("mu max diff %g (cond %.1e)" % (np.abs(np.ravel(mu) - np.ravel(rmu)).max(), cond))
    if not close(mse, rmse, ptol, 1e-7 * ptol / 1e-6 * s2):
        fails.append("mse max diff %g (sigma2 %g, cond %.1e)" % (np.abs(np.ravel(mse) - np.ravel(rmse)).max(), s2, cond))
    if TREND and trend == 1 and kernel in (O.KERNEL_SE, O.KERNEL_MATERN32, O.KERNEL_ABSEXP):  # linear basis: the r01 gradient route (k_gemm64 against V, V^T, W^T)
        x = rng.uniform(-5, 5, d)
        try:
            gm, gv = eng.gradient(x)
            rm, rv_ = orc.gradient(x)
            if not close(gm, rm, ptol, 1e-8 * ptol / 1e-6) or not close(gv, rv_, ptol, 1e-8 * ptol / 1e-6 * s2):
                fails.append("input gradient (linear basis) max diff %g / %g" % (np.abs(np.ravel(gm) - np.ravel(rm)).max(), np.abs(np.ravel(gv) - np.ravel(rv_)).max()))
        except (_lib.BogpError, NotImplementedError) as e:
            fails.append("gradient raised %s" % type(e).__name__)
        eng.upload_candidates(Xs)
        orc.upload_candidates(Xs)
    if TREND and len(theta) == d:  # the restricted likelihood with the basis' three trend products
        rpar = np.r_[theta, 0.8] if mode != O.MODE_NOISE_ESTIM else np.r_[theta, 0.8, 1e-3]
        rnv = nv if mode == O.MODE_NOISY else 0.0
        try:
            rr = orc.nll_restricted(kernel, mode, rpar, rnv, est, beta, eval_grad=True, trend=trend)
        except Exception:  # noqa: BLE001
            rr = None
        try:
            gr = eng.nll_restricted(kernel, mode, rpar, rnv, est, beta, eval_grad=True, trend=trend)
        except _lib.BogpError:
            gr = None
        if rr is not None and gr is not None and np.isfinite(rr[0]):
            if not close(gr[0], rr[0], tol, tol):
                fails.append("REML %r vs %r" % (gr[0], rr[0]))
            if not close(np.ravel(gr[1]), np.ravel(rr[1]), 1e-5 * max(1.0, ptol / 1e-6), 1e-6 * max(1.0, ptol / 1e-6) * (1 + np.abs(rr[1]).max())):
                fails.append("REML grad max diff %g" % np.abs(np.ravel(gr[1]) - np.ravel(rr[1])).max())
        eng.commit(kernel, mode, par, nv, est, beta, trend=trend)
        eng.upload_candidates(Xs)
    if WIDE and trend == 0 and kernel in (O.KERNEL_SE, O.KERNEL_MATERN32, O.KERNEL_ABSEXP):
        # input gradients of the posterior at a random point (gpr.py:537-576) ...
        x = rng.uniform(-5, 5, d)
        try:
            gm, gv = eng.gradient(x)
            rm, rv_ = orc.gradient(x)
            if not close(gm, rm, ptol, 1e-8 * ptol / 1e-6) or not close(gv, rv_, ptol, 1e-8 * ptol / 1e-6 * s2):
                fails.append("input gradient max diff %g / %g" % (np.abs(np.ravel(gm) - np.ravel(rm)).max(), np.abs(np.ravel(gv) - np.ravel(rv_)).max()))
        except (_lib.BogpError, NotImplementedError) as e:
            fails.append("gradient raised %s" % type(e).__name__)
        # ... the batched one-point path (r03: k_point_rhs / k_point_tri / k_point_finish) at B random rows: posterior, input
        # gradients, and the criteria's own values against the sweep kernel's
        if True:
            B = int(rng.choice([1, 2, 3, 17, 40]))
            Xb = rng.uniform(-5, 5, (B, d))
            pacq = [(int(rng.integers(0, 4)), float(rng.uniform(0.1, 3.0))) for _ in range(int(rng.integers(1, 4)))]
            try:
                pm, ps, pdm, pds, pv, pdv = eng.point_eval_batch(Xb, pacq, float(y.min()), True)
                eng.upload_candidates(Xb)
                orc.upload_candidates(Xb)
                rm2, rs2 = orc.predict()
                if not close(pm, rm2, ptol, 1e-7 * ptol / 1e-6 * mscale) or not close(ps, rs2, ptol, 1e-7 * ptol / 1e-6 * s2):
                    fails.append("point batch B=%d: mu / mse max diff %g / %g" % (B, np.abs(pm - np.ravel(rm2)).max(), np.abs(ps - np.ravel(rs2)).max()))
                for bb in range(min(B, 3)):
                    rm, rv_ = orc.gradient(Xb[bb])
                    if not close(pdm[bb], rm, ptol, 1e-8 * ptol / 1e-6) or not close(pds[bb], rv_, ptol, 1e-8 * ptol / 1e-6 * s2):
                        fails.append("point batch B=%d row %d: gradient max diff %g / %g" % (B, bb, np.abs(pdm[bb] - np.ravel(rm)).max(), np.abs(pds[bb] - np.ravel(rv_)).max()))
                _, _, sv = eng.sweep(pacq, float(y.min()), True, return_values=True)
                solid = np.ravel(rs2) > 1e-9 * s2
                if not np.all(solid):
                    carve("T3 criteria rows with MSE <= 1e-9 sigma2 (noise rows) left out of the point-batch / sweep comparison", int(np.sum(~solid)))
                if not close(pv[solid], sv.T[solid], 1e-6, 1e-300):
                    fails.append("point batch B=%d: criteria differ from the sweep's by %g" % (B, np.nanmax(np.abs(pv[solid] - sv.T[solid]))))
                if not np.all(np.isfinite(pdv[solid])):
                    fails.append("point batch B=%d: non-finite criterion gradient" % B)
            except (_lib.BogpError, NotImplementedError) as e:
                fails.append("point_eval_batch raised %s" % type(e).__name__)
            eng.upload_candidates(Xs)
            orc.upload_candidates(Xs)
        # ... and the restricted likelihood (gpr.py:813-918) at [theta, sigma2(, noise_var)]
        if len(theta) == d:
            rpar = np.r_[theta, 0.8] if mode != O.MODE_NOISE_ESTIM else np.r_[theta, 0.8, 1e-3]
            rnv = nv if mode == O.MODE_NOISY else 0.0
            try:
                rr = orc.nll_restricted(kernel, mode, rpar, rnv, est, beta, eval_grad=True, trend=trend)
            except Exception:  # noqa: BLE001
                rr = None
            try:
                gr = eng.nll_restricted(kernel, mode, rpar, rnv, est, beta, eval_grad=True, trend=trend)
            except _lib.BogpError:
                gr = None
            if rr is not None and gr is not None and np.isfinite(rr[0]):
                if not close(gr[0], rr[0], tol, tol):
                    fails.append("REML %r vs %r" % (gr[0], rr[0]))
                if not close(np.ravel(gr[1]), np.ravel(rr[1]), 1e-5, 1e-6 * (1 + np.abs(rr[1]).max())):
                    fails.append("REML grad max diff %g" % np.abs(np.ravel(gr[1]) - np.ravel(rr[1])).max())
            eng.commit(kernel, mode, par, nv, est, beta, trend=trend)  # (the likelihood call dropped the committed state)
            eng.upload_candidates(Xs)
    q = int(rng.integers(1, 5))
    acq = [(int(rng.integers(0, 4)), float(rng.uniform(0.1, 3.0))) for _ in range(q)]
    plugin = float(y.min())
    b, i = eng.sweep(acq, plugin, True)
    rb, ri, rv = orc.sweep(acq, plugin, True, return_values=True)
    for c in range(q):
        if int(i[c]) != int(ri[c]):
            # a different index is a failure only if the oracle's values separate the two candidates
            v = rv[c]
            noise_row = acq[c][0] in (1, 3) and min(np.ravel(rmse)[int(i[c])], np.ravel(rmse)[int(ri[c])]) <= 1e-9 * s2
            if np.isnan(v[int(i[c])]) or abs(v[int(i[c])] - v[int(ri[c])]) <= max(1e-9, 1e-3 * ptol) * (abs(v[int(ri[c])]) + 1e-300):
                carve("T10 argmax index differs where the oracle's two values are a tie (<= 1e-9 relative)")
            elif noise_row:
                # (r06, seed 3150676: N = 2, a candidate ON a training point: the oracle's MSE there is exactly 0.0, the device's 5.3e-16 -- EpsilonPI / MGFI
                # divide by sd, so one side's guard returns 0 and the other's formula 0.4.  The SAME rule as tests/test_gpu_parity.py: the argmax of
                # these two criteria is compared unless the winner on either side is a row whose reference MSE is rounding noise.)
                carve("T3 argmax of EpsilonPI / MGFI not compared: the winner on one side is a noise row (reference MSE <= 1e-9 sigma2)")
            else:
                fails.append("argmax[%d] %d vs %d (values %r / %r)" % (c, i[c], ri[c], v[int(i[c])], v[int(ri[c])]))
        elif ((abs(rb[c]) > 1e10 and abs(b[c]) > 1e10) or (0.0 < abs(rb[c]) < 1e-10 and 0.0 < abs(b[c]) < 1e-10)) and np.sign(b[c]) == np.sign(rb[c]):
            # MGFI far out on its exponential (exp of tens to hundreds) -- or, r05, EI / PI / MGFI far DOWN their Gaussian tail (seed 3060382: a best value of
            # 1.0467e-109, z ~ -22, 3e-5 from the oracle's on the builds before AND after the r05 fit work): a relative error e in the exponent is
            # e |exponent| in the value, so the exponents are what can be compared at the posterior's tolerance
            if abs(b[c]) < 2.3e-308 and abs(rb[c]) < 2.3e-308:
                # r06 (seeds 1150105, 1151224: 5e-324 vs 2.3e-321, 1.5e-314 vs 2.1e-311, the same index on both sides): SUBNORMAL on both sides.  A subnormal
                # double carries 52 - (1022 + log2 value) bits (5e-324: one), and EI / PI down there are differences of products that have themselves
                # underflowed: neither side's value means anything beyond "zero".  The index was compared above.
                carve("T11 best criterion value subnormal (< 2.2e-308) on both sides: index compared, value not")
                continue
            carve("T4 criterion value beyond 1e+-10: exponents compared")
            if not close(np.log(abs(b[c])), np.log(abs(rb[c])), ptol, 0.0):
                fails.append("best[%d] %r vs %r (exponents differ)" % (c, b[c], rb[c]))
        elif not close(b[c], rb[c], 10 * ptol if ptol > 1e-6 else 1e-6, 1e-300):
            fails.append("best[%d] %r vs %r" % (c, b[c], rb[c]))
    return [tag + ": " + f for f in fails], None


def main():
    global WIDE, TREND
    if "--wide" in sys.argv:
        WIDE = True
        sys.argv.remove("--wide")
    if "--trend" in sys.argv:
        TREND = True
        sys.argv.remove("--trend")
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    eng, orc = _lib.Engine(0), OracleEngine()
    t0 = time.time()
    n = nfail = nskip = 0
    while time.time() - t0 < seconds:
        fails, note = one(seed, eng, orc)
        for f in fails:
            print("FAIL", f, flush=True)
        if note:
            nskip += 1
            print("note", note, flush=True)
        nfail += bool(fails)
        n += 1
        seed += 1
    print("fuzz: %d problems in %.0f s, %d with failures, %d notes (next seed %d)" % (n, time.time() - t0, nfail, nskip, seed))
    print("carve-outs taken (problems, or rows where it says rows), of %d problems:" % n)
    for k in sorted(CARVE):
        print("   %6d  %s" % (CARVE[k], k))
    if not CARVE:
        print("   none")
    sys.exit(1 if nfail else 0)


if __name__ == "__main__":
    main()
