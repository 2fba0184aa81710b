"""K_nu(x) as csrc/bogp_device.h computes it since r05, restated step for step in Python (Chebyshev expansions of gam1 / gam2, Temme's
series for x <= 1 with the powers from pow(), the trapezoidal rule on the integral representation above, upward recurrence in
double-double arithmetic; Dekker's two_prod stands in for the device's fma).  Held to the mpmath table tests/golden/G36_kv_table.npz by
tests/test_oracle_golden.py (<= 6 eps).  Test infrastructure: also the ACCURATE kv that tools/fuzz_parity.py swaps into the oracle for
ill-conditioned general-nu Matern problems, where scipy.special.kv's own error (up to hundreds of eps) would be amplified by cond(R)."""
import math

import numpy as np

G1 = [-0.5710113401855839203, 0.0065165112670736880645, 0.00030870901730853682431, -3.470626964904317836e-6, 6.9437664486674495957e-9,
      3.6779539885744101652e-11, -1.3563951023664248708e-13, -3.6802984806357979599e-17, 5.4582162333769858553e-19]  # fmt: skip
G2 = [0.92187029365045265648, -0.07685284084478667369, 0.0012719271366545622927, -4.9717367041957398581e-6, -3.3126119768180852711e-8,
      2.4230957900482704055e-10, -1.7023776642512729175e-13, -1.4943667065169001769e-15, 2.3826220476859635824e-18, 2.9017595056104745456e-21]  # fmt: skip

def cheb(c, t):
    b1 = b2 = 0.0
    for a_ in c[:0:-1]:
        b1, b2 = 2 * t * b1 + (a_ - b2), b1
    return t * b1 + (c[0] - b2)

def split(a_):  # Veltkamp
    c = 134217729.0 * a_
    hi = c - (c - a_)
    return hi, a_ - hi

def two_prod(a_, b_):  # Dekker: p + e == a * b exactly (what fma(a, b, -p) gives the device)
    p_ = a_ * b_
    ah, al = split(a_)
    bh, bl = split(b_)
    return p_, ((ah * bh - p_) + ah * bl + al * bh) + al * bl

def dd_sum(a_, b_):
    s_ = a_ + b_
    bb = s_ - a_
    return s_, (a_ - (s_ - bb)) + (b_ - bb)

def dd_mul(a_, b_):
    p_, e = two_prod(a_[0], b_[0])
    e += a_[0] * b_[1] + a_[1] * b_[0]
    s_ = p_ + e
    return s_, e - (s_ - p_)

def dd_add(a_, b_):
    s_, e = dd_sum(a_[0], b_[0])
    e += a_[1] + b_[1]
    t = s_ + e
    return t, e - (t - s_)

def dd_div_d(a_, x):
    q1 = a_[0] / x
    p_, e = two_prod(q1, x)
    r = ((a_[0] - p_) - e) + a_[1]
    q2 = r / x
    s_ = q1 + q2
    return s_, q2 - (s_ - q1)

def knu(nu, x):
    nl = int(nu + 0.5)
    mu = nu - nl
    mu2 = mu * mu
    tc = 8.0 * mu2 - 1.0
    gam1, gam2 = cheb(G1, tc), cheb(G2, tc)
    gampl, gammi = gam2 - mu * gam1, gam2 + mu * gam1
    if x <= 1.0:
        b = 0.5 * x
        dd = -math.log(b)
        e = mu * dd
        pw = math.pow(b, -mu)
        pwi = 1.0 / pw
        if abs(e) < 0.5:
            fact2 = 1.0 if abs(e) < 1e-8 else math.sinh(e) / e
            ch = math.cosh(e)
        else:
            fact2, ch = 0.5 * (pw - pwi) / e, 0.5 * (pw + pwi)
        pimu = math.pi * mu
        fact = 1.0 if abs(pimu) < 1e-8 else pimu / math.sin(pimu)
        ff = fact * (gam1 * ch + gam2 * fact2 * dd)
        s_ = ff
        p_, q_, c, dd = 0.5 * pw / gampl, 0.5 * pwi / gammi, 1.0, b * b
        s1 = p_
        for i in range(1, 61):
            ff = (i * ff + p_ + q_) / (i * i - mu2)
            c *= dd / i
            p_ /= i - mu
            q_ /= i + mu
            de = c * ff
            s_ += de
            s1 += c * (p_ - i * ff)
            if abs(de) < abs(s_) * 1e-17:
                break
        kmu, kmu1 = s_, s1 * (2.0 / x)
    else:
        h = min(0.2, 0.55 / math.sqrt(x))
        s0 = s1 = 0.5
        for k in range(1, 201):
            t = k * h
            sh = math.sinh(0.5 * t)
            w = math.exp(-2.0 * x * sh * sh)
            wb = w * math.cosh((mu + 1.0) * t)
            s0 += w * math.cosh(mu * t)
            s1 += wb
            if wb < 1e-18 * s1:
                break
        ex = math.exp(-x) * h
        kmu, kmu1 = ex * s0, ex * s1
    k0, k1 = (kmu, 0.0), (kmu1, 0.0)
    for i in range(1, nl + 1):
        c = dd_sum(float(i), mu)
        c = (2.0 * c[0], 2.0 * c[1])
        k0, k1 = k1, dd_add(dd_mul(dd_div_d(c, x), k1), k0)
    return k0[0] + k0[1]


def kv_accurate(nu, x):
    """Drop-in for scipy.special.kv(nu, x) with a scalar order and an array argument."""
    x = np.asarray(x, dtype=np.float64)
    f = np.frompyfunc(lambda v: knu(float(nu), float(v)) if v > 0 else np.inf, 1, 1)
    return f(x).astype(np.float64)
