"""Stand-in for the third-party `pyDOE` package (absent from this image).

Signature-compatible `lhs(n, samples, criterion=...)`: plain (non-maximin) Latin hypercube driven
by the global `np.random` state.  Used only by oracle/make_golden.py to make the reference importable.
"""
import numpy as np


def lhs(n, samples=None, criterion=None, iterations=None):
    samples = samples or n
    cut = np.linspace(0, 1, samples + 1)
    a, b = cut[:samples], cut[1:]
    u = np.random.rand(samples, n)
    H = np.zeros_like(u)
    for j in range(n):
        H[:, j] = (u[:, j] * (b - a) + a)[np.random.permutation(samples)]
    return H
