"""Stand-in for the third-party `sobol_seq` package (absent from this image); see pyDOE.py."""
from scipy.stats import qmc


def i4_sobol_generate(dim, n, skip=1):
    return qmc.Sobol(d=dim, scramble=False).random(n + skip)[skip:]
