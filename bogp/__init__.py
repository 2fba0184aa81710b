"""Import alias: the package directory is `bayesian-optimization_amd/` (the repository's naming contract), which is
not a valid Python identifier; `import bogp` exposes the same modules (`bogp.surrogate`, `bogp.acquisition`, ...)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "bayesian-optimization_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
