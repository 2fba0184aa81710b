"""bogp -- MI355X-native GP-surrogate + batch-acquisition engine behind the `bayes_optim` protocols.

The directory is named `bayesian-optimization_amd/`; import it as `bogp` (bogp/__init__.py is the alias).

  bogp.GaussianProcess           <-> bayes_optim.surrogate.GaussianProcess      (fit / predict / gradient)
  bogp.acquisition.{EI,PI,EpsilonPI,UCB,MGFI} <-> bayes_optim.acquisition.acquisition_fun.*
  bogp.optim.argmax_restart      <-> bayes_optim.acquisition.optim.argmax_restart (+ optimizer="sweep")
  bogp.trend                     <-> bayes_optim.surrogate.trend
  bogp.install(bayes_optim)      re-points the reference's three extension points + ParallelBO's q-criterion loop
  bogp._lib.Engine               ctypes binding of libbogp.so (include/bogp.h)

Every numerical step runs on the GPU through libbogp.so; importing works anywhere, but creating an engine without
a gfx950 device (or without the built library) raises -- there is no CPU fallback.
"""
__version__ = "0.1.0"

from . import _lib, acquisition, distributed, integration, optim  # noqa: E402,F401
from . import prior_mean as trend  # noqa: E402,F401
from .acquisition import EI, MGFI, PI, UCB, EpsilonPI  # noqa: E402,F401
from .optim import argmax_restart, batch_argmax, device_sample, sweep_argmax, sweep_generated, sweep_topk, sweep_topk_generated  # noqa: E402,F401
from .integration import install, uninstall  # noqa: E402,F401
from .surrogate import GaussianProcess  # noqa: E402,F401
