"""bogp -- MI355X-native GP-surrogate + batch-acquisition engine behind the `bayes_optim` protocols.

Import as `bogp` (see bogp/__init__.py).  Modules:
  _lib         ctypes binding of libbogp.so (include/bogp.h); no CPU fallback
"""
__version__ = "0.1.0"
