"""Stand-in for the third-party `py_expression_eval` package (absent from this image).

Only used by oracle/make_golden.py so that `import bayes_optim` succeeds in the build
container; it is NOT reference code and never runs on the GPU box.  Only conditional
search spaces need a working parser, and those are outside the hot path.
"""


class Parser:
    def parse(self, s):
        raise NotImplementedError("py_expression_eval stand-in: expression parsing unavailable")
