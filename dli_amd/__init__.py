"""Importable alias for the ``distributed-llm-inference_amd/`` source tree.

The framework directory carries the project's (hyphenated) name, which Python cannot import
directly.  This shim is the package object: its ``__path__`` points at that directory, so
``import dli_amd.engine`` resolves to ``distributed-llm-inference_amd/engine``.
"""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "distributed-llm-inference_amd")
__path__ = [_REAL]
with open(_os.path.join(_REAL, "__init__.py")) as _fh:
    exec(compile(_fh.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
del _fh
