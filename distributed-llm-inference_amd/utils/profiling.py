"""Tracing hooks (SURVEY.md §5.1 [NEW-BUILD]).

* ``trace_range(name)`` – roctx range (``torch.cuda.nvtx`` is roctx on ROCm) around engine phases, visible in
  ``rocprofv3 --marker-trace`` timelines.  Enabled with ``DLI_PROFILE=1``; a no-op otherwise (zero cost in the hot loop).
* The counter recipes used for ``profiles/`` live in ``docs/PROFILING.md``.
"""
from __future__ import annotations

import contextlib
import os
import time

ENABLED = os.environ.get("DLI_PROFILE", "0") not in ("", "0")
_nvtx = None
if ENABLED:
    try:
        import torch
        if torch.cuda.is_available():
            _nvtx = torch.cuda.nvtx
    except Exception:
        _nvtx = None


@contextlib.contextmanager
def _range(name):
    _nvtx.range_push(name)
    try:
        yield
    finally:
        _nvtx.range_pop()


_NULL = contextlib.nullcontext()


def trace_range(name: str):
    return _range(name) if _nvtx is not None else _NULL
