"""Python face of the C++ block manager (``csrc/runtime/block_manager.cpp`` → ``_runtime.so``).

If the shared object is missing (fresh checkout on a CPU box) it is compiled on first import with
g++ — a ~10 s host-only build with no GPU dependency.
"""
from __future__ import annotations

import importlib
import os


def _load():
    try:
        return importlib.import_module("dli_amd.runtime._runtime")
    except ImportError:
        from dli_amd import build
        build.build_runtime()
        importlib.invalidate_caches()
        return importlib.import_module("dli_amd.runtime._runtime")


_rt = _load()
BlockManager = _rt.BlockManager
RUNTIME_SO = os.path.abspath(_rt.__file__)

__all__ = ["BlockManager", "RUNTIME_SO"]
