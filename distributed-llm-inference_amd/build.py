"""In-tree native build (no JIT cache, no site-packages install).

Two shared objects are produced next to the sources so that they travel with the tree:

* ``runtime/_runtime.so`` – C++ host runtime (block manager, batch builder); g++ + pybind11.
* ``ops/_C.so``           – HIP kernels for gfx950 + their ``torch.ops.dli`` bindings; hipcc.

``python -m dli_amd.build`` builds both; ``__graft_entry__.build()`` calls :func:`build_all`.
Each translation unit is rebuilt only when its source (or a header) is newer than its object.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ_DIR = os.path.join(PKG, "csrc", "build")
RUNTIME_SO = os.path.join(PKG, "runtime", "_runtime.so")
KERNELS_SO = os.path.join(PKG, "ops", "_C.so")
ARCH = "gfx950"


def _newer(src_list, target) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def _run(cmd):
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("build step failed:\n  " + " ".join(cmd) + "\n" + proc.stdout)
    return proc.stdout


def _headers():
    out = []
    for root, _d, files in os.walk(CSRC):
        out += [os.path.join(root, f) for f in files if f.endswith((".h", ".hpp", ".cuh"))]
    return out


def build_runtime(verbose: bool = False) -> str:
    import pybind11

    src = os.path.join(CSRC, "runtime", "block_manager.cpp")
    if _newer([src, os.path.join(CSRC, "runtime", "block_manager.h")], RUNTIME_SO):
        cxx = os.environ.get("CXX", "g++")
        # several processes may get here at once on a fresh checkout (TP workers, router replicas import the module
        # concurrently): each links into its own temp file and renames it into place — nobody can import a half-written .so
        tmp = f"{RUNTIME_SO}.{os.getpid()}.tmp"
        cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
               "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"], src, "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        try:
            _run(cmd)
            os.replace(tmp, RUNTIME_SO)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return RUNTIME_SO


def _torch_flags():
    import torch
    from torch.utils.cpp_extension import include_paths

    tdir = os.path.dirname(torch.__file__)
    inc = ["-I" + p for p in include_paths("cuda")]
    defs = ["-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2",
            "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    libdir = os.path.join(tdir, "lib")
    libs = ["-L" + libdir, "-Wl,-rpath," + libdir, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch"]
    return inc, defs, libs


def build_kernels(verbose: bool = False) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    kdir = os.path.join(CSRC, "kernels")
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdrs = _headers()
    kernel_srcs = sorted(os.path.join(kdir, f) for f in os.listdir(kdir) if f.endswith(".hip"))
    binding_src = os.path.join(CSRC, "torch_bindings.cpp")
    inc, defs, libs = _torch_flags()
    common = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-I" + CSRC, "-Wno-unused-result"]

    jobs = []
    objs = []
    for src in kernel_srcs:
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        objs.append(obj)
        if _newer([src] + hdrs, obj):
            jobs.append([hipcc] + common + ["-c", src, "-o", obj])
    bobj = os.path.join(OBJ_DIR, "torch_bindings.o")
    objs.append(bobj)
    if _newer([binding_src] + hdrs, bobj):
        jobs.append([hipcc] + common + inc + defs + ["-x", "hip", "-c", binding_src, "-o", bobj])

    def one(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        return _run(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), int(os.environ.get("MAX_JOBS", "8")))) as pool:
            list(pool.map(one, jobs))
    if jobs or _newer(objs, KERNELS_SO):
        tmp = f"{KERNELS_SO}.{os.getpid()}.tmp"
        try:
            one([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH] + objs + libs + ["-o", tmp])
            os.replace(tmp, KERNELS_SO)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return KERNELS_SO


def build_all(verbose: bool = False):
    return build_runtime(verbose), build_kernels(verbose)


if __name__ == "__main__":
    for path in build_all(verbose="-q" not in sys.argv):
        print("built", path)
