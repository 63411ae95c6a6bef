"""Build libnrv_hip.so (the C-ABI HIP library of include/nrv.h) in-tree for gfx950.

    python -m noise_robust_vit_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects go to noise_robust_vit_amd/csrc/_obj/, the shared library to
noise_robust_vit_amd/lib/libnrv_hip.so (git-ignored, shipped to the GPU box with the tree snapshot).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libnrv_hip.so")
SOURCES = ["nrv_gemm.hip", "nrv_norm.hip", "nrv_attn.hip", "nrv_attn_gen.hip", "nrv_misc.hip", "nrv_sinkhorn.hip", "nrv_sinknorm.hip", "nrv_optim.hip", "nrv_bgemm.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP toolchain is required to build libnrv_hip.so")


def _deps(csrc: str = CSRC) -> list:
    hdrs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]
    hdrs.append(os.path.join(ROOT, "include", "nrv.h"))
    return hdrs


def _stale(target: str, srcs: list) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def _compile(src: str, force: bool, objdir: str = OBJ, extra: tuple = (), csrc: str = CSRC) -> str:
    obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
    path = os.path.join(csrc, src)
    if force or _stale(obj, [path] + _deps(csrc)):
        flags = [f if f != "-I" + CSRC else "-I" + csrc for f in FLAGS]
        cmd = [_hipcc()] + list(extra) + flags + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force: bool = False, verbose: bool = False, lib: str = LIB, objdir: str = OBJ, extra_flags: tuple = (),
          csrc: str = CSRC, sources=None) -> str:
    """The product library by default.  `lib` / `objdir` / `extra_flags` exist for tools/build_dev.py, which compiles
    instrumented or experimental variants of the same sources into tools/_build/ (never loaded by the package)."""
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    want = list(sources or SOURCES)
    srcs = [s for s in want if os.path.exists(os.path.join(csrc, s))]
    if len(srcs) != len(want):
        raise RuntimeError(f"missing HIP sources: {sorted(set(want) - set(srcs))}")
    with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, objdir, extra_flags, csrc), srcs))
    if force or _stale(lib, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {lib} ({os.path.getsize(lib)} bytes)")
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
