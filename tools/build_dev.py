#!/usr/bin/env python3
"""Build a developer variant of the HIP library into tools/_build/ (never loaded by the package or the tests).

    python tools/build_dev.py NAME [-DNRV_DEV_STAMPS] [-DNRV_X=1 ...]   ->  tools/_build/libnrv_hip_NAME.so

Used for same-process A/B runs (tools/_devlib.use_library) and for the phase-stamp instrumentation of the NT GEMM
(-DNRV_DEV_STAMPS adds `nrv_dev_read_stamps`).  hipcc cross-compiles: build here, the .so travels to the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noise_robust_vit_amd import build as B

if __name__ == "__main__":
    name = sys.argv[1]
    flags = tuple(a for a in sys.argv[2:] if a.startswith("-"))
    out = os.path.join(ROOT, "tools", "_build")
    B.build(force=False, verbose=True, lib=os.path.join(out, f"libnrv_hip_{name}.so"),
            objdir=os.path.join(out, f"obj_{name}"), extra_flags=flags)
