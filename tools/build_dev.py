#!/usr/bin/env python3
"""Build a developer variant of the HIP library into tools/_build/ (never loaded by the package or the tests).

    python tools/build_dev.py NAME                   ->  tools/_build/libnrv_hip_NAME.so from the working tree
    python tools/build_dev.py NAME --instrument      ->  the same with the stamp hooks of tools/dev/nrv_dev.hpp compiled in
    python tools/build_dev.py NAME --rev GITREV      ->  the sources of an earlier commit (A/B baseline)

Used for same-process A/B runs (tools/_devlib.use_library) and for the phase-stamp instrumentation of the GEMM / Sinkhorn
kernels (`--instrument` adds nrv_dev_stamps_enable_<tu> / nrv_dev_read_stamps_<tu>).  The product sources contain no
experiment switches: an A/B baseline is an earlier commit (`--rev`).  hipcc cross-compiles: build here, the .so travels to
the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noise_robust_vit_amd import build as B

if __name__ == "__main__":
    import subprocess
    name = sys.argv[1]
    args = sys.argv[2:]
    out = os.path.join(ROOT, "tools", "_build")
    csrc, sources = B.CSRC, None
    if "--rev" in args:
        rev = args[args.index("--rev") + 1]
        args = [a for a in args if a not in ("--rev", rev)]
        csrc = os.path.join(out, f"src_{name}")
        os.makedirs(csrc, exist_ok=True)
        listing = subprocess.run(["git", "ls-tree", "--name-only", rev, "noise_robust_vit_amd/csrc/"], cwd=ROOT,
                                 capture_output=True, text=True, check=True).stdout.split()
        sources = []
        for f in listing:
            base = os.path.basename(f)
            text = subprocess.run(["git", "show", f"{rev}:{f}"], cwd=ROOT, capture_output=True, text=True, check=True).stdout
            dst = os.path.join(csrc, base)
            if not os.path.exists(dst) or open(dst).read() != text:
                open(dst, "w").write(text)
            if base.endswith(".hip"):
                sources.append(base)
    instrument = "--instrument" in args
    args = [a for a in args if a != "--instrument"]
    flags = tuple(a for a in args if a.startswith("-"))
    if instrument:      # <nrv_dev.hpp> resolves to tools/dev/nrv_dev.hpp (real hooks) instead of csrc/nrv_dev.hpp (empty ones)
        flags = ("-I" + os.path.join(ROOT, "tools", "dev"),) + flags
    B.build(force=False, verbose=True, lib=os.path.join(out, f"libnrv_hip_{name}.so"),
            objdir=os.path.join(out, f"obj_{name}"), extra_flags=flags, csrc=csrc, sources=sources)
