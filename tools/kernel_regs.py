#!/usr/bin/env python3
"""Register / spill / scratch table of the kernels of one source file (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_regs.py noise_robust_vit_amd/csrc/nrv_gemm.hip [name-filter] [extra hipcc flags...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "noise_robust_vit_amd", "csrc"),
       "-I" + os.path.join(ROOT, "include"), "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    if filt and filt not in name:
        continue
    print(f"{name[:100]:100s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>3s} SGPR {r.get('TotalSGPRs', r.get('SGPRs','?')):>4s} "
          f"spill v {r.get('VGPRs Spill','?'):>4s} s {r.get('SGPRs Spill','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>5s}")
