#!/usr/bin/env python3
"""Kernel sequence of one training step from a rocprofv3 --kernel-trace CSV, run-length compressed, with the small PyTorch
kernels (fill / copy / elementwise) marked: where in the step do they come from?   python tools/trace_sequence.py <csv>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
idx = [i for i, e in enumerate(ev) if "adamw_kernel" in e[2]]
bounds = [i for j, i in enumerate(idx) if j + 1 == len(idx) or idx[j + 1] - i > 50]
a, b = bounds[-2], bounds[-1]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    m = re.search(r"at::native::(\w+)<.*?(FillFunctor|CUDAFunctor_add|MulFunctor|copy_kernel|\w+Functor|\w+_kernel_cuda)?", n)
    if n.startswith("at::native") or "at::native" in n[:40]:
        inner = re.findall(r"at::native::([A-Za-z_0-9]+)", n)
        return "TORCH:" + "/".join(inner[:3])
    return n.split("(")[0][:60]
seq = [(short(e[2]), (e[1] - e[0]) / 1e3) for e in ev[a + 1:b + 1]]
out, i = [], 0
while i < len(seq):
    j = i
    while j < len(seq) and seq[j][0] == seq[i][0]:
        j += 1
    out.append((seq[i][0], j - i, sum(s[1] for s in seq[i:j])))
    i = j
for name, n, us in out:
    print(f"{n:4d} x {name:70s} {us:9.1f} us")
