#!/usr/bin/env python3
"""What is the host doing during the largest device idle gaps of a step?  Inputs: rocprofv3 --kernel-trace --hip-trace CSVs.
    python tools/gap_api.py <kernel_trace.csv> <hip_api_trace.csv>"""
import csv, sys
k = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda t: t[0])
api = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in csv.DictReader(open(sys.argv[2]))]
api.sort()
idx = [i for i, e in enumerate(k) if "adamw_kernel" in e[2]]
bounds = [i for j, i in enumerate(idx) if j + 1 == len(idx) or idx[j + 1] - i > 50]
a, b = bounds[-2], bounds[-1]
seg = k[a + 1:b + 1]
gaps = sorted(((seg[i + 1][0] - seg[i][1], i) for i in range(len(seg) - 1)), reverse=True)[:3]
for g, i in gaps:
    t0, t1 = seg[i][1], seg[i + 1][0]
    print(f"gap {g / 1e3:.1f} us between {seg[i][2][:60]} -> {seg[i + 1][2][:60]}")
    inside = [(s, e, f) for s, e, f in api if e > t0 - 300000 and s < t1]
    # the launch call of the kernel that ends the gap: the last hipLaunchKernel / hipModuleLaunchKernel starting before t1
    for s, e, f in inside[-40:]:
        mark = "*" if e - s > 20000 else " "
        print(f"   {mark} {f:38s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us")
