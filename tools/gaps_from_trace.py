#!/usr/bin/env python3
"""Device idle time between kernels from a rocprofv3 --kernel-trace CSV: is a step waiting for the host?
    python tools/gaps_from_trace.py <kernel_trace.csv> [steps-to-skip]"""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
# steps are delimited by the AdamW launch
idx = [i for i, e in enumerate(ev) if "adamw_kernel" in e[2]]
bounds = [i for j, i in enumerate(idx) if j + 1 == len(idx) or idx[j + 1] - i > 50]      # last AdamW range of each step
print(f"{len(ev)} dispatches, {len(bounds)} steps")
for a, b in list(zip(bounds, bounds[1:]))[-3:]:
    seg = ev[a + 1:b + 1]
    span = seg[-1][1] - seg[0][0]
    busy = sum(e[1] - e[0] for e in seg)
    gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
    big = sorted(((g, seg[i][2][:50], seg[i + 1][2][:50]) for i, g in enumerate(gaps) if g > 20000), reverse=True)[:8]
    print(f"step: {len(seg)} dispatches, span {span / 1e6:.3f} ms, kernels busy {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms "
          f"(median gap {statistics.median(gaps) / 1e3:.1f} us, gaps > 20 us: {sum(1 for g in gaps if g > 20000)})")
    for g, p, n in big:
        print(f"    {g / 1e3:8.1f} us between {p} -> {n}")
