#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv: per kernel (last dispatch of each name), counter values."""
import csv, sys, collections, re
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
per = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"]
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = short[:70]
    key = (short, r["Dispatch_Id"])
    per.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
last = collections.OrderedDict()
for (short, did), c in per.items():
    last[short] = c           # keep the last dispatch of each kernel name
for k, c in last.items():
    if not any(t in k for t in ("gemm", "attn", "splitk", "ln_", "colsum")): continue
    print(k)
    print("   " + "  ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())))
