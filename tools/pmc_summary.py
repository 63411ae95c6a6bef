#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv: per kernel name, the MEAN counter values over its dispatches (and how many)."""
import collections, csv, re, sys
rows = csv.DictReader(open(sys.argv[1]))
per = collections.OrderedDict()
for r in rows:
    short = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:90]
    d = per.setdefault(short, {})
    d.setdefault(r["Counter_Name"], {})[r["Dispatch_Id"]] = float(r["Counter_Value"])
for k, c in per.items():
    if not any(t in k for t in ("gemm", "attn", "splitk", "ln_", "colsum", "sinkhorn")):
        continue
    n = max(len(v) for v in c.values())
    print(f"{k}  [{n} dispatches]")
    print("   " + "  ".join(f"{name}={sum(v.values()) / len(v):.4g}" for name, v in sorted(c.items())))
