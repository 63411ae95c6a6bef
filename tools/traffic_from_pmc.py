#!/usr/bin/env python3
"""Turn the two PMC passes (FETCH_SIZE, WRITE_SIZE) of `rocprofv3 --pmc ... -- python3 bench.py ...` into per-launch HBM/fabric
traffic per kernel class, with the gfx950 corrections of MI355X_MICROARCH.md §HBM:
    read bytes  = 2 x FETCH_SIZE x 1024   (FETCH_SIZE tallies 128-B requests at 64 B)
    write bytes =     WRITE_SIZE x 1024
Usage: traffic_from_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import collections, csv, json, re, sys

def load(path, counter, by_name=False):
    per = collections.defaultdict(lambda: [0.0, 0])
    seen = set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        key = (r["Dispatch_Id"], r["Kernel_Name"])
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        cls = ("gemm_nt" if ("gemm_nt_kernel" in name or "gemm_nt8_kernel" in name) else
               "gemm_tn" if ("gemm_tn_kernel" in name or "gemm_tn8_kernel" in name) else
               "splitk_reduce" if "splitk_reduce" in name else "attn_fwd" if "attn_fwd" in name else
               "attn_bwd" if "attn_bwd" in name else "ln_fwd" if "ln_fwd" in name else "ln_bwd" if "ln_bwd" in name else None)
        if cls is None:
            continue
        if by_name:
            cls = re.sub(r"\(Gemm(NT|TN)Params\)|^void ", "", name)[:80]
        per[cls][0] += float(r["Counter_Value"])
        if key not in seen:
            seen.add(key)
            per[cls][1] += 1
    return per

fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
out = {"correction": "read = 2 x FETCH_SIZE KiB, write = WRITE_SIZE KiB (MI355X_MICROARCH.md, HBM section)", "per_launch_bytes": {}}
for cls in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(cls, [0, 0]); w, nw = write.get(cls, [0, 0])
    n = max(nf, nw, 1)
    out["per_launch_bytes"][cls] = {"launches_profiled": n, "read": round(2 * f * 1024 / n), "write": round(w * 1024 / n),
                                    "total": round((2 * f + w) * 1024 / n)}
# the same per kernel NAME (the GEMM's tile / epilogue template arguments): which launches move what
fetch_n, write_n = load(sys.argv[1], "FETCH_SIZE", True), load(sys.argv[2], "WRITE_SIZE", True)
out["per_launch_bytes_by_kernel"] = {}
for name in sorted(set(fetch_n) | set(write_n)):
    f, nf = fetch_n.get(name, [0, 0]); w, nw = write_n.get(name, [0, 0])
    n = max(nf, nw, 1)
    out["per_launch_bytes_by_kernel"][name] = {"launches_profiled": n, "read": round(2 * f * 1024 / n), "write": round(w * 1024 / n)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["per_launch_bytes"], indent=1))
