#!/usr/bin/env python
"""Per-kernel means of rocprofv3 --pmc counter_collection.csv files (one or more passes)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        k = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items(), key=lambda kv: -sum(len(v) for v in kv[1].values())):
    if not any(s in k for s in ("gemm", "attn", "modnorm", "qkv")): continue
    print(k)
    for c, v in cs.items():
        print(f"    {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
