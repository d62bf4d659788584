#!/usr/bin/env python
"""Top rows of a rocprofv3 *kernel_stats.csv with short kernel names: python tools/kstats.py <dir-or-file> [rows]"""
import csv, glob, os, re, sys
p = sys.argv[1]
f = p if os.path.isfile(p) else glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
tot = 0.0
rows = list(csv.DictReader(open(f)))
for r in rows:
    tot += float(r["TotalDurationNs"])
for r in rows[:n]:
    name = r["Name"]
    m = re.search(r"(gemm_kernel_p<[^>]*>|gemm_kernel<[^>]*>|\w+_kernel(<[^>]*>)?)", name)
    short = (m.group(1) if m else name[:56]).replace("unsigned short", "bf16")
    print(f"{short:58s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:9.1f} us  {r['Percentage']:>6s}%")
print(f"total kernel time {tot / 1e6:.1f} ms")
