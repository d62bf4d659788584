#!/usr/bin/env python
"""Which HIP runtime calls hold the host up inside one training iteration?
tools/trace_api_gaps.py <kernel_trace.csv> <hip_api_trace.csv> [top]
The iteration window is the span between the third-last and second-last optimizer-step kernels of the kernel trace (as
tools/trace_gaps.py); prints the longest HIP API calls that START inside it (name, duration, offset) and the per-function
totals -- a hipMalloc / hipFree / synchronous hipMemcpy here is GPU idle time when the host is not running ahead."""
import collections, csv, sys
kr = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in kr)
idx = [i for i, e in enumerate(ev) if "adamw_ema" in e[2] or "muon" in e[2].lower()]
t0, t1 = ev[idx[-3]][0], ev[idx[-2]][0]
api = []
for r in csv.DictReader(open(sys.argv[2])):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 <= s < t1:
        api.append((e - s, s - t0, r.get("Function") or r.get("Name") or "?"))
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
print(f"iteration window {(t1 - t0) / 1e6:.1f} ms, {len(api)} HIP API calls inside")
tot = collections.Counter(); cnt = collections.Counter()
for d, _, f in api:
    tot[f] += d; cnt[f] += 1
for f, d in tot.most_common(12):
    print(f"  {f:40s} total {d / 1e6:8.2f} ms in {cnt[f]} calls")
print("longest calls:")
for d, at, f in sorted(api, reverse=True)[:top]:
    print(f"  {d / 1e3:9.1f} us at {at / 1e6:7.1f} ms  {f}")
