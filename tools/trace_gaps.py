#!/usr/bin/env python
"""GPU idle time inside one training iteration of a rocprofv3 kernel trace (csv): tools/trace_gaps.py <kernel_trace.csv>
An iteration = the span between two optimizer-step kernels (adamw_ema / muon); prints busy time, idle time by gap size
and the largest gaps with the kernels on either side."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r in rows)
idx = [i for i, e in enumerate(ev) if "adamw_ema" in e[2]]
a, b = idx[-3], idx[-2]
seg = ev[a:b]
busy = sum(e[1] - e[0] for e in seg)
span = seg[-1][1] - seg[0][0]
gaps = [(q[0] - p[1], p[2], q[2], p[1] - seg[0][0]) for p, q in zip(seg[:-1], seg[1:]) if q[0] > p[1]]
print(f"iteration span {span / 1e6:.1f} ms, kernels busy {busy / 1e6:.1f} ms, idle {sum(g[0] for g in gaps) / 1e6:.1f} ms in {len(gaps)} gaps")
for lo, hi in ((0, 10e3), (10e3, 100e3), (100e3, 1e6), (1e6, 1e12)):
    sel = [g for g in gaps if lo <= g[0] < hi]
    print(f"  gaps {lo / 1e3:.0f}-{hi / 1e3:.0f} us: {sum(g[0] for g in sel) / 1e6:7.2f} ms ({len(sel)})")
for g, p, q, at in sorted(gaps, reverse=True)[: int(sys.argv[2]) if len(sys.argv) > 2 else 15]:
    print(f"{g / 1e3:9.1f} us at {at / 1e6:7.1f} ms  after {p[:40]:40s} before {q[:40]}")
