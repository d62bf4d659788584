#!/usr/bin/env python
"""From a rocprofv3 --kernel-trace CSV: every dispatch of the runtime's fill kernel (__amd_rocclr_fillBufferAligned = a device
memset) with the kernel that FOLLOWS it in time, as a histogram -- is any accumulating kernel of this library fed by a memset?
usage: fillbuffer_neighbours.py <dir-or-kernel_trace.csv>"""
import collections, csv, glob, os, re, sys
p = sys.argv[1]
f = p if os.path.isfile(p) else glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: re.sub(r"\(.*", "", n).replace("(anonymous namespace)::", "").replace("void ", "")[:70]
t0 = int(rows[0]["Start_Timestamp"])
span = int(rows[-1]["End_Timestamp"]) - t0
hist, when = collections.Counter(), []
for i, r in enumerate(rows):
    if "fillBuffer" in r["Kernel_Name"]:
        nxt = next((short(q["Kernel_Name"]) for q in rows[i + 1:i + 4] if "fillBuffer" not in q["Kernel_Name"]), "(end of trace)")
        hist[nxt] += 1
        when.append((int(r["Start_Timestamp"]) - t0) / span)
print(f"{len(rows)} dispatches, {len(when)} fill-kernel dispatches; position in the trace (fraction of its span): "
      f"first {min(when):.3f}, last {max(when):.3f}, after the first 25 % of the span: {sum(1 for w in when if w > 0.25)}")
for k, c in hist.most_common(25):
    print(f"  x{c:4d} followed by {k}")
