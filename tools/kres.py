#!/usr/bin/env python
"""Kernel resource usage of one csrc/*.hip (VGPRs / SGPRs / scratch / LDS per kernel): tools/kres.py gemm.hip [-DFOO=1 ...]"""
import os, re, subprocess, sys
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "swift_amd", "csrc", sys.argv[1])
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
                      *sys.argv[2:], "-c", src, "-o", "/tmp/kres.o"], capture_output=True, text=True).stderr
cur = {}
rows = []
for ln in out.splitlines():
    m = re.search(r"(Function Name|Name): (\S+)", ln)
    if m:
        cur = {"name": m.group(2)}
        rows.append(cur)
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
        m = re.search(pat, ln)
        if m and cur is not None:
            cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n).replace("unsigned short", "bf16")
    print(f"{n:70s} vgpr {r.get('vgpr')} agpr {r.get('agpr')} sgpr {r.get('sgpr')} scratch {r.get('scratch')} lds {r.get('lds')} occ {r.get('occ')}")
