#!/usr/bin/env python
"""Per-kernel register / scratch / LDS use of every HIP source (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).
usage: kernel_resources.py [--spills] [file.hip ...]      --spills: only kernels with scratch (= spilled registers)"""
import os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "swift_amd", "csrc")
only = "--spills" in sys.argv
files = [a for a in sys.argv[1:] if a.endswith(".hip")] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(anonymous namespace\)::", "", o).split("(")[0].replace("void ", "") for o in out]


NO_SLP = {"gemm.hip", "gemm_tn.hip", "gemm_rownorm.hip", "qkv_attn.hip"}  # the Makefile's per-file -fno-slp-vectorize: same code as shipped


def one(f):
    extra = ["-fno-slp-vectorize"] if os.path.basename(f) in NO_SLP else []
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", *extra,
                        "-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/dev/null"], cwd=CSRC, capture_output=True, text=True)
    rows, cur = [], None
    for ln in r.stderr.splitlines():
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[bytes/lane\]| \[bytes/block\])?: +(\S+)", ln)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v, "file": f}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    return rows


with ThreadPoolExecutor(4) as ex:
    rows = [r for rs in ex.map(one, files) for r in rs]
for r, n in zip(rows, demangle([r["name"] for r in rows])):
    r["name"] = n
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'LDS':>7s}  file")
bad = 0
for r in rows:
    sc = int(r.get("ScratchSize", 0))
    bad += sc > 0
    if only and sc == 0:
        continue
    print(f"{r['name'][:70]:70s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {sc:8d} {r.get('LDS Size','?'):>7s}  {r['file']}")
print(f"{bad} of {len(rows)} kernels use scratch")
