#!/usr/bin/env python
"""Derived figures from the SQ counter passes of tools/profile_attn_counters.sh: attn_counters_report.py <dir of one mode> ...
Per mode (directory pair *_1 / *_2): mean kernel duration (kernel trace of the same runs), effective clock
(GRBM_GUI_ACTIVE / 8 XCDs / duration), MFMA-pipe busy share of the SIMD cycles (SQ_VALU_MFMA_BUSY_CYCLES counts cycles,
1024 SIMDs), the split of wave time (SQ_WAVE_CYCLES = ACTIVE_INST_ANY + WAIT_INST_ANY + WAIT_ANY, quad-cycles), VALU and LDS
instructions per MFMA, LDS bank-conflict share.  With `fused` and `fused_nocore` both given: the attention core alone (difference)."""
import csv, glob, os, sys, collections
def load(prefix):
    c = collections.defaultdict(list); dur = []
    for d in (prefix + "_1", prefix + "_2"):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if any(k in r["Kernel_Name"] for k in ("qkv_attn", "gemm_kernel_p", "gemm_tn_kernel", "gemm_rownorm_kernel")):
                    c[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if any(k in r["Kernel_Name"] for k in ("qkv_attn", "gemm_kernel_p", "gemm_tn_kernel", "gemm_rownorm_kernel")):
                    dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    m = {k: sum(v[1:]) / max(len(v) - 1, 1) for k, v in c.items()}  # (first launch of a process dropped)
    m["dur_ns"] = sum(sorted(dur)[: max(len(dur) // 2, 1)]) / max(len(dur) // 2, 1)  # faster half: the un-throttled launches
    return m
def show(name, m):
    ns = m["dur_ns"]; clk = m.get("GRBM_GUI_ACTIVE", 0) / 8 / ns if ns else 0
    simd_cyc = (clk if clk else 2.0) * ns * 1024
    wc = m["SQ_WAVE_CYCLES"]
    print(f"{name}: {ns / 1e6:.3f} ms per launch, effective clock {clk:.2f} GHz")
    print(f"  MFMA pipe busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cyc:.3f} of the SIMD cycles "
          f"({m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1e9:.2f}e9 of {simd_cyc / 1e9:.2f}e9)")
    print(f"  wave time: issuing {m['SQ_ACTIVE_INST_ANY'] / wc:.3f} (VALU incl. MFMA issue {m['SQ_ACTIVE_INST_VALU'] / wc:.3f}, LDS {m['SQ_ACTIVE_INST_LDS'] / wc:.3f}), "
          f"issue-stalled {m['SQ_WAIT_INST_ANY'] / wc:.3f} (LDS {m['SQ_WAIT_INST_LDS'] / wc:.3f}), parked at waitcnt / barrier {m['SQ_WAIT_ANY'] / wc:.3f}")
    mf = m["SQ_INSTS_MFMA"]
    print(f"  per MFMA: {(m['SQ_INSTS_VALU'] - mf) / mf:.2f} other VALU, {m['SQ_INSTS_LDS'] / mf:.2f} LDS, {m['SQ_INSTS_SALU'] / mf:.2f} SALU, {m['SQ_INSTS_VMEM'] / mf:.3f} VMEM instructions"
          f"   (if SQ_INSTS_VALU excludes MFMA: {m['SQ_INSTS_VALU'] / mf:.2f})")
    print(f"  LDS bank-conflict cycles {m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:.3f} of the LDS-active cycles")
ms = {os.path.basename(p).split("_", 2)[-1] if False else p.rsplit("_", 0)[0]: load(p) for p in sys.argv[1:]}
for p, m in ms.items():
    show(os.path.basename(p), m)
f = next((m for p, m in ms.items() if p.endswith("fused")), None)
n = next((m for p, m in ms.items() if p.endswith("fused_nocore")), None)
if f and n:
    d = {k: f[k] - n[k] for k in f if k in n}
    clk = f.get("GRBM_GUI_ACTIVE", 0) / 8 / f["dur_ns"] or 2.0
    simd = clk * d["dur_ns"] * 1024
    print(f"attention core alone (fused minus fused_nocore): {d['dur_ns'] / 1e6:.3f} ms per launch")
    print(f"  MFMA pipe busy {d['SQ_VALU_MFMA_BUSY_CYCLES'] / simd:.3f} of its SIMD cycles; per MFMA: "
          f"{(d['SQ_INSTS_VALU'] - d['SQ_INSTS_MFMA']) / d['SQ_INSTS_MFMA']:.2f} other VALU (or {d['SQ_INSTS_VALU'] / d['SQ_INSTS_MFMA']:.2f}), {d['SQ_INSTS_LDS'] / d['SQ_INSTS_MFMA']:.2f} LDS instructions; "
          f"LDS bank conflicts {d['SQ_LDS_BANK_CONFLICT'] / max(d['SQ_LDS_IDX_ACTIVE'], 1):.3f} of LDS-active cycles")
    print(f"  wave time of the core: issuing {d['SQ_ACTIVE_INST_ANY'] / d['SQ_WAVE_CYCLES']:.3f}, issue-stalled {d['SQ_WAIT_INST_ANY'] / d['SQ_WAVE_CYCLES']:.3f}, parked {d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES']:.3f}")
