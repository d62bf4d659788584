#!/usr/bin/env python
"""Kernel-level stressor for round 5's data-parallel gradient overflow (VERDICT r5 item 3 / ADVICE r5): the REAL
``swiftk_modnorm_bwd`` one-kernel form (per-sample column sums accumulated with atomics into a workspace) in a loop shaped
like the eager backward pass of a data-parallel CRPS iteration -- GEMM-sized neighbours on the stream, a one-rank RCCL group
whose asynchronous all-reduces put cross-stream event waits on the stream -- with the workspace cleared per call by

  memset : hipMemsetAsync                       (``swiftk_set_tuning(25, 3)``: the round-4/5 library)
  kernel : the library's fill kernel            (what ships)
  ws0    : nothing per call; the finishing kernel leaves the workspace zero (what the training engine uses)

and every call's d gamma / d beta / d modulation compared with the two-kernel form (row pass + column pass, no workspace
sums at all) on the same inputs.  Prints one line per (mode, collectives on/off): calls, mismatching calls, worst deviation.
usage: zero_order_stress.py [calls per configuration, default 2000] [--no-group]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.distributed as tdist
from swift_amd import _lib, ops, dist

CALLS = int(next((a for a in sys.argv[1:] if a.isdigit()), 2000))
USE_GROUP = "--no-group" not in sys.argv
if USE_GROUP:
    dist.setup_torch(single_rank_group=True)
L = _lib.lib()
B, rps, d = 8, 8192, 1056
M = B * rps
dev = torch.device("cuda", 0)
BF = torch.bfloat16
torch.manual_seed(0)
y = torch.randn(M, d, device=dev).to(BF)
g = torch.randn(M, d, device=dev)
gamma, beta = torch.randn(d, device=dev), torch.randn(d, device=dev)
mod = torch.randn(B, 2 * d, device=dev) * 0.1
dy = torch.zeros(M, ops.k_pad(BF, d), dtype=BF, device=dev)
ws = torch.empty(2 * M, device=dev)
a = torch.randn(8192, 1088, device=dev).to(BF)
w = torch.randn(1056, 1088, device=dev).to(BF)
c = torch.empty(8192, 1056, dtype=BF, device=dev)
flat = torch.zeros(1 << 22, device=dev)  # 16 MB "gradient slice" for the collectives
s = lambda: torch.cuda.current_stream().cuda_stream


def call(fn, out):
    dg, db, dm = out
    assert fn(y.data_ptr(), d, g.data_ptr(), dy.data_ptr(), dy.stride(0), gamma.data_ptr(), beta.data_ptr(), mod.data_ptr(), 2 * d,
              dg.data_ptr(), db.data_ptr(), dm.data_ptr(), 2 * d, ws.data_ptr(), M, d, rps, 1e-6, _lib.BF16, s()) == 0


def fresh():
    return [ops.zeros_acc(d, device=dev), ops.zeros_acc(d, device=dev), ops.zeros_acc(B, 2 * d, device=dev)]


# reference: the two-kernel form (tuning key 16 = 0): no atomically accumulated workspace
L.swiftk_set_tuning(16, 0)
ref = fresh()
call(L.swiftk_modnorm_bwd, ref)
torch.cuda.synchronize()
L.swiftk_set_tuning(16, 1)
scale = [float(t.abs().max()) for t in ref]
print(f"# {CALLS} calls per configuration, M = {M}, d = {d}, process group: {tdist.get_backend() if USE_GROUP else None}; reference "
      f"maxima {scale[0]:.3g} {scale[1]:.3g} {scale[2]:.3g}", flush=True)

for mode in ("memset", "kernel", "ws0"):
    for coll in ((False, True) if USE_GROUP else (False,)):
        L.swiftk_set_tuning(25, 3 if mode == "memset" else 0)
        fn = L.swiftk_modnorm_bwd_ws0 if mode == "ws0" else L.swiftk_modnorm_bwd
        ops.zero_acc_(ws)
        torch.cuda.synchronize()
        bad = torch.zeros(1, device=dev)
        worst = torch.zeros(1, device=dev)
        pending = []
        outs = [fresh() for _ in range(8)]
        for i in range(CALLS):
            out = outs[i % 8]
            for t in out:
                ops.zero_acc_(t)
            ops.gemm(a, w, c)  # a neighbour on the stream, as in the backward pass
            if coll and i % 3 == 0:
                pending.append(tdist.all_reduce(flat, op=tdist.ReduceOp.AVG, async_op=True))
            # poison what the clear is supposed to clear (a late / skipped clear then shows as an overflow-sized error);
            # the ws0 form never clears per call, so it is not poisoned
            if mode != "ws0":
                ws[:2 * B * d].fill_(3e19)
            call(fn, out)
            if coll and len(pending) >= 2:
                pending.pop(0).wait()  # an event wait on the compute stream
            err = torch.stack([((o - r).abs().max() / sc) for o, r, sc in zip(out, ref, scale)]).max()
            err = torch.nan_to_num(err, nan=1e30, posinf=1e30)
            bad += (err > 1e-3).float()
            worst = torch.maximum(worst, err.reshape(1))
            if i % 64 == 63:
                torch.cuda.synchronize()
        for h in pending:
            h.wait()
        torch.cuda.synchronize()
        print(f"clear={mode:7s} collectives={int(coll)}: {CALLS} calls, {int(bad.item())} mismatching, worst relative deviation "
              f"{float(worst.item()):.3g}", flush=True)
L.swiftk_set_tuning(25, 0)
if USE_GROUP:
    tdist.destroy_process_group()
