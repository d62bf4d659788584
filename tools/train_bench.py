#!/usr/bin/env python
"""BASELINE config 5 on one GPU: Swift-B multistep-CRPS finetune iteration (ensemble 2, `steps` rollout steps, AdamW),
local batch B.  Prints seconds per iteration and a per-kernel share if run under rocprofv3."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swift_amd.data.era5 import SyntheticERA5Dataset
from swift_amd.models.precond import PassPrecond
from swift_amd.train import adamw_param_groups
from swift_amd.training.loss import CRPSLoss, SCMLoss, TrigFlowLoss
from swift_amd.training.trainer import Trainer
from swift_amd.utils.detinit import swinv2_state

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--iters", type=int, default=6); ap.add_argument("--depth", type=int, default=12)
ap.add_argument("--loss", default="crps", choices=["crps", "scm", "trigflow"])
ap.add_argument("--opt", default="adamw", choices=["adamw", "muon"])
ap.add_argument("--dim", type=int, default=1056); ap.add_argument("--heads", type=int, default=12)  # 1280 / 16, 1536 / 16: the larger variants
ap.add_argument("--grad-digest", default=None, help="write per-parameter gradient norms of the first and the last iteration here (JSON)")
ap.add_argument("--dist", type=int, default=1, help="1: run the gradient collectives for real (a one-rank RCCL group unless launched "
                "under torchrun), so that the record carries the all-reduce's serial time, exposed wait and overlap fraction")
a = ap.parse_args()
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
torch.cuda.set_device(dev)
torch.manual_seed(1234)  # (the losses draw their latents from the global generator: two runs see the same draws)
if a.dist:
    import socket
    import torch.distributed as tdist
    if "RANK" in os.environ:  # torchrun: one rank per GPU
        tdist.init_process_group("nccl", device_id=dev)
    else:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        tdist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    import atexit
    atexit.register(lambda: tdist.is_initialized() and tdist.destroy_process_group())
names = ["2m_temperature", "10m_u_component_of_wind", "10m_v_component_of_wind", "mean_sea_level_pressure"]
for v in ["geopotential", "u_component_of_wind", "v_component_of_wind", "temperature", "specific_humidity"]:
    names += [f"{v}_{l}" for l in [50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925, 1000]]
ds = SyntheticERA5Dataset(names, ["f0", "f1", "f2"], img_resolution=(128, 256), length=64, seed=1)
mcfg = dict(_target_="swift.models.swinv2.SwinV2", window_size=[16, 16], shift_size=[8, 8], patch_size=[2, 2], depth=a.depth,
            dim=a.dim, heads=a.heads)
net = PassPrecond(mcfg, img_resolution=[128, 256], img_channels=69, condition_channels=72, auxiliary_dim=1)
net.load_state_dict(swinv2_state(grid=(64, 128), in_channels=141, out_channels=69, patch_size=(2, 2), depth=a.depth, dim=a.dim,
                                 heads=a.heads, seed=1))
net = net.to(dev).train().requires_grad_(True)
if a.opt == "muon":  # the sCM experiment's optimiser (train.py:286-309 parameter split)
    from swift_amd.training.optimizers.muon import MuonWithAuxAdam
    mp = [p for n, p in net.named_parameters() if p.ndim >= 2 and "transformer" in n]
    ap_ = [p for n, p in net.named_parameters() if not (p.ndim >= 2 and "transformer" in n)]
    opt = MuonWithAuxAdam([dict(params=mp, use_muon=True, lr=0.02, weight_decay=0.01),
                           dict(params=ap_, use_muon=False, lr=3e-4, betas=(0.9, 0.95), weight_decay=0.01, eps=1e-10)])
else:
    opt = torch.optim.AdamW(adamw_param_groups(net, 1e-5), lr=1e-5, betas=(0.9, 0.95), eps=1e-6)
noise = dict(dist="loguniform", sigma_min=0.02, sigma_max=200.0)
loss_fn = (CRPSLoss(ds, 1.0, 2, 1.0) if a.loss == "crps" else TrigFlowLoss(ds, noise, 1.0) if a.loss == "trigflow" else
           SCMLoss(ds, noise, 1.0, tangent_warmup_kimg=1)).to(dev)
tr = Trainer(net, opt, loss_fn, total_kimg=1, lr_rampup_kimg=0, lr_min_factor=1.0, device=dev,
             checkpoint_ticks=None)
tr.global_batch_size = a.batch
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(a.batch, 72, 128, 256, generator=g, device=dev)
t = 0.3 * torch.randn(a.batch, 69, 128, 256, generator=g, device=dev)
delta, idx = torch.full((a.batch,), 0.6).pin_memory(), list(range(a.batch))  # host side, as Trainer._get_batch hands them over
losses = []
def nan_report(tag):  # SWIFTK_NAN_DEBUG=1: where does a non-finite value first appear?
    if not os.environ.get("SWIFTK_NAN_DEBUG"):
        return
    torch.cuda.synchronize()
    bad_p = [n for n, p in net.named_parameters() if not torch.isfinite(p).all()]
    bad_g = [n for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(f"[nan-debug] {tag}: loss {float(loss):.4f}; non-finite params {len(bad_p)} {bad_p[:3]}; non-finite grads {len(bad_g)} {bad_g[:3]}", file=sys.stderr)


digest = {}
def grad_digest(tag):  # per-tensor (L2 norm, sum) of the gradients as the optimiser saw them (after the all-reduce), fp64 on the device
    if a.grad_digest:
        digest[tag] = {n: [float(p.grad.double().norm()), float(p.grad.double().sum())] for n, p in net.named_parameters() if p.grad is not None}


for _ in range(3):  # warm-up: operand prep and allocator, then the HIP-graph capture of every launch sequence, then one replay
    loss = tr.train_step(x, t, idx, delta, 1000, steps=a.steps)
    losses.append(loss)
    nan_report(f"warm-up {_}")
    if _ == 0:
        grad_digest("first")
torch.cuda.synchronize()
if os.environ.get("SWIFTK_SYNC_DEBUG"):  # list every call that makes the host wait for the GPU inside the timed iterations
    import collections, traceback, warnings
    sync_sites = collections.Counter()
    def _show(message, category, filename, lineno, file=None, line=None):
        if "synchroniz" in str(message):
            st = [f for f in traceback.extract_stack() if "swift_amd" in f.filename or "train_bench" in f.filename]
            sync_sites[" <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-4:][::-1])] += 1
    warnings.showwarning = _show
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode(1)
tr.ddp._timing, tr.ddp._announced_hist = [], []  # (the warm-up iterations' waits are not the steady state's)
t0 = time.perf_counter()
host_ms = []
for k in range(a.iters):
    h0 = time.perf_counter()
    loss = tr.train_step(x, t, idx, delta, 1000 * (k + 2), steps=a.steps)
    host_ms.append(1e3 * (time.perf_counter() - h0))  # time the host needs to ISSUE an iteration (it may run ahead of the GPU)
    losses.append(loss)
    nan_report(f"iteration {k}")
if os.environ.get("SWIFTK_SYNC_DEBUG"):
    torch.cuda.set_sync_debug_mode(0)
    for site, n in sync_sites.most_common(40):
        print(f"sync x{n}: {site}", file=sys.stderr)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
grad_digest("last")
if a.grad_digest:
    with open(a.grad_digest, "w") as f:
        json.dump(digest, f)
_bad = [n for n, p in net.named_parameters() if not torch.isfinite(p).all()]
# which parameters ever saw an overflowing gradient (exp_avg_sq = inf)?  Always checked: the second moment remembers an overflow of
# ANY iteration of the run, also where nan_to_num kept the parameters finite (the round-5 data-parallel overflow, DESIGN)
_inf = [(n, int((~torch.isfinite(opt.state[p]["exp_avg_sq"])).sum()), p.numel()) for n, p in net.named_parameters()
        if p in opt.state and "exp_avg_sq" in opt.state[p] and not torch.isfinite(opt.state[p]["exp_avg_sq"]).all()]
print(f"OVERFLOW-CHECK parameters with non-finite exp_avg_sq: {len(_inf)}: {_inf[:40]}", file=sys.stderr)
for n, p in net.named_parameters():  # where inside a tensor: contiguous ranges? a stride?  (first few partially hit tensors)
    if any(n == q[0] and q[1] < q[2] for q in _inf[:60]) and ("layers.0." in n or "layers.11." in n):
        ix = (~torch.isfinite(opt.state[p]["exp_avg_sq"])).flatten().nonzero().flatten().tolist()
        runs, a0 = [], ix[0]
        for u, v in zip(ix, ix[1:] + [None]):
            if v != u + 1:
                runs.append((a0, u))
                a0 = v
        print(f"OVERFLOW-WHERE {n} {tuple(p.shape)}: {len(ix)} elements in {len(runs)} runs: {runs[:12]}", file=sys.stderr)
if _bad:  # (a non-finite parameter after the run: name the first few -- which kernel's output went wrong?)
    print(f"NON-FINITE PARAMETERS after {a.iters} iterations: {len(_bad)} of {len(list(net.parameters()))}: {_bad[:12]}", file=sys.stderr)
    for n, p in net.named_parameters():
        if n in _bad[:4]:
            bad = (~torch.isfinite(p)).flatten().nonzero().flatten()
            st = opt.state.get(p, {})
            ea, ev = st.get("exp_avg"), st.get("exp_avg_sq")
            print(f"  {n}: {bad.numel()} of {p.numel()} elements, flat indices {int(bad[0])}..{int(bad[-1])}; exp_avg non-finite "
                  f"{int((~torch.isfinite(ea)).sum()) if ea is not None else None}, |exp_avg| max {float(ea[torch.isfinite(ea)].abs().max()) if ea is not None else None:.3e}; "
                  f"exp_avg_sq non-finite {int((~torch.isfinite(ev)).sum()) if ev is not None else None}", file=sys.stderr)
if os.environ.get("SWIFTK_TUNE", "").startswith("25:") and int(os.environ["SWIFTK_TUNE"].split(",")[0][3:]) & 4:
    import ctypes
    from swift_amd import _lib
    rep = (ctypes.c_ulonglong * 26)()
    _lib.lib().swiftk_zero_check_report(rep)
    import struct
    f = lambda b: struct.unpack("f", struct.pack("I", int(b) & 0xffffffff))[0]
    print(f"ZERO-CHECK {rep[0]} clears checked; non-zero dwords left behind by index mod 4: {list(rep[1:5])}; largest |value| by index mod 4: "
          f"{[f(b) for b in rep[5:9]]}; samples (index, bits, as float): {[(int(rep[10 + k]), hex(rep[18 + k]), f(rep[18 + k])) for k in range(min(int(rep[9]), 8))]}",
          file=sys.stderr)
print("host issue time per iteration (ms):", " ".join(f"{h:.0f}" for h in host_ms), file=sys.stderr)
print("loss per iteration (warm-up included):", " ".join(f"{float(l):.4f}" for l in losses), file=sys.stderr)
PEAK = 2.5e15  # dense bf16 MFMA peak (MI355X_MICROARCH.md)
def _fwd_flop(dim, depth):  # one network evaluation of one sample (SURVEY.md section 8d's count, any width)
    ntok, mlp = 64 * 128, int(8 / 3 * dim)
    layer = 2 * ntok * (dim * 3 * dim + dim * dim + dim * 2 * mlp + mlp * dim) + 4 * ntok * 256 * dim + 2 * 2 * dim * 2 * dim
    return 2 * ntok * 564 * dim + depth * layer + 2 * ntok * dim * 276
FWD = _fwd_flop(a.dim, a.depth) if a.dim != 1056 else 2.7535e12 * a.depth / 12
AR = {}
if a.dist:
    tr.ddp.calibrate_serial()
    AR = tr.ddp.allreduce_stats()
    AR["note"] = ("gradient all-reduce of this rank: serial_ms = one blocking all-reduce of the whole flat buffer, exposed_wait_ms = what the "
                  "compute stream waited for the collectives in sync() per iteration, overlap_frac = 1 - exposed / serial; a one-rank group "
                  "measures the call path, not xGMI")
_dumps = json.dumps
json.dumps = lambda rec, *aa, **kk: _dumps(dict(rec, allreduce=AR) if isinstance(rec, dict) and "metric" in rec else rec, *aa, **kk)
fused = bool(getattr(tr, "_fused", None))
if a.loss == "trigflow":  # TrigFlowLoss (loss.py:117-160): one forward, one backward (2x) per sample
    fl = 3 * a.batch * FWD
    print(json.dumps({"metric": "TrigFlow training iteration (Swift-B, local batch %d, optimizer %s)" % (a.batch, a.opt), "value": dt, "unit": "s/iteration",
                      "samples_per_s": a.batch / dt, "flop_per_iteration": fl, "what": "3 forward-equivalents per sample (forward 1, backward 2)",
                      "roofline": {"bound": "mfma", "achieved": fl / dt / 1e12, "peak": PEAK / 1e12, "unit": "TFLOP/s", "frac": fl / dt / PEAK, "traffic": None},
                      "fused_optimizer_step": fused, "peak_mem_gib": torch.cuda.max_memory_allocated() / 2**30}))
    print(f"TrigFlow training: batch {a.batch}, depth {a.depth}: {dt:.3f} s/iteration, loss {float(loss):.4f}; {a.batch / dt:.2f} samples/s; "
          f"{fl / dt / 1e12:.0f} TFLOP/s; peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    sys.exit(0)
if a.loss == "scm":
    # tangent pass = every GEMM on 2M rows (2 fwd) + backward (2); the reference's schedule adds a grad-enabled forward (1), which
    # the one-pass form does not execute (the tangent pass's primal rows are the saved activations)
    one_pass = bool(getattr(loss_fn, "last_one_pass", False))
    fwd_eq = 4 if one_pass else 5
    fl = fwd_eq * a.batch * FWD
    print(json.dumps({"metric": "sCM pre-training iteration (Swift-B, local batch %d, optimizer %s)" % (a.batch, a.opt), "value": dt, "unit": "s/iteration",
                      "samples_per_s": a.batch / dt, "flop_per_iteration": fl,
                      "what": f"~{fwd_eq} forward-equivalents executed per sample (tangent pass 2, backward 2" + ("" if one_pass else ", forward 1") +
                              "; the reference's schedule: 5)", "one_pass": one_pass,
                      "roofline": {"bound": "mfma", "achieved": fl / dt / 1e12, "peak": PEAK / 1e12, "unit": "TFLOP/s", "frac": fl / dt / PEAK, "traffic": None},
                      "fused_optimizer_step": fused, "peak_mem_gib": torch.cuda.max_memory_allocated() / 2**30}))
    print(f"sCM pre-training: batch {a.batch}, depth {a.depth}: {dt:.3f} s/iteration, loss {float(loss):.4f}; "
          f"{a.batch / dt:.2f} samples/s; ~{fwd_eq} fwd-equivalents -> {fl / dt / 1e12:.0f} TFLOP/s; "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    sys.exit(0)
evals = 2 * a.steps
n_keep = int(getattr(loss_fn, "last_n_keep", 0))  # rollout steps whose activations stayed in HBM: no recomputed forward for those
fwd_eq = 4 * evals - n_keep  # EXECUTED work; the reference's schedule (checkpoint_sequential) is 4 * evals
fl = fwd_eq * a.batch * FWD
print(json.dumps({"metric": "multistep-CRPS finetune iteration (%s, steps %d, ensemble 2, local batch %d; BASELINE configs[4] per GPU)" % ("Swift-B" if a.dim == 1056 else "dim %d / %d heads / depth %d" % (a.dim, a.heads, a.depth), a.steps, a.batch),
                  "value": dt, "unit": "s/iteration", "samples_per_s": a.batch / dt, "flop_per_iteration": fl,
                  "what": f"{evals} rollout forwards + {evals - n_keep} recomputed forwards + {evals} backwards (2x) = {fwd_eq} forward-equivalents executed per sample "
                          f"({n_keep} of {evals} rollout steps keep their activations; the reference's schedule recomputes all: {4 * evals})",
                  "kept_rollout_steps": n_keep,
                  "roofline": {"bound": "mfma", "achieved": fl / dt / 1e12, "peak": PEAK / 1e12, "unit": "TFLOP/s", "frac": fl / dt / PEAK, "traffic": None},
                  "fused_optimizer_step": fused, "peak_mem_gib": torch.cuda.max_memory_allocated() / 2**30}))
print(f"CRPS finetune: batch {a.batch}, steps {a.steps}, depth {a.depth}: {dt:.3f} s/iteration, loss {float(loss):.4f}; "
      f"{a.batch / dt:.2f} samples/s; fwd-equivalents/iter = {evals} fwd + {evals - n_keep} recompute + {evals} bwd(2x) -> "
      f"{fl / dt / 1e12:.0f} TFLOP/s; peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
