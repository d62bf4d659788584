#!/usr/bin/env python
"""Which host call sites issue a device MEMSET (hipMemsetAsync / hipMemset: the runtime's fill kernel) during a training
iteration?  A memset captured into a HIP graph and replayed on the null stream writes a stale pattern under the HIP runtime
this PyTorch bundles (tools/memset_graph_repro.hip), so NONE may sit inside a captured
launch sequence.  torch.profiler over one eager iteration (SWIFTK_TRAIN_GRAPHS=0) of tools/train_bench.py's set-up; every
runtime memset call is listed with the operator and Python frames above it.
usage: memset_sites.py [crps|scm|trigflow]"""
import collections, os, sys
os.environ["SWIFTK_TRAIN_GRAPHS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from swift_amd.data.era5 import SyntheticERA5Dataset
from swift_amd.models.precond import PassPrecond
from swift_amd.train import adamw_param_groups
from swift_amd.training.loss import CRPSLoss, SCMLoss, TrigFlowLoss
from swift_amd.training.trainer import Trainer
from swift_amd.utils.detinit import swinv2_state
which = sys.argv[1] if len(sys.argv) > 1 else "crps"
dev = torch.device("cuda", 0)
names = [f"v{i}" for i in range(69)]
ds = SyntheticERA5Dataset(names, ["f0", "f1", "f2"], img_resolution=(128, 256), length=64, seed=1)
mcfg = dict(_target_="swift.models.swinv2.SwinV2", window_size=[16, 16], shift_size=[8, 8], patch_size=[2, 2], depth=2, dim=1056, heads=12)
net = PassPrecond(mcfg, img_resolution=[128, 256], img_channels=69, condition_channels=72, auxiliary_dim=1)
net.load_state_dict(swinv2_state(grid=(64, 128), in_channels=141, out_channels=69, patch_size=(2, 2), depth=2, dim=1056, heads=12, seed=1))
net = net.to(dev).train().requires_grad_(True)
opt = torch.optim.AdamW(adamw_param_groups(net, 1e-5), lr=1e-5, betas=(0.9, 0.95), eps=1e-6)
noise = dict(dist="loguniform", sigma_min=0.02, sigma_max=200.0)
loss_fn = (CRPSLoss(ds, 1.0, 2, 1.0) if which == "crps" else TrigFlowLoss(ds, noise, 1.0) if which == "trigflow" else
           SCMLoss(ds, noise, 1.0, tangent_warmup_kimg=1)).to(dev)
tr = Trainer(net, opt, loss_fn, total_kimg=1, lr_rampup_kimg=0, lr_min_factor=1.0, device=dev, checkpoint_ticks=None)
tr.global_batch_size = 2
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(2, 72, 128, 256, generator=g, device=dev)
t = 0.3 * torch.randn(2, 69, 128, 256, generator=g, device=dev)
delta, idx = torch.full((2,), 0.6).pin_memory(), [0, 1]
steps = 4 if which == "crps" else 1
for k in range(2):
    tr.train_step(x, t, idx, delta, 1000 * (k + 1), steps=steps)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(x, t, idx, delta, 4000, steps=steps)
    torch.cuda.synchronize()
evs = list(prof.events())
sites = collections.Counter()
n_rt = 0
for e in evs:
    nm = e.name.lower()
    if "memset" in nm and e.device_type == torch.autograd.DeviceType.CPU:  # the runtime call (hipMemsetAsync), host side
        n_rt += 1
        par, chain = e.cpu_parent, []
        while par is not None and len(chain) < 3:
            chain.append(par.name)
            par = par.cpu_parent
        stack = [s for s in (e.stack or []) if "swift_amd" in s or "tools/" in s][:3]
        sites[(e.name, " <- ".join(chain), " | ".join(stack))] += 1
dev_memsets = sum(1 for e in evs if "memset" in e.name.lower() and e.device_type != torch.autograd.DeviceType.CPU)
print(f"{which}: {n_rt} runtime memset calls, {dev_memsets} device memset activities in one eager iteration (depth 2, batch 2)")
for (nm, chain, stack), c in sites.most_common(30):
    print(f"  x{c:4d} {nm}  under [{chain}]  at [{stack}]")
