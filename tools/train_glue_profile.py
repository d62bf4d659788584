#!/usr/bin/env python
"""Which ATen operators (host call sites) are behind the non-swiftk kernels of a training iteration: torch.profiler over two
TrigFlow iterations of tools/train_bench.py's set-up, eager (SWIFTK_TRAIN_GRAPHS=0 if supported), grouped by operator and stack."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from swift_amd.data.era5 import SyntheticERA5Dataset
from swift_amd.models.precond import PassPrecond
from swift_amd.train import adamw_param_groups
from swift_amd.training.loss import TrigFlowLoss
from swift_amd.training.trainer import Trainer
from swift_amd.utils.detinit import swinv2_state
dev = torch.device("cuda", 0)
names = [f"v{i}" for i in range(69)]
ds = SyntheticERA5Dataset(names, ["f0", "f1", "f2"], img_resolution=(128, 256), length=64, seed=1)
mcfg = dict(_target_="swift.models.swinv2.SwinV2", window_size=[16, 16], shift_size=[8, 8], patch_size=[2, 2], depth=12, dim=1056, heads=12)
net = PassPrecond(mcfg, img_resolution=[128, 256], img_channels=69, condition_channels=72, auxiliary_dim=1)
net.load_state_dict(swinv2_state(grid=(64, 128), in_channels=141, out_channels=69, patch_size=(2, 2), depth=12, dim=1056, heads=12, seed=1))
net = net.to(dev).train().requires_grad_(True)
opt = torch.optim.AdamW(adamw_param_groups(net, 1e-5), lr=1e-5, betas=(0.9, 0.95), eps=1e-6)
loss_fn = TrigFlowLoss(ds, dict(dist="loguniform", sigma_min=0.02, sigma_max=200.0), 1.0).to(dev)
tr = Trainer(net, opt, loss_fn, total_kimg=1, lr_rampup_kimg=0, lr_min_factor=1.0, device=dev, checkpoint_ticks=None)
tr.global_batch_size = 8
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(8, 72, 128, 256, generator=g, device=dev)
t = 0.3 * torch.randn(8, 69, 128, 256, generator=g, device=dev)
delta, idx = torch.full((8,), 0.6).pin_memory(), list(range(8))
for k in range(3):
    tr.train_step(x, t, idx, delta, 1000 * (k + 1))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for k in range(2):
        tr.train_step(x, t, idx, delta, 1000 * (k + 4))
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cuda_time_total", row_limit=28, max_name_column_width=60, max_src_column_width=90))
