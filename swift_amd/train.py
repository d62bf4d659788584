"""Training entry point -- ``python -m swift_amd.train experiment=... data.batch_size=... resume=... [finetune=multistep]``
(mirrors reference src/swift/train.py: Hydra-style overrides, run directory ``results/<experiment>/<HYDRA_RUN_ID>``
with ``.hydra/config.yaml``, resume / finetune hoisting, per-rank seeding, AdamW parameter groups, Trainer wiring).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import sys
from datetime import datetime
from glob import glob

import numpy as np
import torch
import torch.distributed as tdist
from torch.utils.data import DataLoader

from . import dist
from .config import Cfg, compose, instantiate, load_saved, to_yaml
from .data.samplers import DeltaBatchSampler, InfiniteSampler
from .generate import get_ckpt_num
from .models.swinv2 import SwinV2

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


def string_to_int(s: str) -> int:
    return int(hashlib.sha256(s.encode("utf-8")).hexdigest(), 16) % (1 << 31)


def resume_setup(cfg: Cfg):
    """train.py:44-99: continue from the latest checkpoint of run `cfg.resume`; finetune keys overwrite the old config."""
    if cfg.get("resume") is None:
        return cfg, None
    finetune = cfg.get("finetune")
    run_dir = str(cfg.resume)  # (a bare number would be taken for a file descriptor by os.path.isdir)
    if not os.path.isdir(run_dir):
        run_dir = os.path.join(os.path.dirname(os.getcwd()), str(cfg.resume))
    assert os.path.isdir(run_dir), FileNotFoundError(f"{run_dir} is not a directory")
    config = load_saved(os.path.join(run_dir, ".hydra", "config.yaml"))
    ckpts = sorted(glob(os.path.join(run_dir, "checkpoints", "checkpoint*.pt")), key=get_ckpt_num)
    assert ckpts, FileNotFoundError(f"No checkpoints in {os.path.join(run_dir, 'checkpoints')}")
    ckpt = ckpts[-1]
    if dist.get_rank() == 0:
        shutil.copytree(os.path.join(run_dir, ".hydra"), os.path.join(os.getcwd(), ".hydra"), dirs_exist_ok=True)
    if finetune is not None:
        for k, v in finetune.items():
            config[k] = v
        if config.finetune.get("name") == "multistep":
            config.trainer.total_kimg = get_ckpt_num(ckpt) + sum(iv["kimg"] for iv in config.finetune.get("intervals", []))
            config.trainer.lr_cosine_anneal = False
            config.trainer.checkpoint_ticks = 200
            config.trainer.val_ticks = 50
        if dist.get_rank() == 0:
            with open(os.path.join(os.getcwd(), ".hydra", "config.yaml"), "w") as f:
                f.write(to_yaml(config))
    dist.log0(f"Resuming from {ckpt}")
    return config, ckpt


def distill_setup(cfg, dataset):
    """Teacher for sCM distillation: the EMA weights of the latest checkpoint of run ``cfg.distill`` (train.py:100-132)."""
    if cfg.get("distill") is None:
        return None
    run_dir = cfg.distill
    config = load_saved(os.path.join(run_dir, ".hydra", "config.yaml"))
    ckpts = sorted(glob(os.path.join(run_dir, "checkpoints", "checkpoint*.pt")), key=get_ckpt_num)
    assert ckpts, FileNotFoundError(f"No checkpoints in {os.path.join(run_dir, 'checkpoints')}")
    dist.log0(f"Loading distillation model: {ckpts[-1]}")
    teacher = instantiate(config.precond, model_config=config.model, img_resolution=dataset.img_resolution,
                          img_channels=dataset.n_target_channels, condition_channels=dataset.n_condition_channels,
                          _recursive_=False, _convert_="object")
    teacher.eval().to(dist.get_torch_device())
    state = torch.load(ckpts[-1], map_location=dist.get_torch_device(), weights_only=True)
    teacher.load_state_dict(state["ema"])
    return teacher


def adamw_param_groups(net, weight_decay: float):
    """no weight decay for pos_embed and LayerNorm affine parameters (train.py:275-286)."""
    decay, no_decay = [], []
    for name, p in net.named_parameters():
        (no_decay if ("pos_embed" in name or ("norm" in name and "modulation" not in name)) else decay).append(p)
    return [{"params": decay, "weight_decay": weight_decay}, {"params": no_decay, "weight_decay": 0.0}]


def apply_distill_flag(cfg: Cfg) -> Cfg:
    """train.py:318-319: with a teacher run (``distill=<run dir>``) SCMLoss does consistency distillation -- the teacher's
    velocity replaces the analytic dx_t/dt -- instead of consistency training."""
    if str(cfg.loss._target_).endswith("SCMLoss") and cfg.get("distill") is not None:
        cfg.loss.distillation = True
    return cfg


def shared_run_id() -> str:
    """HYDRA_RUN_ID names the run directory and seeds the samplers (train.py:154): every rank must hold the same one.  The
    reference's launch scripts export it (scripts/aurora-general.sh:85); under a plain torchrun it is unset, and per-rank
    clocks can straddle a second -- so rank 0's id is broadcast once the process group is up."""
    rid = os.environ.get("HYDRA_RUN_ID")
    if dist.get_world_size() > 1:
        dist.setup_torch()
        box = [rid or datetime.now().strftime("%Y%m%d_%H%M%S")]
        tdist.broadcast_object_list(box, src=0)
        rid = box[0]
    rid = rid or datetime.now().strftime("%Y%m%d_%H%M%S")
    os.environ["HYDRA_RUN_ID"] = rid
    return rid


def main(overrides=None):
    argv = list(sys.argv[1:] if overrides is None else overrides)
    gpus = None
    for a in list(argv):  # additive: `--gpus N` / `--gpus=N` starts the N ranks from a bare `python -m swift_amd.train`
        if a.startswith("--gpus"):
            i = argv.index(a)
            gpus = int(a.split("=", 1)[1]) if "=" in a else int(argv[i + 1])
            del argv[i:i + (1 if "=" in a else 2)]
    if overrides is None and gpus:
        os.environ.setdefault("HYDRA_RUN_ID", datetime.now().strftime("%Y%m%d_%H%M%S"))  # inherited by every rank
        dist.maybe_launch_ranks(gpus, "swift_amd.train")
    shared_run_id()
    cfg = compose(CONFIG_DIR, "train", argv)
    run_dir = cfg.hydra.run.dir
    os.makedirs(os.path.join(run_dir, ".hydra"), exist_ok=True)
    os.chdir(run_dir)  # hydra.job.chdir: true
    dist.setup_torch(backend=cfg.system.torch.backend)
    if dist.get_rank() == 0:
        with open(os.path.join(".hydra", "config.yaml"), "w") as f:
            f.write(to_yaml({k: v for k, v in cfg.items() if k != "hydra"}))
    cfg, ckpt = resume_setup(cfg)
    if cfg.get("finetune") is not None and ckpt is None:
        dist.log0("ERROR: must have resume path to finetune")
        return None
    cfg.seed = cfg.seed + string_to_int(os.environ["HYDRA_RUN_ID"])
    np.random.seed((cfg.seed * dist.get_world_size() + dist.get_rank()) % (1 << 31))
    torch.manual_seed(np.random.randint(1 << 31))
    device = dist.get_torch_device()

    dataset = instantiate(cfg.data.dataset, _convert_="object")
    sampler = InfiniteSampler(dataset, rank=dist.get_rank(), num_replicas=dist.get_world_size(), shuffle=True, seed=cfg.seed)
    local_bs = cfg.data.batch_size // dist.get_world_size()
    common = dict(dataset=dataset, pin_memory=True, num_workers=cfg.data.data_workers,
                  prefetch_factor=(2 if cfg.data.data_workers > 0 else None), persistent_workers=cfg.data.data_workers > 0)
    if cfg.get("finetune") is not None:
        loader = DataLoader(batch_sampler=DeltaBatchSampler(sampler, local_bs, dataset.intervals, seed=cfg.seed), **common)
    else:
        loader = DataLoader(sampler=sampler, batch_size=local_bs, **common)

    # rollout validation set (train.py:224-260): the roll-out flavour of the training dataset's class, split "val"
    val_loader = None
    tgt = str(cfg.data.dataset._target_)
    if cfg.trainer.get("val_ticks") is not None and "era5" in tgt.lower():
        roll = tgt.replace("SyntheticERA5Dataset", "SyntheticERA5RollOutDataset").replace("ERA5Dataset", "ERA5RollOutDataset") \
            if "RollOut" not in tgt else tgt
        val_cfg = {k: v for k, v in cfg.data.dataset.items() if k not in ("intervals",)}
        val_cfg.update(_target_=roll, split="val", interval=cfg.trainer.val_target_interval)
        val_dataset = instantiate(val_cfg, _convert_="object")
        val_sampler = InfiniteSampler(dataset=val_dataset, rank=dist.get_rank(), num_replicas=dist.get_world_size(), shuffle=True,
                                      seed=cfg.seed)
        val_loader = DataLoader(dataset=val_dataset, sampler=val_sampler, batch_size=cfg.data.get("val_local_batch_size", 4),
                                pin_memory=True, num_workers=0)

    net = instantiate(cfg.precond, model_config=cfg.model, img_resolution=dataset.img_resolution,
                      img_channels=dataset.n_target_channels, condition_channels=dataset.n_condition_channels,
                      _recursive_=False, _convert_="object")
    net.train().requires_grad_(True).to(device)
    if dist.collectives_active():  # identical initial weights on every rank (DDP's initial broadcast)
        for p in net.parameters():
            tdist.broadcast(p.data, src=0)

    target = cfg.optimizer.get("_target_", "")
    params = net.parameters()
    if isinstance(net.model, SwinV2) and target in ("torch.optim.Adam", "torch.optim.AdamW"):
        params = adamw_param_groups(net, cfg.optimizer.weight_decay)
    elif target.endswith("MuonWithAuxAdam"):  # train.py:286-309: matrices of the transformer -> Muon, the rest -> Adam
        muon_p = [p for n, p in net.named_parameters() if p.ndim >= 2 and "transformer" in n]
        adam_p = [p for n, p in net.named_parameters() if not (p.ndim >= 2 and "transformer" in n)]
        params = [dict(params=muon_p, use_muon=True, lr=cfg.optimizer.lr, weight_decay=cfg.optimizer.weight_decay),
                  dict(params=adam_p, use_muon=False, lr=cfg.optimizer.adam_lr, betas=tuple(cfg.optimizer.adam_betas),
                       weight_decay=cfg.optimizer.adam_weight_decay, eps=cfg.optimizer.adam_eps)]
        optimizer = instantiate(dict(_target_=target), params, _convert_="object")
    if not target.endswith("MuonWithAuxAdam"):
        optimizer = instantiate(cfg.optimizer, params, _convert_="object")
    apply_distill_flag(cfg)
    loss_fn = instantiate(cfg.loss, dataset=dataset, _convert_="object").to(device)
    trainer_cfg = {k: v for k, v in cfg.trainer.items()}
    trainer = instantiate(trainer_cfg, net=net, optimizer=optimizer, loss_fn=loss_fn, amp_type=cfg.system.torch.amp_type,
                          ckpt=ckpt, flop_count=0, net_pretrained=distill_setup(cfg, dataset), solver_kwargs=cfg.get("solver"),
                          finetune_kwargs=cfg.get("finetune"))
    out = trainer.train(loader, val_loader)
    if tdist.is_initialized():
        tdist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
