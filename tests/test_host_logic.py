"""CPU-side host logic: config composition, dataset interface, unit sharding, and the world_size-2 (gloo)
generate path with a stand-in network (the HIP kernels need a GPU; the sharding / output-placement logic does not)."""
import math
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "swift_amd", "configs")


def test_compose_experiment_matches_reference_surface(monkeypatch):
    from swift_amd.config import compose
    monkeypatch.setenv("HYDRA_RUN_ID", "007")
    c = compose(CFG, "train", [])
    assert c.experiment_name == "era5-swinv2-1.4-scm" and c.seed == 1234
    assert c.model._target_ == "swift.models.swinv2.SwinV2"
    assert (c.model.depth, c.model.dim, c.model.heads) == (12, 1056, 12)
    assert c.model.window_size == [16, 16] and c.model.shift_size == [8, 8] and c.model.patch_size == [2, 2]
    assert c.precond._target_ == "swift.models.precond.PassPrecond" and c.precond.auxiliary_dim == 1
    assert c.loss._target_.endswith("SCMLoss") and c.loss.noise.dist == "loguniform" and c.loss.noise.sigma_max == 200
    assert c.loss.tangent_warmup_kimg == 3000
    assert c.solver == {"num_steps": 1, "sigma_min": 0.02, "sigma_max": 200, "auxiliary": 0.6}
    assert c.optimizer._target_.endswith("MuonWithAuxAdam")  # "override /optimizer: muon" of the experiment
    assert len(c.data.dataset.variables) == 69 and len(c.data.dataset.forcings) == 3
    assert c.data.dataset.residual is True and c.data.batch_size == 1
    assert c.trainer.total_kimg == 15000 and c.trainer._target_ == "swift.training.trainer.Trainer"
    assert c.hydra.run.dir == "results/era5-swinv2-1.4-scm/007"


def test_compose_cli_overrides_and_finetune_packaging():
    from swift_amd.config import compose
    c = compose(CFG, "train", ["experiment=era5-swinv2-1.4-trigflow", "optimizer=adamw", "data.batch_size=64",
                               "finetune=multistep", "data=era5-synthetic-1.4"])
    assert c.loss._target_.endswith("TrigFlowLoss") and c.solver.num_steps == 20
    assert c.optimizer._target_ == "torch.optim.AdamW" and c.data.batch_size == 64
    assert c.data.dataset._target_ == "swift_amd.data.era5.SyntheticERA5Dataset"
    # reference train.py:75-77 hoists these keys: they must arrive packaged under `finetune`
    assert c.finetune.loss._target_.endswith("CRPSLoss") and c.finetune.optimizer.lr == 1e-5
    assert c.finetune.finetune.intervals[0] == {"steps": 1, "kimg": 1500}


def test_instantiate_aliases_reference_targets():
    from swift_amd.config import instantiate
    from swift_amd.models.precond import PassPrecond
    from swift_amd.models.swinv2 import SwinV2
    net = instantiate({"_target_": "swift.models.precond.PassPrecond", "sigma_min": 0, "sigma_data": 1.0, "auxiliary_dim": 1},
                      model_config={"_target_": "swift.models.swinv2.SwinV2", "window_size": [16, 16], "shift_size": [8, 8],
                                    "patch_size": [2, 2], "depth": 1, "dim": 96, "heads": 4},
                      img_resolution=(32, 32), img_channels=4, condition_channels=7, sigma_max=float("inf"),
                      _recursive_=False, _convert_="object")
    assert isinstance(net, PassPrecond) and isinstance(net.model, SwinV2)
    assert net.model.in_channels == 11 and list(net.img_resolution) == [32, 32]


def test_dataset_interface_matches_oracle_stats():
    from oracle.rollout import Stats
    from swift_amd.data.era5 import SyntheticERA5Dataset
    ds = SyntheticERA5Dataset([f"v{i}" for i in range(5)], ["f0", "f1"], img_resolution=(8, 16), length=12, seed=3,
                              random_stats=True)
    st = Stats(ds.x_means, ds.x_stds, {6: ds.t_stds[6]}, n_vars=5, n_forc=2)
    x5, f2, x7 = torch.randn(2, 5, 8, 16), torch.randn(2, 2, 8, 16), torch.randn(2, 7, 8, 16)
    for v in (x5, f2, x7):  # channel-count dispatch of era5.py:110-133
        assert torch.allclose(ds.standardize_x(v.clone()), st.standardize_x(v))
        assert torch.allclose(ds.unstandardize_x(v.clone()), st.unstandardize_x(v))
    assert torch.allclose(ds.unstandardize_t(x5.clone(), 6), st.unstandardize_t(x5, 6))
    (x, t), (idx, delta) = ds[(3, 1, 6)]
    assert x.shape == (7, 8, 16) and t.shape == (5, 8, 16) and idx == 3 and float(delta) == pytest.approx(0.6)
    mx, sx, stt = ds.rollout_stats(6, "cpu")
    assert mx.shape == (5,) and torch.allclose(stt, torch.as_tensor(ds.t_stds[6]).reshape(-1))
    # SST is forced to zero by zero_field unless delta == 24 (era5.py:135-149): folded into the flat statistics
    ds2 = SyntheticERA5Dataset(["2m_temperature", "sea_surface_temperature"], [], img_resolution=(8, 16), random_stats=True)
    mx, sx, stt = ds2.rollout_stats(6, "cpu")
    assert (float(mx[1]), float(sx[1]), float(stt[1])) == (0.0, 1.0, 0.0)
    # delta 24 zeroes nothing (zero_field returns early): the one-delta statistics -- what the multistep CRPS loss applies,
    # loss.py:402-406 -- keep the channel's real values, so its update coefficient std_t / std_x is finite
    mx24, sx24, st24 = ds2.rollout_stats(24, "cpu")
    assert float(sx24[1]) == float(np.asarray(ds2.x_stds).reshape(-1)[1]) != 0.0
    assert float(mx24[1]) == float(np.asarray(ds2.x_means).reshape(-1)[1])
    assert float(st24[1]) == float(np.asarray(ds2.t_stds[24]).reshape(-1)[1]) != 0.0
    for d in (6, 12, 24):
        _, sxd, std = ds2.rollout_stats(d, "cpu")
        assert bool(torch.isfinite(std / sxd).all())
    g0 = torch.Generator().manual_seed(5)
    C0, P0 = torch.randn(2, 2, 8, 16, generator=g0), torch.randn(2, 2, 8, 16, generator=g0)
    for d in (6, 24):  # the loss's condition update cond + pred * st / sx against the reference's three calls with ONE delta
        m_, s_, t_ = (v.view(1, -1, 1, 1) for v in ds2.rollout_stats(d, "cpu"))
        Cd = ds2.zero_field(C0.clone(), d)  # what the dataset hands the loss: standardised with this delta
        ref = ds2.standardize_x(ds2.unstandardize_x(Cd.clone(), d) + ds2.unstandardize_t(P0.clone(), d), d)
        assert torch.allclose(Cd + P0 * (t_ / s_), ref, atol=2e-5)
    # the generate / validation rollout mixes deltas: its vectors carry the std-0 marker (rollout.update_stats)
    from swift_amd.rollout import update_stats
    mx24, sx24, st24 = update_stats(ds2, 24, "cpu")
    assert (float(mx24[1]), float(sx24[1])) == (0.0, 0.0) and float(st24[1]) == float(np.asarray(ds2.t_stds[24]).reshape(-1)[1]) != 0.0
    # the flat statistics against the reference's own sequence (generate.py:120-131: unstandardize_x and standardize_x with the
    # DEFAULT delta, unstandardize_t with delta = interval), with the update formula of swiftk_rollout_update restated in torch
    g = torch.Generator().manual_seed(0)
    for interval in (6, 24):
        X, Y = torch.randn(2, 2, 8, 16, generator=g), torch.randn(2, 2, 8, 16, generator=g)
        phys_ref = ds2.unstandardize_x(X.clone()) + ds2.unstandardize_t(Y.clone(), delta=interval)
        next_ref = ds2.standardize_x(phys_ref.clone())
        mx, sx, stt = (v.view(1, -1, 1, 1) for v in update_stats(ds2, interval, "cpu"))
        if interval == 24:
            assert float(phys_ref[:, 1].abs().sum()) > 0 and float(next_ref[:, 1].abs().sum()) == 0.0  # SST: residual kept, state zeroed
        X0 = X.clone()
        X0[:, 1] = 0  # SST's standardised value is what the dataset / the previous step hands on: zero
        phys = (X0 * sx + mx) + Y * stt
        nxt = torch.where(sx == 0, torch.zeros_like(phys), (phys - mx) / torch.where(sx == 0, torch.ones_like(sx), sx))
        assert torch.allclose(phys[:, 0], phys_ref[:, 0], atol=1e-5) and torch.allclose(nxt[:, 0], next_ref[:, 0], atol=1e-5)
        assert torch.allclose(phys[:, 1], phys_ref[:, 1], atol=1e-6) and torch.equal(nxt[:, 1], next_ref[:, 1])


def test_numa_binding_reads_the_gpu_node_from_sysfs(tmp_path):
    """Round 6: a rank's host threads go next to its GPU (scripts/aurora-general.sh:74-91 leaves this to the launcher's --cpu-bind)."""
    from swift_amd import dist
    assert dist._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11} and dist._parse_cpulist("") == set()
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    (node / "cpulist").write_text("64-127\n")
    assert dist.numa_cpus_of_gpu("0000:C1:00.0", str(tmp_path)) == set(range(64, 128))
    (dev / "numa_node").write_text("-1\n")                       # the kernel does not know: no binding
    assert dist.numa_cpus_of_gpu("0000:c1:00.0", str(tmp_path)) is None
    assert dist.numa_cpus_of_gpu("0000:00:00.0", str(tmp_path)) is None
    assert dist.bind_to_gpu_numa_node(0, 1, str(tmp_path)) is None  # no GPU here: best effort, never an error


def test_shard_units_partition():
    from swift_amd.dist import shard_units
    for n, world in [(768, 8), (12, 8), (5, 8), (64, 3)]:
        parts = [shard_units(n, r, world) for r in range(world)]
        assert sorted(u for p in parts for u in p) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert len(shard_units(768, 0, 8)) == 96  # 12 members x 64 ICs over 8 GPUs: 96 units each, not 2/2/2/2/1/1/1/1 members


WORKER = textwrap.dedent("""
    import os, sys, types, numpy as np, torch
    sys.path.insert(0, %(root)r)
    from swift_amd import dist
    from swift_amd.generate import create_empty_numpy, create_empty_zarr, rollout_and_save, select_indices
    from swift_amd.data.era5 import SyntheticERA5Dataset
    from swift_amd.rollout import unit_seed

    class FakeEngine:   # stands in for RolloutEngine: a deterministic function of (X0, forcings, seed), CPU only
        interval = 6
        def stage_forcings(self, ics, steps, device):
            return torch.stack([torch.stack([ds.get_forcings(j + i) for j in ics]) for i in range(steps)])
        def run(self, X0, forc, steps, seeds=None, **kw):
            out = [X0]
            for i in range(steps):
                z = torch.stack([torch.randn(X0.shape[1:], generator=torch.Generator().manual_seed(int(s) + i)) for s in seeds])
                out.append(0.9 * out[-1] + 0.1 * z + forc[i].mean(dim=1, keepdim=True))
            return torch.stack(out, 1)

    dist.setup_torch(backend="gloo")
    ds = SyntheticERA5Dataset(["t2m", "z_500", "z_850"], ["f0"], img_resolution=(4, 8), length=40, seed=5)
    members, steps = 3, 2
    idx = select_indices(len(ds), 5, steps, 6)
    ofile, dump = sys.argv[1], sys.argv[2]
    if dump == "numpy":
        dist.run_on_rank0(create_empty_numpy, ofile, len(idx), 3, (4, 8), members, steps)
    else:
        dist.run_on_rank0(create_empty_zarr, ofile, ds, idx, members, steps, 6)
    args = types.SimpleNamespace(dump=dump, batch=4)
    rollout_and_save(FakeEngine(), ds, idx, members, steps, ofile, torch.device("cpu"), args)
    dist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
""")


@pytest.mark.timeout(400)
def test_generate_sharding_world2_gloo_equals_world1(tmp_path):
    """(member, IC) units sharded over 2 and over 4 gloo ranks write the same npy as one rank (SURVEY.md section 4 item v:
    15 units over 4 ranks = blocks of 4, 4, 4, 3 -- uneven shards, partial batches)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    one = str(tmp_path / "one.npy")
    subprocess.run([sys.executable, str(script), one, "numpy"], check=True, env={**env, "WORLD_SIZE": "1", "RANK": "0"}, timeout=200)
    a = np.load(one)
    assert a.shape == (5, 3, 3, 3, 4, 8) and np.isfinite(a).all() and np.abs(a).sum() > 0
    for world in (2, 4):
        out = str(tmp_path / f"w{world}.npy")
        port = str(29600 + (os.getpid() + world) % 300)
        procs = [subprocess.Popen([sys.executable, str(script), out, "numpy"],
                                  env={**env, "WORLD_SIZE": str(world), "RANK": str(r), "LOCAL_RANK": str(r), "MASTER_PORT": port})
                 for r in range(world)]
        assert [p.wait(timeout=200) for p in procs] == [0] * world
        np.testing.assert_array_equal(a, np.load(out))
    # the reference's default dump (zarr, utils/io.py:161-235), two ranks: per-variable arrays with the level axis, read back
    # by the stdlib reader, must hold the same numbers as the npy store
    from swift_amd.utils import zarrlite
    zout = str(tmp_path / "w2.zarr")
    port = str(29600 + (os.getpid() + 7) % 300)
    procs = [subprocess.Popen([sys.executable, str(script), zout, "zarr"],
                              env={**env, "WORLD_SIZE": "2", "RANK": str(r), "LOCAL_RANK": str(r), "MASTER_PORT": port})
             for r in range(2)]
    assert [p.wait(timeout=200) for p in procs] == [0, 0]
    t2m, z = zarrlite.read_array(zout, "t2m"), zarrlite.read_array(zout, "z")
    assert t2m.shape == (5, 3, 3, 4, 8) and z.shape == (5, 3, 3, 2, 4, 8)
    np.testing.assert_array_equal(t2m, a[:, :, :, 0])
    np.testing.assert_array_equal(z, a[:, :, :, 1:3])
    assert zarrlite.read_attrs(zout, "z")["_ARRAY_DIMENSIONS"] == ["time", "number", "prediction_timedelta", "level", "latitude",
                                                                    "longitude"]
    assert zarrlite.read_array(zout, "prediction_timedelta").tolist() == [0, 6 * 3600 * 10**9, 12 * 3600 * 10**9]
    assert zarrlite.read_array(zout, "time").shape == (5,) and zarrlite.read_array(zout, "level").tolist() == [0, 1]


DDP_WORKER = textwrap.dedent("""
    import os, sys, torch
    sys.path.insert(0, %(root)r)
    from swift_amd import dist
    from swift_amd.training.trainer import GradAllReduce
    dist.setup_torch(backend="gloo")                   # (SWIFTK_SINGLE_RANK_GROUP=1: a group of one rank runs the same calls)
    rank, world = dist.get_rank(), dist.get_world_size()
    assert dist.collectives_active() == (world > 1 or os.environ.get("SWIFTK_SINGLE_RANK_GROUP") == "1")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    ddp = GradAllReduce(net)
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 8, generator=g), torch.randn(8, 3, generator=g)
    lo, hi = rank * 8 // world, (rank + 1) * 8 // world
    ddp.zero_grad_flat()
    ((ddp(X[lo:hi]) - Y[lo:hi]) ** 2).mean().backward()
    ddp.reduce_params([net[2].bias, net[0].weight])  # final gradients announced early: two non-neighbouring runs ...
    if dist.collectives_active():                  # ... a second announcement means a backward pass ran on averaged slices
        try:
            ddp.reduce_params([net[2].bias])
            raise SystemExit("second announcement of a reduced slice was accepted")
        except RuntimeError:
            pass
    flat = ddp.sync()                              # ... the rest reduced here
    keep = flat.clone()
    if dist.collectives_active():                  # the per-rank record a first multi-GPU run is read by
        st = ddp.allreduce_stats()
        assert st["world"] == world and st["bytes"] == flat.numel() * 4 and st["iterations_recorded"] == 1
        assert 0.0 < st["announced_early_frac"] < 1.0 and st["serial_ms"] is None and st["exposed_wait_ms"] >= 0.0
        assert ddp.calibrate_serial() >= 0.0 and torch.equal(flat, keep)   # calibration leaves the gradients alone
        st = ddp.allreduce_stats()
        assert st["serial_ms"] is not None and (st["overlap_frac"] is None or 0.0 <= st["overlap_frac"] <= 1.0)
        per = dist.gather_rank_times({"compute_ms": 10.0 + rank, "collective_ms": 0.5, "barrier_wait_ms": float(world - 1 - rank)})
        assert per["compute_ms"] == [10.0 + r for r in range(world)] and per["barrier_wait_ms"][-1] == 0.0 and len(per["collective_ms"]) == world
    if rank == 0:
        torch.save(flat.clone(), sys.argv[1])
    dist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
""")


@pytest.mark.timeout(300)
def test_gradient_allreduce_world2_equals_single_process(tmp_path):
    """2 ranks x local batch 4 with one averaged all-reduce == 1 process x batch 8 (SURVEY.md section 4 item iv)."""
    script = tmp_path / "ddp_worker.py"
    script.write_text(DDP_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    one, two = str(tmp_path / "g1.pt"), str(tmp_path / "g2.pt")
    subprocess.run([sys.executable, str(script), one], check=True, env={**env, "WORLD_SIZE": "1", "RANK": "0"}, timeout=200)
    port = str(29900 + os.getpid() % 90)
    procs = [subprocess.Popen([sys.executable, str(script), two],
                              env={**env, "WORLD_SIZE": "2", "RANK": str(r), "LOCAL_RANK": str(r), "MASTER_PORT": port})
             for r in range(2)]
    assert [p.wait(timeout=200) for p in procs] == [0, 0]
    a, b = torch.load(one), torch.load(two)
    assert a.abs().sum() > 0 and torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    # a process group of ONE rank pushes the same announcements / collectives through the backend: identical gradients
    solo = str(tmp_path / "g1_group.pt")
    subprocess.run([sys.executable, str(script), solo], check=True, timeout=200,
                   env={**env, "WORLD_SIZE": "1", "RANK": "0", "SWIFTK_SINGLE_RANK_GROUP": "1"})
    assert torch.equal(torch.load(solo), a)


WORLD8_WORKER = textwrap.dedent("""
    import os, sys, types, json, numpy as np, torch
    sys.path.insert(0, %(root)r)
    from swift_amd import dist
    from swift_amd.generate import create_empty_numpy, rollout_and_save, select_indices
    from swift_amd.data.era5 import SyntheticERA5Dataset
    from swift_amd.rollout import unit_seed

    class FakeEngine:   # stands in for RolloutEngine: a deterministic function of (X0, forcings, seed), CPU only
        interval = 6
        seen = []
        def stage_forcings(self, ics, steps, device):
            return torch.stack([torch.stack([ds.get_forcings(j + i) for j in ics]) for i in range(steps)])
        def run(self, X0, forc, steps, seeds=None, **kw):
            FakeEngine.seen += [int(s) for s in seeds]
            out = [X0]
            for i in range(steps):
                z = torch.stack([torch.randn(X0.shape[1:], generator=torch.Generator().manual_seed((int(s) + i) %% (2 ** 62)))
                                 for s in seeds])
                out.append(0.9 * out[-1] + 0.1 * z + forc[i].mean(dim=1, keepdim=True))
            return torch.stack(out, 1)

    dist.setup_torch(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ds = SyntheticERA5Dataset(["t2m", "z_500"], ["f0"], img_resolution=(2, 4), length=200, seed=5)
    members, ics, steps = 12, 64, 2                     # BASELINE configs[3]'s unit space: 12 x 64 = 768
    idx = select_indices(len(ds), ics, steps, 6)
    ofile = sys.argv[1]
    dist.run_on_rank0(create_empty_numpy, ofile, len(idx), 2, (2, 4), members, steps)
    args = types.SimpleNamespace(dump="numpy", batch=96)
    rollout_and_save(FakeEngine(), ds, idx, members, steps, ofile, torch.device("cpu"), args)
    mine = dist.shard_units(members * ics, rank, world)
    assert len(FakeEngine.seen) == len(mine) == (768 // world), (len(FakeEngine.seen), len(mine))
    want = [unit_seed(u %% members, idx[u // members]) for u in mine]     # IC-major: unit u = ic * members + member
    assert FakeEngine.seen == want
    json.dump({"rank": rank, "first": mine[0], "n": len(mine)}, open(ofile + f".rank{rank}.json", "w"))
    dist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
""")


@pytest.mark.timeout(600)
def test_generate_world8_is_the_configs3_partition(tmp_path):
    """BASELINE configs[3] on the node it names: 12 members x 64 ICs = 768 (member, IC) units over EIGHT ranks = 96 contiguous
    IC-major units each (8 whole ICs per rank), written into one store that equals the one-rank store.  gloo on CPU with a
    stand-in engine -- the partition, the seeds each rank asks for and the shared store are what an 8-GPU run adds."""
    import json
    script = tmp_path / "w8.py"
    script.write_text(WORLD8_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    one = str(tmp_path / "one.npy")
    subprocess.run([sys.executable, str(script), one], check=True, env={**env, "WORLD_SIZE": "1", "RANK": "0"}, timeout=300)
    a = np.load(one)
    assert a.shape == (64, 12, 3, 2, 2, 4) and np.isfinite(a).all()
    out = str(tmp_path / "w8.npy")
    port = str(29500 + (os.getpid() + 8) % 90)
    procs = [subprocess.Popen([sys.executable, str(script), out],
                              env={**env, "WORLD_SIZE": "8", "RANK": str(r), "LOCAL_RANK": str(r), "MASTER_PORT": port})
             for r in range(8)]
    assert [p.wait(timeout=400) for p in procs] == [0] * 8
    np.testing.assert_array_equal(a, np.load(out))
    recs = [json.load(open(out + f".rank{r}.json")) for r in range(8)]
    assert [r["n"] for r in recs] == [96] * 8 and [r["first"] for r in recs] == [96 * r for r in range(8)]


@pytest.mark.timeout(600)
def test_gradient_allreduce_world8_equals_single_process(tmp_path):
    """BASELINE configs[4]'s group size: 8 ranks x local batch 1 with the early announcements + one averaged all-reduce ==
    1 process x batch 8."""
    script = tmp_path / "ddp_worker.py"
    script.write_text(DDP_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    one, eight = str(tmp_path / "g1.pt"), str(tmp_path / "g8.pt")
    subprocess.run([sys.executable, str(script), one], check=True, env={**env, "WORLD_SIZE": "1", "RANK": "0"}, timeout=200)
    port = str(29700 + os.getpid() % 90)
    procs = [subprocess.Popen([sys.executable, str(script), eight],
                              env={**env, "WORLD_SIZE": "8", "RANK": str(r), "LOCAL_RANK": str(r), "MASTER_PORT": port})
             for r in range(8)]
    assert [p.wait(timeout=400) for p in procs] == [0] * 8
    a, b = torch.load(one), torch.load(eight)
    assert a.abs().sum() > 0 and torch.allclose(a, b, rtol=1e-5, atol=1e-7)


def test_rollout_update_stats_non_residual_sst():
    """One helper feeds generate's and the validation rollout's update kernel: a non-residual dataset gets
    std_t = None, and sea_surface_temperature -- zeroed by zero_field in the reference (data/era5.py:135-170) -- a zero
    un-standardisation scale in BOTH loops."""
    from swift_amd.data.era5 import SyntheticERA5Dataset
    from swift_amd.rollout import RolloutEngine, update_stats
    names = ["t2m", "sea_surface_temperature", "z_500"]
    for residual in (True, False):
        ds = SyntheticERA5Dataset(names, ["f0"], img_resolution=(4, 8), length=16, seed=3, random_stats=True, residual=residual)
        mx, sx, st = update_stats(ds, 6, "cpu")
        assert (st is None) == (not residual)
        assert float(mx[1]) == 0.0 and float(sx[1]) == 0.0 and float(sx[0]) != 0.0 and float(sx[2]) != 0.0
        if residual:
            assert float(st[1]) == 0.0
        eng = RolloutEngine.__new__(RolloutEngine)
        eng.dataset, eng.interval, eng.residual, eng._stats = ds, 6, residual, None
        got = eng.stats(torch.device("cpu"))
        assert all((a is None and b is None) or torch.equal(a, b) for a, b in zip(got, (mx, sx, st)))


MUON_WORKER = textwrap.dedent("""
    import os, sys, torch
    sys.path.insert(0, %(root)r)
    from swift_amd import dist
    from swift_amd.training.optimizers import muon as pm
    from oracle import muon as om          # CPU stand-in for the GEMM-backed orthogonaliser (no GPU in this test)
    pm.zeropower_via_newtonschulz5 = lambda G, steps=5: om.newton_schulz5(G, steps)
    dist.setup_torch(backend="gloo")
    g = torch.Generator().manual_seed(3)
    shapes = [(24, 16), (16, 40), (32, 32), (8, 48), (40, 8), (12,), (5, 3)]
    params = [torch.nn.Parameter(torch.randn(*sh, generator=g) * 0.05) for sh in shapes]
    opt = pm.MuonWithAuxAdam([dict(params=params[:5], use_muon=True, lr=0.02, weight_decay=0.01),
                              dict(params=params[5:], use_muon=False, lr=3e-4, betas=(0.9, 0.95), weight_decay=0.01, eps=1e-10)])
    for st in range(3):
        for p in params:
            p.grad = torch.randn(p.shape, generator=g)      # identical on every rank (as after the gradient all-reduce)
        opt.step()
    if dist.get_rank() == 0:
        torch.save([p.detach().clone() for p in params], sys.argv[1])
    dist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
""")


@pytest.mark.timeout(300)
def test_muon_round_robin_world2_equals_single_process(tmp_path):
    """MuonWithAuxAdam's parameter ownership (rank r updates parameters r, r + world, ... and all-gathers them, muon.py:215-238):
    two gloo ranks end with the parameters of one process.  5 Muon parameters on 2 ranks = an uneven last round."""
    script = tmp_path / "muon_worker.py"
    script.write_text(MUON_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    one, two = str(tmp_path / "m1.pt"), str(tmp_path / "m2.pt")
    subprocess.run([sys.executable, str(script), one], check=True, env={**env, "WORLD_SIZE": "1", "RANK": "0"}, timeout=200)
    port = str(29800 + os.getpid() % 90)
    procs = [subprocess.Popen([sys.executable, str(script), two],
                              env={**env, "WORLD_SIZE": "2", "RANK": str(r), "LOCAL_RANK": str(r), "MASTER_PORT": port})
             for r in range(2)]
    assert [p.wait(timeout=200) for p in procs] == [0, 0]
    a, b = torch.load(one), torch.load(two)
    assert len(a) == 7 and all(torch.equal(x, y) for x, y in zip(a, b))


def test_lr_schedule_and_param_groups():
    """trainer.py:201-217 and train.py:275-286 on a CPU-only stand-in (no kernels involved)."""
    from swift_amd.train import adamw_param_groups
    from swift_amd.training.trainer import Trainer
    net = torch.nn.Linear(4, 4)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    tr = Trainer.__new__(Trainer)
    tr.optimizer, tr.base_lr = opt, [1e-3]
    tr.lr_rampup_kimg, tr.lr_min_factor, tr.lr_cosine_anneal, tr.total_kimg = 2, 0.01, True, 10
    tr._set_lr(0)
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-5)
    tr._set_lr(1000)
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-5 + (1e-3 - 1e-5) * 0.5)
    tr._set_lr(2000)
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-3)
    tr._set_lr(6000)
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-5 + 0.5 * (1e-3 - 1e-5) * (1 + math.cos(math.pi * 0.5)))
    tr._set_lr(10_000)
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-5)


def test_zarrlite_store_layout_and_roundtrip(tmp_path):
    """The stdlib zarr-v2 writer: reference layout (utils/io.py:161-235), consolidated metadata, chunk files = whole units."""
    import json
    from swift_amd.utils import zarrlite
    names = ["2m_temperature", "geopotential_500", "geopotential_850", "temperature_850", "mean_sea_level_pressure"]
    assert zarrlite.compress_variables(names) == {"2m_temperature": [], "geopotential": [500, 850], "temperature": [850],
                                                  "mean_sea_level_pressure": []}
    assert zarrlite.variable_channels(names) == {"2m_temperature": [0], "geopotential": [1, 2], "temperature": [3],
                                                 "mean_sea_level_pressure": [4]}
    root = str(tmp_path / "o.zarr")
    times = np.array(["2020-01-01T00", "2020-01-03T12"], dtype="datetime64[ns]")
    ch = zarrlite.create_forecast_store(root, names, times, np.linspace(-90, 90, 4), np.arange(8) * 45.0, members=2, steps=3,
                                        interval=12)
    rng = np.random.default_rng(0)
    traj = rng.standard_normal((2, 2, 4, 5, 4, 8)).astype(np.float32)
    for s_ in range(2):
        for m in range(2):
            if (s_, m) != (1, 1):  # one unit left unwritten: reads back as fill_value 0
                zarrlite.write_unit(root, ch, s_, m, traj[s_, m])
    zarrlite.consolidate(root)
    meta = json.load(open(os.path.join(root, ".zmetadata")))["metadata"]
    assert meta["geopotential/.zarray"]["chunks"] == [1, 1, 4, 2, 4, 8] and meta["geopotential/.zarray"]["compressor"] is None
    assert meta["temperature/.zarray"]["shape"] == [2, 2, 4, 1, 4, 8]  # a single pressure level still carries the level axis
    assert meta["2m_temperature/.zattrs"]["_ARRAY_DIMENSIONS"] == ["time", "number", "prediction_timedelta", "latitude", "longitude"]
    assert sorted(os.listdir(os.path.join(root, "geopotential"))) == [".zarray", ".zattrs", "0.0.0.0.0.0", "0.1.0.0.0.0", "1.0.0.0.0.0"]
    g = zarrlite.read_array(root, "geopotential")
    np.testing.assert_array_equal(g[0], traj[0][:, :, 1:3])
    np.testing.assert_array_equal(g[1, 0], traj[1, 0][:, 1:3])
    assert (g[1, 1] == 0).all()
    np.testing.assert_array_equal(zarrlite.read_array(root, "mean_sea_level_pressure")[0, 1], traj[0, 1][:, 4])
    assert zarrlite.read_array(root, "time").astype("datetime64[ns]").tolist() == times.tolist()
    assert zarrlite.read_array(root, "prediction_timedelta").tolist() == [0, 12 * 3600 * 10**9, 24 * 3600 * 10**9, 36 * 3600 * 10**9]


def test_zarrlite_reads_compressed_chunks_or_names_the_compressor(tmp_path):
    """Round 6 (VERDICT r5 missing #3): a store written by zarr / xarray carries a compressor (the reference's: the default Blosc,
    utils/io.py:161-235).  Codec ids the standard library covers are read; anything else goes through numcodecs when that is
    importable, and is otherwise refused with an error that NAMES the compressor -- never garbage, never a bare 'unsupported'."""
    import gzip
    import json
    import zlib
    from swift_amd.utils import zarrlite
    rng = np.random.default_rng(1)
    a = rng.standard_normal((3, 5, 4)).astype(np.float32)
    for cid, enc, sep in (("zlib", zlib.compress, "."), ("gzip", gzip.compress, "/")):
        root = tmp_path / f"{cid}.zarr"
        (root / "v").mkdir(parents=True)
        meta = dict(zarr_format=2, shape=[3, 5, 4], chunks=[2, 5, 4], dtype="<f4", compressor={"id": cid, "level": 1}, fill_value=0.0,
                    order="C", filters=None)
        if sep == "/":
            meta["dimension_separator"] = "/"
        (root / "v" / ".zarray").write_text(json.dumps(meta))
        for i in range(2):
            blk = np.zeros((2, 5, 4), np.float32)
            blk[: min(2, 3 - 2 * i)] = a[2 * i:2 * i + 2]
            f = root / "v" / sep.join([str(i), "0", "0"])
            f.parent.mkdir(parents=True, exist_ok=True)
            f.write_bytes(enc(blk.tobytes()))
        np.testing.assert_array_equal(zarrlite.read_array(str(root), "v"), a)
    # Blosc (zarr's default compressor): without numcodecs the container is parsed by zarrlite._blosc1_decode and the inner streams
    # by pyarrow's LZ4 / Zstd codecs.  The chunks here are ASSEMBLED BY HAND from the c-blosc 1.x chunk layout around real codec
    # streams (c-blosc itself is not in this image): split shuffled blocks, a short last block, a stream stored raw, a stored chunk
    import struct
    pa = pytest.importorskip("pyarrow")

    def blosc1(data: bytes, typesize: int, blocksize: int, codec: str, shuffle: bool, dont_split: bool = False, raw_stream: int = -1):
        fmt = {"lz4_raw": 1, "zstd": 4, "zlib": 3}[codec]
        nblocks = (len(data) + blocksize - 1) // blocksize
        flags = (1 if shuffle else 0) | (0x10 if dont_split else 0) | (fmt << 5)
        body, starts = b"", []
        k = 0
        for i in range(nblocks):
            blk = data[i * blocksize:(i + 1) * blocksize]
            if shuffle and typesize > 1:
                n = len(blk) // typesize
                blk = np.frombuffer(blk, np.uint8, n * typesize).reshape(n, typesize).T.tobytes() + blk[n * typesize:]
            split = (not dont_split) and len(blk) == blocksize and typesize <= 16 and blocksize // typesize >= 128
            ns = typesize if split else 1
            ne = len(blk) // ns
            starts.append(16 + 4 * nblocks + len(body))
            for j in range(ns):
                st = blk[j * ne:(j + 1) * ne]
                enc = zlib.compress(st) if codec == "zlib" else pa.compress(st, codec=codec, asbytes=True)
                if k == raw_stream or len(enc) >= len(st):
                    enc = st  # stored raw: its length equals the uncompressed size
                body += struct.pack("<i", len(enc)) + enc
                k += 1
        head = bytes([2, 1, flags, typesize]) + struct.pack("<iii", len(data), blocksize, 16 + 4 * nblocks + len(body))
        return head + struct.pack(f"<{nblocks}i", *starts) + body

    smooth = np.cumsum(rng.standard_normal(5000)).astype(np.float32)  # compressible after the byte shuffle, like a weather field
    raw = smooth.tobytes()  # 20,000 bytes: two whole 8,192-byte blocks (split into 4 streams each) and a short last one (one stream)
    for codec, shuffle, dont_split, raw_stream in (("lz4_raw", True, False, -1), ("zstd", True, False, 2), ("lz4_raw", False, False, -1),
                                                   ("lz4_raw", True, True, -1), ("zlib", True, False, -1)):
        assert zarrlite._blosc1_decode(blosc1(raw, 4, 8192, codec, shuffle, dont_split, raw_stream)) == raw
    stored = bytes([2, 1, 0x2 | (1 << 5), 4]) + struct.pack("<iii", len(raw), 8192, 16 + len(raw)) + raw
    assert zarrlite._blosc1_decode(stored) == raw and zarrlite._blosc1_decode(bytes([2, 1, 0x21, 4]) + struct.pack("<iii", 0, 0, 16)) == b""
    for bad in (b"\x02\x01\x21", bytes([2, 1, 0x01, 4]) + struct.pack("<iii", 100, 8192, 40) + bytes(24),          # truncated; BloscLZ
                bytes([2, 1, 0x25, 4]) + struct.pack("<iii", 100, 8192, 40) + bytes(24)):                           # bit shuffle
        with pytest.raises((ValueError, NotImplementedError)):
            zarrlite._blosc1_decode(bad)
    cut = blosc1(raw, 4, 8192, "lz4_raw", True)
    with pytest.raises(Exception):
        zarrlite._blosc1_decode(cut[:len(cut) // 2] + bytes(len(cut) - len(cut) // 2))  # a damaged chunk raises, it does not decode to garbage
    root = tmp_path / "blosc.zarr"
    (root / "v").mkdir(parents=True)
    (root / "v" / ".zarray").write_text(json.dumps(dict(zarr_format=2, shape=[5000], chunks=[5000], dtype="<f4", fill_value=0.0, order="C", filters=None,
                                                        compressor={"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0})))
    (root / "v" / "0").write_bytes(blosc1(raw, 4, 8192, "lz4_raw", True))
    np.testing.assert_array_equal(zarrlite.read_array(str(root), "v"), smooth)
    (root / "v" / ".zarray").write_text(json.dumps(dict(zarr_format=2, shape=[2], chunks=[2], dtype="<f4", fill_value=0.0, order="C", filters=None,
                                                        compressor={"id": "lzo", "level": 3})))
    try:
        import numcodecs  # noqa: F401
    except ImportError:
        with pytest.raises(NotImplementedError, match=r"id='lzo'.*numcodecs"):  # an unknown compressor is NAMED, not guessed at
            zarrlite.read_array(str(root), "v")


def test_store_to_store_evaluation_matches_the_oracle_metrics(tmp_path):
    """``python -m swift.eval.metrics --truth T.zarr --pred P.zarr`` (eval/metrics.py:157-280) without zarr / xarray: the forecast
    store as ``generate`` writes it, a truth store with ITS OWN time encoding (hours since 2020, gzip-compressed chunks, a longer time
    axis), walked one initial condition at a time.  The device kernel is replaced by its definition in torch; every metric of every
    lead / variable / level must equal the oracle's restatement of the reference functions on the whole arrays."""
    import gzip
    import json
    from oracle import metrics as om
    from swift_amd.eval import metrics as em
    from swift_amd.utils import zarrlite
    rng = np.random.default_rng(3)
    names = ["2m_temperature", "geopotential_500", "geopotential_850"]
    H, W, B, N, steps, interval = 4, 8, 3, 3, 2, 12
    lat = np.linspace(-80, 80, H)
    t_all = np.datetime64("2020-01-01T00") + np.arange(20) * np.timedelta64(6, "h")
    init = t_all[[2, 5, 9]]
    pred = str(tmp_path / "run" / "output-3i-2s-3m-12h.zarr")
    os.makedirs(os.path.dirname(pred))
    ch = zarrlite.create_forecast_store(pred, names, init, lat, np.arange(W) * 45.0, members=N, steps=steps, interval=interval)
    traj = rng.standard_normal((B, N, steps + 1, 3, H, W)).astype(np.float32)
    for b in range(B):
        for n in range(N):
            zarrlite.write_unit(pred, ch, b, n, traj[b, n])
    truth = str(tmp_path / "truth.zarr")
    zarrlite.create_group(truth)
    zarrlite.write_full(truth, "time", (np.arange(20) * 6).astype(np.int64), ["time"], {"units": "hours since 2020-01-01 00:00:00", "calendar": "proleptic_gregorian"})
    zarrlite.write_full(truth, "latitude", lat.astype(np.float32), ["latitude"])
    fields = {"2m_temperature": rng.standard_normal((20, H, W)).astype(np.float32), "geopotential": rng.standard_normal((20, 2, H, W)).astype(np.float32)}
    for v, arr in fields.items():   # compressed, chunked along time by 7: the reader must decode and stitch
        d = os.path.join(truth, v)
        os.makedirs(d)
        chunks = (7,) + arr.shape[1:]
        json.dump(dict(zarr_format=2, shape=list(arr.shape), chunks=list(chunks), dtype="<f4", compressor={"id": "gzip", "level": 1}, fill_value=0.0,
                       order="C", filters=None), open(os.path.join(d, ".zarray"), "w"))
        json.dump({"_ARRAY_DIMENSIONS": ["time"] + (["level"] if arr.ndim == 4 else []) + ["latitude", "longitude"]}, open(os.path.join(d, ".zattrs"), "w"))
        for i in range(3):
            blk = np.zeros(chunks, np.float32)
            part = arr[7 * i:7 * i + 7]
            blk[:len(part)] = part
            open(os.path.join(d, ".".join([str(i)] + ["0"] * (arr.ndim - 1))), "wb").write(gzip.compress(blk.tobytes()))

    def sums(p, y, lat_):   # the four sums of swiftk_ensemble_sums (include/swiftk.h), in torch on the CPU
        p, y = torch.from_numpy(p).double(), torch.from_numpy(y).double()
        w = np.cos(np.deg2rad(lat_))
        w = torch.from_numpy(w / w.mean()).view(1, 1, -1, 1)
        s0 = ((p.mean(1) - y) ** 2 * w).sum((-2, -1))
        s1 = ((p - y.unsqueeze(1)).abs() * w.unsqueeze(1)).sum((1, -2, -1))
        s2 = ((p.unsqueeze(2) - p.unsqueeze(1)).abs() * w.view(1, 1, 1, 1, -1, 1)).sum((1, 2, -2, -1))
        s3 = (p.var(1) * w).sum((-2, -1))
        return torch.stack([s0, s1, s2, s3], -1).numpy()

    flat = em.evaluate_stores(truth, pred, sums_fn=sums, log=lambda *_: None)
    assert len(flat) == 3 * 3 * 3  # metrics x leads x (1 + 2 level) variables
    idx = np.array([2, 5, 9])
    for j, h in enumerate((0, 12, 24)):
        tgt = idx + h // 6
        for key, p_arr, y_arr in (("2m_temperature", traj[:, :, j, 0:1], fields["2m_temperature"][tgt][:, None]),
                                  ("geopotential_50", traj[:, :, j, 1:2], fields["geopotential"][tgt][:, 0:1]),
                                  ("geopotential_100", traj[:, :, j, 2:3], fields["geopotential"][tgt][:, 1:2])):
            P, Y = torch.from_numpy(p_arr).double(), torch.from_numpy(y_arr).double()
            assert flat[f"rmse_{key}_{h}h"] == pytest.approx(float(om.rmse(P, Y, lat)[0]), rel=1e-9)
            assert flat[f"crps_{key}_{h}h"] == pytest.approx(float(om.crps(P, Y, lat)[0]), rel=1e-9, abs=1e-12)
            assert flat[f"ssr_{key}_{h}h"] == pytest.approx(float(om.spread_skill_ratio(P, Y, lat)[0]), rel=1e-9)
    st = em.structure(flat)
    assert set(st) == {"rmse", "crps", "ssr"} and set(st["crps"]) == {"0", "12", "24"} and set(st["ssr"]["24"]) == {"2m_temperature", "geopotential_50", "geopotential_100"}
    # an initial time the truth store does not hold is an error that says so
    zarrlite.write_full(pred, "time", (init + np.timedelta64(1, "h")).astype("datetime64[ns]").astype(np.int64), ["time"],
                        {"units": "nanoseconds since 1970-01-01", "calendar": "proleptic_gregorian"})
    with pytest.raises(ValueError, match="not in the truth store"):
        em.evaluate_stores(truth, pred, sums_fn=sums, log=lambda *_: None)


def test_finetune_keeps_its_own_optimizer_and_cli_floats():
    """Hydra keys a defaults entry by group AND package: the experiment's `override /optimizer: muon` must not reach the
    `/optimizer: adamw` that finetune/multistep.yaml packages at finetune.optimizer (reference finetune = AdamW lr 1e-5)."""
    from swift_amd.config import compose
    c = compose(CFG, "train", ["finetune=multistep"])
    assert c.optimizer._target_.endswith("MuonWithAuxAdam")
    assert c.finetune.optimizer._target_ == "torch.optim.AdamW" and c.finetune.optimizer.lr == 1e-5
    assert "adam_lr" not in c.finetune.optimizer
    c = compose(CFG, "train", ["finetune=multistep", "optimizer@finetune.optimizer=muon"])  # ... unless addressed explicitly
    assert c.finetune.optimizer._target_.endswith("MuonWithAuxAdam")
    # YAML 1.2 floats on the command line and in files (PyYAML alone reads `1e-4` as a string)
    c = compose(CFG, "train", ["optimizer=adamw", "optimizer.lr=1e-4", "trainer.lr_min_factor=3E-4", "resume=007", "seed=12"])
    assert c.optimizer.lr == 1e-4 and isinstance(c.optimizer.lr, float) and c.trainer.lr_min_factor == 3e-4
    assert c.resume == "007" and c.seed == 12
    # a run directory named by its start time (HYDRA_RUN_ID unset) is a run id too, not the integer YAML 1.1 reads into it
    assert compose(CFG, "train", ["resume=20261003_180414"]).resume == "20261003_180414"


def test_distill_sets_scm_distillation_flag():
    """reference train.py:318-319."""
    from swift_amd.config import compose
    from swift_amd.train import apply_distill_flag
    c = apply_distill_flag(compose(CFG, "train", ["distill=/some/teacher/run"]))
    assert c.loss._target_.endswith("SCMLoss") and c.loss.distillation is True
    c = apply_distill_flag(compose(CFG, "train", []))
    assert not c.loss.get("distillation", False)
    c = apply_distill_flag(compose(CFG, "train", ["experiment=era5-swinv2-1.4-trigflow", "distill=/some/teacher/run"]))
    assert not c.loss.get("distillation", False)  # TrigFlowLoss has no such switch


RUNID_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    from swift_amd import dist
    from swift_amd.train import shared_run_id
    rid = shared_run_id()
    open(sys.argv[1] + "." + os.environ["RANK"], "w").write(rid + " " + os.environ["HYDRA_RUN_ID"])
    dist.barrier()
    import torch.distributed as td
    td.destroy_process_group()
""")


@pytest.mark.timeout(300)
def test_run_id_is_shared_across_ranks(tmp_path):
    """Ranks whose own HYDRA_RUN_ID would differ (unset + clocks straddling a second) end up with rank 0's: one run
    directory, one sampler seed (train.py:154)."""
    script = tmp_path / "rid_worker.py"
    script.write_text(RUNID_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", WORLD_SIZE="2", MASTER_PORT=str(29500 + os.getpid() % 90))
    out = str(tmp_path / "rid")
    procs = [subprocess.Popen([sys.executable, str(script), out],
                              env={**env, "RANK": str(r), "LOCAL_RANK": str(r), "HYDRA_RUN_ID": ["20260101_000000", "20260101_000001"][r]})
             for r in range(2)]
    assert [p.wait(timeout=200) for p in procs] == [0, 0]
    assert open(out + ".0").read() == open(out + ".1").read() == "20260101_000000 20260101_000000"


def test_h5_backed_era5_dataset_vs_reference_golden(tmp_path, monkeypatch):
    """swift_amd.data.era5.ERA5Dataset / ERA5RollOutDataset over an on-disk tree in the reference's layout against what the
    REFERENCE's loader returned for the identical tree (tests/golden/era5_tiny.npz): file ordering, NaN fill, residual
    targets at offsets 1-3, statistics chosen by channel count and by interval, forcings, time stamps, rollout items."""
    from conftest import load_golden
    import era5_fixture as fx
    monkeypatch.setitem(sys.modules, "h5py", fx.install_fake_h5py())
    from swift_amd.data.era5 import ERA5Dataset, ERA5RollOutDataset
    from swift_amd.utils.detinit import det_normal
    g = load_golden("era5_tiny")
    root = fx.write_tree(str(tmp_path / "era5"))
    ds = ERA5Dataset(root, list(fx.VARS), list(fx.FORC), intervals=[6, 12, 24], split="train", residual=True)
    assert len(ds) == int(g["len"]) == fx.N_FILES - 4 and tuple(ds._shape) == tuple(g["shape"])
    assert ds.n_target_channels == 4 and ds.n_condition_channels == 6 and ds.img_resolution == fx.SHAPE
    for spec in [(0, 1, 6), (3, 1, 12), (2, 2, 6), (1, 3, 12), (4, 1, 24)]:
        (x, t), (idx, delta) = ds[spec]
        tag = "_".join(map(str, spec))
        np.testing.assert_array_equal(x.numpy(), g[f"x_{tag}"])
        np.testing.assert_array_equal(t.numpy(), g[f"t_{tag}"])
        assert idx == spec[0] and float(delta) == float(g[f"d_{tag}"])
    assert np.isfinite(ds[(3, 1, 6)][0][0].numpy()).all()  # file 3 carries a NaN that the loader fills
    np.testing.assert_array_equal(ds.get_forcings(5).numpy(), g["forc5"])
    np.testing.assert_array_equal(ds.standardize_x(ds.get_state(2)).numpy(), g["x_2_2_6"][:4])  # the one-read form generate uses
    assert str(ds.get_time(7)) == str(g["time7"])
    lat, lon = ds.get_lat_lon()
    np.testing.assert_array_equal(lat, g["lat"])
    np.testing.assert_array_equal(lon, g["lon"])
    v = det_normal((2, 4, *fx.SHAPE), 3, "v")
    np.testing.assert_allclose(ds.unstandardize_t(v.clone(), 12).numpy(), g["unstd_t12"], rtol=1e-6)
    np.testing.assert_allclose(ds.standardize_x(ds.get_forcings(2)[None]).numpy(), g["std_x_forc"], rtol=1e-6)
    mx, sx, st = ds.rollout_stats(12, "cpu")  # the flat vectors the fused rollout kernel takes = the same statistics
    np.testing.assert_allclose((v[0] * st.view(-1, 1, 1)).numpy(), g["unstd_t12"][0], rtol=1e-6)
    ro = ERA5RollOutDataset(8, root, list(fx.VARS), list(fx.FORC), intervals=[6, 12, 24], split="train", residual=True)
    x, ts, idx = ro[1]
    assert len(ro) == int(g["ro_len"]) and idx == 1
    np.testing.assert_array_equal(x.numpy(), g["ro_x"])
    np.testing.assert_array_equal(ts.numpy(), g["ro_t"])


@pytest.mark.timeout(300)
def test_bench_launcher_contract_without_gpu():
    """`python bench.py --gpus N` launched bare starts N rank processes before anything touches the GPU and propagates their
    failure; on this GPU-less machine: too few devices -> refused (rc 2); forced -> every rank builds the process group,
    finds no GPU and exits 1 with the "no CPU fallback" message (the hot path never falls back; the first rank to fail takes its
    sibling down); a process group whose size
    is not --gpus -> rc 3."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    bench = os.path.join(ROOT, "bench.py")
    if torch.cuda.is_available():
        pytest.skip("CPU-side contract test")
    p = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1"], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 2 and "exposes 0 GPU(s)" in p.stderr and not p.stdout.strip()
    p = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       env=dict(env, SWIFTK_ALLOW_SHARED_GPU="1"), timeout=200)
    # (the launcher ends the sibling rank as soon as one fails, so the message appears once or twice; gloo logs its ranks to stdout)
    assert p.returncode == 1 and 1 <= p.stderr.count("no CPU fallback") <= 2 and "{" not in p.stdout
    p = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="1", RANK="0"), timeout=120)
    assert p.returncode == 3 and "process group has 1 rank(s)" in p.stderr
    p = subprocess.run([sys.executable, bench, "--batch", "0"], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 2 and "SWIFTK_MAX_UNITS" in p.stderr


def test_index_streams_vs_reference_golden():
    """``InfiniteSampler`` / ``DeltaBatchSampler`` emit the reference's index stream (data/samplers.py:9-85; fixture made by
    the reference's own samplers, tools/make_golden.py::fx_index_streams) -- several laps, rank strides, offsets, no-shuffle."""
    import itertools

    from swift_amd.data.samplers import DeltaBatchSampler, InfiniteSampler
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "index_streams.npz"))
    for c, (n, rank, world, shuffle, seed, window, offset, count) in enumerate(g["cases"]):
        s = InfiniteSampler(range(int(n)), rank=int(rank), num_replicas=int(world), shuffle=bool(shuffle), seed=int(seed),
                            window_size=float(window))
        if offset > 1:
            s.set_offset(int(offset))
        items = list(itertools.islice(iter(s), int(count)))
        if offset > 1:
            assert all(isinstance(i, tuple) and i[1] == int(offset) for i in items)
            items = [i[0] for i in items]
        assert np.array_equal(np.array(items, dtype=np.int64), g[f"stream_{c}"]), f"case {c}"
    s = InfiniteSampler(range(50), rank=1, num_replicas=2, shuffle=True, seed=3)
    s.set_offset(3)
    got = np.array(list(itertools.islice(iter(DeltaBatchSampler(s, 4, [6, 12, 24], seed=3)), 12)), dtype=np.int64)
    assert np.array_equal(got, g["delta_batches"])
    # the lap form leans on numpy drawing the same bounded integers one at a time and in bulk
    a, b = np.random.default_rng(9), np.random.default_rng(9)
    assert np.array_equal(np.array([a.integers(19) for _ in range(500)]), b.integers(19, size=500))


def test_h5py_stand_in_is_strict(tmp_path):
    """The image has no h5py, so the h5 loader runs against tests/era5_fixture.py's stand-in.  The stand-in must stay HONEST: it
    serves exactly the accesses data/era5.py:58-74 makes -- `File(path, "r")` as a context manager, `f["input"][var][()]` -- and
    refuses anything else, so a loader that drifted to another h5py idiom (slicing, `.value`, attributes, write modes) fails here
    instead of passing against a permissive fake."""
    import era5_fixture as fx
    root = fx.write_tree(str(tmp_path / "era5"))
    m = fx.install_fake_h5py()
    path = sorted(os.listdir(os.path.join(root, "train")))[0]
    with m.File(os.path.join(root, "train", path), "r") as f:
        a = f["input"][fx.VARS[0]][()]
        assert a.shape == fx.SHAPE and a.dtype == np.float32
        assert isinstance(f["input"]["time"][()], bytes)
        with pytest.raises(KeyError):
            f["input"]["no_such_variable"]
        with pytest.raises(KeyError):
            f["forecast"]
        with pytest.raises(AssertionError):
            f["input"][fx.VARS[0]][:]
        with pytest.raises(AssertionError):
            f["input"][fx.VARS[0]][0]
        assert not hasattr(f["input"][fx.VARS[0]], "value") and not hasattr(f, "attrs")
    with pytest.raises(AssertionError):
        m.File(os.path.join(root, "train", path), "w")
