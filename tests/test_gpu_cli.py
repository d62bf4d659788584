"""-m gpu: the two entry points end to end on a small run directory (reference CLI surface: generate.py:23-43,
train.py:135-343): `python -m swift_amd.train ...` writes .hydra/config.yaml + checkpoints, `python -m swift_amd.generate`
reads them back and writes the (samples, members, steps+1, C, H, W) npy."""
import os
import subprocess
import sys

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, cwd, env=None):
    e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HYDRA_RUN_ID="000")
    e.update(env or {})
    p = subprocess.run([sys.executable, "-m"] + args, cwd=cwd, env=e, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return p.stdout


def test_train_then_generate_cli(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    small = ["data=era5-synthetic-1.4", "data.dataset.img_resolution=[64,64]", "data.dataset.length=48", "data.data_workers=0",
             "model.depth=2", "optimizer=adamw", "trainer.total_kimg=0.008", "trainer.kimg_per_tick=0.004",
             "trainer.checkpoint_ticks=1", "trainer.lr_rampup_kimg=0", "trainer.val_ticks=null", "data.batch_size=2"]
    out = run(["swift_amd.train", "experiment=era5-swinv2-1.4-trigflow"] + small, cwd=str(tmp_path))
    rdir = tmp_path / "results" / "era5-swinv2-1.4-trigflow" / "000"
    cfg = yaml.safe_load(open(rdir / ".hydra" / "config.yaml"))
    assert cfg["model"]["depth"] == 2 and cfg["loss"]["_target_"].endswith("TrigFlowLoss")
    ckpts = sorted(os.listdir(rdir / "checkpoints"))
    assert ckpts, out[-2000:]
    lines = [l for l in open(rdir / "stats.jsonl")]
    assert len(lines) >= 2 and all(np.isfinite(yaml.safe_load(l)["train/loss"]) for l in lines)
    # multistep-CRPS finetune resumed from that run (reference: resume=<run id> finetune=multistep)
    out = run(["swift_amd.train", "experiment=era5-swinv2-1.4-trigflow", "resume=000", "finetune=multistep",
               "finetune.finetune.intervals=[{steps: 2, kimg: 0.004}]"] + small[:5] + ["data.batch_size=2"],
              cwd=str(tmp_path), env={"HYDRA_RUN_ID": "001"})
    rdir2 = tmp_path / "results" / "era5-swinv2-1.4-trigflow" / "001"
    cfg2 = yaml.safe_load(open(rdir2 / ".hydra" / "config.yaml"))
    assert cfg2["loss"]["_target_"].endswith("CRPSLoss") and cfg2["finetune"]["name"] == "multistep"
    # sCM training (forward-mode tangent through the network) distilling from the trigflow run's EMA weights
    run(["swift_amd.train", "experiment=era5-swinv2-1.4-scm", f"distill={rdir}",
         "loss.tangent_warmup_kimg=1", "trainer.val_target_interval=4", "data.val_local_batch_size=2"] +
        [o for o in small if o not in ("optimizer=adamw", "trainer.val_ticks=null")] + ["trainer.val_ticks=1"], cwd=str(tmp_path),
        env={"HYDRA_RUN_ID": "002"})  # the experiment's own optimiser: MuonWithAuxAdam
    rdir3 = tmp_path / "results" / "era5-swinv2-1.4-scm" / "002"
    cfg3 = yaml.safe_load(open(rdir3 / ".hydra" / "config.yaml"))
    assert cfg3["loss"]["_target_"].endswith("SCMLoss") and cfg3["distill"] == str(rdir)  # distillation flag: train.py:318-319
    assert cfg3["optimizer"]["_target_"].endswith("MuonWithAuxAdam")
    # ... with the in-training validation rollout switched on (4 six-hour steps, dpm solver on the EMA weights)
    val = [yaml.safe_load(l) for l in open(rdir3 / "val_stats.jsonl")]
    assert val and np.isfinite(val[0]["val/rmse"]) and len(val[0]["val/rmse/2m_temperature"]) == 2
    lines3 = [yaml.safe_load(l) for l in open(rdir3 / "stats.jsonl")]
    assert len(lines3) >= 2 and all(np.isfinite(l["train/loss"]) for l in lines3)
    assert sorted(os.listdir(rdir3 / "checkpoints"))
    # generation from the first run's latest checkpoint
    run(["swift_amd.generate", "--input", str(rdir), "--members", "2", "--steps", "3", "--samples", "3", "--batch", "4",
         "--dump", "numpy"], cwd=str(tmp_path))
    f = rdir / "output" / "latest" / "output-3i-3s-2m-6h.npy"
    a = np.load(f)
    assert a.shape == (3, 2, 4, 69, 64, 64) and np.isfinite(a).all()
    assert np.abs(a[:, 0] - a[:, 1])[:, 1:].max() > 0  # members differ after the first step, share the initial state
    assert np.array_equal(a[:, 0, 0], a[:, 1, 0])
    # the same job on the fp32-grade split-bf16 engine (--dtype bf16x3, round 4): same counter-based noise, so the trajectories
    # agree with the exact engine's to fp32-grade accuracy -- and are not bit-equal (another engine did run)
    os.rename(f, str(f) + ".exact")
    run(["swift_amd.generate", "--input", str(rdir), "--members", "2", "--steps", "3", "--samples", "3", "--batch", "4",
         "--dump", "numpy", "--dtype", "bf16x3"], cwd=str(tmp_path))
    a3 = np.load(f)
    d3 = np.linalg.norm((a3 - a).astype(np.float64)) / np.linalg.norm(a.astype(np.float64))
    assert np.isfinite(a3).all() and 0 < d3 < 1e-4, d3
    os.rename(str(f) + ".exact", f)
    # the reference's DEFAULT invocation (--dump zarr, generate.py:41-43) + on-device ensemble metrics: same numbers in the
    # per-variable zarr arrays (level axis for the pressure-level variables), metrics file next to the store
    import json
    from swift_amd.utils import zarrlite
    run(["swift_amd.generate", "--input", str(rdir), "--members", "2", "--steps", "3", "--samples", "3", "--batch", "3",
         "--metrics"], cwd=str(tmp_path))
    z = rdir / "output" / "latest" / "output-3i-3s-2m-6h.zarr"
    names = cfg["data"]["dataset"]["variables"]
    chans = zarrlite.variable_channels(names)
    assert os.path.exists(z / ".zmetadata") and sum(len(c) for c in chans.values()) == 69
    for var, ch in chans.items():
        arr = zarrlite.read_array(str(z), var)
        ref = a[:, :, :, ch[0]] if arr.ndim == 5 else a[:, :, :, ch]
        np.testing.assert_array_equal(arr, ref)
    met = json.load(open(rdir / "output" / "latest" / "evaluation_metrics.json"))
    assert len(met) == 3 * 3 * 69 and all(np.isfinite(v) for v in met.values())
    assert f"crps_{names[0]}_6h" in met and f"ssr_{names[-1]}_18h" in met
    # the same job as two ranks started by `--gpus 2` (sharing this box's one GPU, collectives over gloo): IC-major units
    # sharded over the ranks, weights broadcast, per-IC ensemble sums all-gathered -> same store, same metrics
    os.rename(rdir / "output" / "latest" / "evaluation_metrics.json", rdir / "output" / "latest" / "metrics_1rank.json")
    run(["swift_amd.generate", "--gpus", "2", "--input", str(rdir), "--members", "2", "--steps", "3", "--samples", "3", "--batch",
         "3", "--metrics", "--dump", "numpy"], cwd=str(tmp_path), env={"SWIFTK_ALLOW_SHARED_GPU": "1", "SWIFTK_DIST_BACKEND": "gloo"})
    np.testing.assert_array_equal(np.load(f), a)
    met2 = json.load(open(rdir / "output" / "latest" / "evaluation_metrics.json"))
    assert met2.keys() == met.keys() and all(met2[k] == pytest.approx(met[k], rel=1e-6) for k in met)
    # round 6: two ranks and NO --dump: the multi-rank default is metrics only (no raw store: 3.4 GB/s per rank into one filesystem)
    os.remove(rdir / "output" / "latest" / "evaluation_metrics.json")
    os.remove(f)
    run(["swift_amd.generate", "--gpus", "2", "--input", str(rdir), "--members", "2", "--steps", "3", "--samples", "3", "--batch",
         "3"], cwd=str(tmp_path), env={"SWIFTK_ALLOW_SHARED_GPU": "1", "SWIFTK_DIST_BACKEND": "gloo"})
    met3 = json.load(open(rdir / "output" / "latest" / "evaluation_metrics.json"))
    assert met3.keys() == met.keys() and all(met3[k] == pytest.approx(met[k], rel=1e-6) for k in met)
    assert not os.path.exists(f) and not os.path.exists(str(f)[:-4] + ".none")


def test_train_cli_5p6deg_one_by_one_patches(tmp_path):
    """experiment=era5-swinv2-5.6-scm (1x1 patches on the 32x64 grid, head width 69): sCM training + validation run."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    run(["swift_amd.train", "experiment=era5-swinv2-5.6-scm", "data=era5-synthetic-5.6", "data.dataset.length=48",
         "data.data_workers=0", "model.depth=2", "trainer.total_kimg=0.008", "trainer.kimg_per_tick=0.004",
         "trainer.checkpoint_ticks=1", "trainer.lr_rampup_kimg=0", "trainer.val_ticks=1", "trainer.val_target_interval=4",
         "data.val_local_batch_size=2", "data.batch_size=2", "loss.tangent_warmup_kimg=1"], cwd=str(tmp_path))
    rdir = tmp_path / "results" / "era5-swinv2-5.6-scm" / "000"
    lines = [yaml.safe_load(l) for l in open(rdir / "stats.jsonl")]
    assert len(lines) >= 2 and all(np.isfinite(l["train/loss"]) for l in lines)
    val = [yaml.safe_load(l) for l in open(rdir / "val_stats.jsonl")]
    assert val and np.isfinite(val[0]["val/rmse"])
    run(["swift_amd.generate", "--input", str(rdir), "--members", "2", "--steps", "2", "--samples", "2", "--batch", "4",
         "--dump", "numpy"], cwd=str(tmp_path))
    a = np.load(rdir / "output" / "latest" / "output-2i-2s-2m-6h.npy")
    assert a.shape == (2, 2, 3, 69, 32, 64) and np.isfinite(a).all()
    # the data-parallel training step as two ranks started by `--gpus 2` (one GPU shared, collectives over gloo): per-layer
    # gradient all-reduces overlapped with the eager final backward pass, shared run id, rank-0 checkpoint
    run(["swift_amd.train", "--gpus", "2", "experiment=era5-swinv2-5.6-scm", "data=era5-synthetic-5.6", "data.dataset.length=48",
         "data.data_workers=0", "model.depth=2", "optimizer=adamw", "trainer.total_kimg=0.008", "trainer.kimg_per_tick=0.004",
         "trainer.checkpoint_ticks=1", "trainer.lr_rampup_kimg=0", "trainer.val_ticks=null", "data.batch_size=2",
         "loss.tangent_warmup_kimg=1"], cwd=str(tmp_path),
        env={"HYDRA_RUN_ID": "003", "SWIFTK_ALLOW_SHARED_GPU": "1", "SWIFTK_DIST_BACKEND": "gloo"})
    rdir2 = tmp_path / "results" / "era5-swinv2-5.6-scm" / "003"
    lines = [yaml.safe_load(l) for l in open(rdir2 / "stats.jsonl")]
    assert len(lines) >= 2 and all(np.isfinite(l["train/loss"]) for l in lines)
    assert sorted(os.listdir(rdir2 / "checkpoints"))


def test_bench_contract_line(tmp_path):
    """bench.py prints ONE JSON line with the driver's keys, the two roofline objects and (when asked) the CPU baseline."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    e = dict(os.environ, PYTHONPATH=ROOT)
    e.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-steps", "0",
                        "--batch", "16"], cwd=str(tmp_path), env=e, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"]
    for r in (d["roofline"], d["attention_roofline"]):
        assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r)
        assert 0.05 < r["frac"] < 1.0 and r["achieved"] == pytest.approx(r["frac"] * r["peak"], rel=1e-6)
    assert d["value"] == pytest.approx(16 * 2 / (d["ms_per_step"] * 2 / 1e3), rel=1e-3)
    # round-2 objects: what was collected, the process group, the parity (fp32) engine and the bf16 drift
    assert d["rccl"]["world"] == 1 and d["checksum"]["units_collected"] == 16 and len(d["checksum"]["per_step_all_units_sum"]) == 2
    assert np.isfinite(d["checksum"]["last_step_rank0_units_sum"])
    pe, dr = d["parity_engine"], d["bf16_vs_fp32"]
    assert pe["dtype"] == "f32" and 0.3 < pe["frac_of_fp32_matrix_peak"] < 1.0 and 0.2 < pe["attention_mfma_frac"] < 1.0
    assert 0 < dr["rel_l2_after_1_steps"] < dr["rel_l2_after_60_steps"] < 1.0
    # round-4 objects: SURVEY 8d's batch sweep, configs[2] (39-evaluation sampler), the north-star rollout loop, the split engine
    sw = d["batch_sweep"]
    assert "error" not in sw and all(str(b) in sw for b in (1, 4, 8, 16)) and sw["1"]["best"] < sw["8"]["best"]
    assert "hip_graph" in sw["1"] and sw["1"]["best"] > 100
    c3 = d["config3_2s"]
    assert "error" not in c3 and c3["1"]["network_evals_per_s"] == pytest.approx(39 * c3["1"]["value"])
    ro = d["rollout_12x8x60"]
    assert "error" not in ro and 0.5 * d["value"] < ro["value"] < 1.5 * d["value"] and np.isfinite(ro["checksum"])
    sp = pe["split_bf16_engine"]
    assert "error" not in sp and sp["value"] > pe["value"] and 0 < sp["rel_l2_vs_exact_engine_after_1_step"] < 1e-4
    # (the profiled SQ figures ride along only in the profiled configuration -- 96 units, bf16, default tuning; any other run
    # carries the pointer to the profile and no numbers that are not its own)
    assert d["config"]["noise"].startswith("swiftk_unit_noise") and "profiles/" in d["attention_roofline"]["sq_counters"]
    assert ("mfma_pipe_busy" in d["attention_roofline"]) == (d["config"]["units_per_gpu_per_step"] == 96)
    # the training step behind the path's weights, measured in child processes (reported extra)
    for leg, lo, hi in (("crps_finetune_steps4", 0.4, 2.0), ("scm_pretrain", 0.05, 0.5), ("trigflow", 0.04, 0.4)):
        t = d["training"][leg]
        assert "error" not in t, t
        assert t["unit"] == "s/iteration" and lo < t["value"] < hi and 0.1 < t["roofline"]["frac"] < 0.6
    # N ranks asked for, one GPU here: the launcher refuses before touching the device; a mismatching process group exits 3
    p2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=str(tmp_path), env=e,
                        capture_output=True, text=True, timeout=300)
    if torch.cuda.device_count() < 2:
        assert p2.returncode != 0 and "GPU(s)" in p2.stderr and not p2.stdout.strip()
    p3 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=str(tmp_path),
                        env=dict(e, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=300)
    assert p3.returncode == 3 and "process group" in p3.stderr
    p4 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "300"], cwd=str(tmp_path), env=e,
                        capture_output=True, text=True, timeout=300)
    assert p4.returncode == 2 and "SWIFTK_MAX_UNITS" in p4.stderr


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """The N-rank path of bench.py end to end on this one-GPU box: `bench.py --gpus 2` starts two fresh rank processes that
    share GPU 0 (SWIFTK_ALLOW_SHARED_GPU) and run their collectives over gloo (SWIFTK_DIST_BACKEND; RCCL refuses two ranks on
    one device): weight broadcast, per-step all-gather of the unit checksums, max-over-ranks timing.  Rank 0's units are the
    1-rank run's units, so their checksums must agree bit for bit; `generate --gpus 2` shards IC-major units the same way."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    e = dict(os.environ, PYTHONPATH=ROOT, SWIFTK_ALLOW_SHARED_GPU="1", SWIFTK_DIST_BACKEND="gloo", SWIFTK_BENCH_STRONG="2x3x2")
    e.pop("WORLD_SIZE", None)
    common = ["--steps", "3", "--warmup", "1", "--batch", "4", "--no-extras"]
    out = {}
    for n in (1, 2):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + common, cwd=str(tmp_path), env=e,
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]
        out[n] = json.loads(lines[0])
    d1, d2 = out[1], out[2]
    assert d1["n_gpus"] == 1 and d2["n_gpus"] == 2 and d2["rccl"]["world"] == 2 and d2["rccl"]["backend"] == "gloo"
    assert d2["checksum"]["units_collected"] == 8 and d1["checksum"]["units_collected"] == 4
    assert d2["checksum"]["last_step_rank0_units_sum"] == d1["checksum"]["last_step_rank0_units_sum"]
    assert d2["checksum"]["last_step_first_units"] == d1["checksum"]["last_step_first_units"]
    assert d2["checksum"]["last_step_all_units_sum"] != d1["checksum"]["last_step_all_units_sum"]
    assert d2["value"] == pytest.approx(2 * 4 * 3 / (d2["ms_per_step"] * 3 / 1e3), rel=1e-3)
    # the per-rank split of the timed region and, for N > 1, the strong-scaling job (BASELINE configs[3]; shrunk here) in the same line
    for d, n in ((d1, 1), (d2, 2)):
        pr = d["per_rank"]
        assert all(len(pr[k]) == n for k in ("compute_ms", "collective_ms", "barrier_wait_ms", "timed_region_s"))
        assert all(v > 0 for v in pr["compute_ms"]) and all(v >= 0 for v in pr["collective_ms"]) and min(pr["barrier_wait_ms"]) < 50.0
        assert max(c + x for c, x in zip(pr["compute_ms"], pr["collective_ms"])) <= 1e3 * max(pr["timed_region_s"]) * 1.05
    st = d2["rollout_2x3x2"]
    assert "rollout_2x3x2" not in d1 and st["scaling"] == "strong" and st["value"] > 0 and st["config"]["units"] == 6
    assert st["per_rank"]["units"] == [3.0, 3.0] and len(st["per_rank"]["barrier_wait_ms"]) == 2


@pytest.mark.timeout(900)
def test_train_and_generate_cli_at_swiftb_size_with_loader_workers(tmp_path):
    """The entry points as a user runs them -- full Swift-B, the data config's own loader workers and pin-memory thread, run
    directory named by its start time -- for a few dozen iterations.  The small CLI tests above run with `data.data_workers=0`;
    only at this size and with the loader threads alive did two faults show: the first captured launch sequence invalidated by
    the pin-memory thread (graphs.capture), and `resume=<start-time id>` read as an integer (config._parse_value)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    big = ["data=era5-synthetic-1.4", "data.batch_size=8", "data.dataset.length=64", "trainer.total_kimg=0.16",
           "trainer.kimg_per_tick=0.08", "trainer.val_ticks=null", "trainer.checkpoint_ticks=1", "trainer.lr_rampup_kimg=0"]
    e = {k: v for k, v in os.environ.items() if k != "HYDRA_RUN_ID"}
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    p = subprocess.run([sys.executable, "-m", "swift_amd.train", "experiment=era5-swinv2-1.4-scm"] + big, cwd=str(tmp_path), env=e,
                       capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    runs = sorted(os.listdir(tmp_path / "results" / "era5-swinv2-1.4-scm"))
    assert len(runs) == 1 and len(runs[0]) == 15 and runs[0][8] == "_", runs  # YYYYMMDD_HHMMSS
    rdir = tmp_path / "results" / "era5-swinv2-1.4-scm" / runs[0]
    lines = [yaml.safe_load(l) for l in open(rdir / "stats.jsonl")]
    assert len(lines) >= 2 and all(np.isfinite(l["train/loss"]) for l in lines)
    per_iter = lines[-1]["train/dt/kimg"] * 8 / 1000.0
    print(f"sCM + MuonWithAuxAdam through the CLI, Swift-B, local batch 8: {per_iter:.3f} s per iteration")
    assert per_iter < 0.5  # (0.16-0.19 s measured; tools/train_bench.py without a loader: 0.155)
    assert sorted(os.listdir(rdir / "checkpoints"))
    # multistep-CRPS finetune resumed from the start-time id, two iterations
    e2 = dict(e, HYDRA_RUN_ID="ft")
    p = subprocess.run([sys.executable, "-m", "swift_amd.train", "experiment=era5-swinv2-1.4-scm", f"resume={runs[0]}", "finetune=multistep",
                        "finetune.finetune.intervals=[{steps: 4, kimg: 0.016}]", "data=era5-synthetic-1.4", "data.batch_size=8",
                        "data.dataset.length=64", "trainer.kimg_per_tick=0.008", "trainer.val_ticks=null",
                        "trainer.checkpoint_ticks=null"], cwd=str(tmp_path), env=e2, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    cfg2 = yaml.safe_load(open(tmp_path / "results" / "era5-swinv2-1.4-scm" / "ft" / ".hydra" / "config.yaml"))
    assert cfg2["loss"]["_target_"].endswith("CRPSLoss")
    # the 15-day job's per-GPU share, shortened: 12 members x 2 ICs x 6 steps from the run's checkpoint, zarr + metrics
    e3 = dict(e)
    p = subprocess.run([sys.executable, "-m", "swift_amd.generate", "--input", str(rdir), "--members", "12", "--steps", "6", "--samples",
                        "2", "--batch", "12", "--dtype", "bf16", "--metrics"], cwd=str(tmp_path), env=e3, capture_output=True, text=True,
                       timeout=800)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    outs = os.listdir(rdir / "output" / sorted(os.listdir(rdir / "output"))[0])
    assert "evaluation_metrics.json" in outs and any(o.endswith(".zarr") for o in outs)


def test_crps_iterations_with_a_process_group_stay_finite(tmp_path):
    """Regression guard for the overflow of the norm / modulation / embedding gradients seen at the end of round 5 in data-parallel
    CRPS runs.  Round 6 pinned it (tools/memset_graph_repro.hip): a hipMemsetAsync captured into a
    HIP graph and replayed on the null stream writes a STALE fill pattern under the HIP runtime this PyTorch bundles -- with the
    clears done by hipMemsetAsync (tuning key 25) 16 of 16 such runs overflow, with the library's fill kernel none.  Here: twelve
    multistep-CRPS iterations at Swift-B size, local batch 8, TWICE with a one-rank RCCL group (early-announced all-reduces, sync,
    fused optimizer) and once without a group.  Every run must end finite with NO overflowed Adam moment, and the gradients of the
    first iteration -- same weights, same draws -- must agree between the data-parallel path and the group-less one."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for k, dist_flag in enumerate(("1", "1", "0")):
        dg = str(tmp_path / f"digest{k}.json")
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "train_bench.py"), "--loss", "crps", "--iters", "12", "--dist", dist_flag,
                            "--grad-digest", dg], capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        both = p.stdout + p.stderr
        assert "NON-FINITE" not in both and "loss nan" not in both and " nan" not in both.split("loss per iteration")[-1].splitlines()[0], both[-2500:]
        assert "OVERFLOW-CHECK parameters with non-finite exp_avg_sq: 0:" in both, both[-2500:]
        rec = json.loads(next(ln for ln in p.stdout.splitlines() if ln.startswith("{")))
        assert rec["value"] > 0
        digests.append(json.load(open(dg)))
    ref = digests[2]["first"]  # the group-less run
    for d in digests[:2]:
        worst = 0.0
        for n, (nrm, _) in ref.items():
            g = d["first"][n][0]
            assert nrm == nrm and g == g and abs(g) < 1e18
            worst = max(worst, abs(g - nrm) / max(nrm, 1e-30))
        print(f"  one-rank group vs group-less, first iteration: worst per-tensor gradient-norm deviation {worst:.2e}")
        assert worst < 1e-4, worst
    for d in digests:
        assert all(v[0] == v[0] and abs(v[0]) < 1e18 for v in d["last"].values())


def test_training_iterations_issue_no_device_memset():
    """No launch sequence of the training path may contain a device memset (hipMemsetAsync): captured into a HIP graph and replayed
    on the null stream it writes a stale pattern under this PyTorch's HIP runtime (tools/memset_graph_repro.hip).  torch.profiler over
    one eager iteration of each loss (depth 2): zero runtime memset calls -- neither from the library (swiftk_zero_f32 is a kernel)
    nor from the ATen glue around it."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for which in ("crps", "scm", "trigflow"):
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "memset_sites.py"), which], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        line = next(ln for ln in p.stdout.splitlines() if ln.startswith(which + ":"))
        assert f"{which}: 0 runtime memset calls, 0 device memset activities" in line, p.stdout[-1500:]
