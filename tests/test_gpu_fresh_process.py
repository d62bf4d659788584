"""-m gpu: paths whose FIRST GPU work in a fresh interpreter is the thing under test.

Every other GPU test runs inside one warm pytest process (caching allocator full of slack, libraries initialised, kernels
loaded).  ``generate``, ``bench.py`` and the round-end driver's smoke call start differently: a new process whose first and
only work is one forecast step.  These tests start such processes.

  * ``__graft_entry__.smoke()`` exactly as the driver invokes it (``python3 -c``, stdout block-buffered into a file), several
    times: rc 0, the fp32 figure under the north star's 1e-4, the success marker after it.
  * RCCL for real on one GPU: a process group of ONE rank over the ``nccl`` backend (no gloo override) carrying the
    collectives of both multi-GPU paths -- weight broadcast, all-gather of the per-unit checksums, the asynchronous per-layer
    gradient all-reduce (AVG) started by the backward pass and finished by ``sync()``, barrier, destroy (reference
    training/trainer.py:76-84, generate.py:277-279).
"""
import json
import os
import re
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER_SMOKE = textwrap.dedent("""
    import sys; sys.path.insert(0, ".")
    import __graft_entry__ as e
    f = getattr(e, "smoke", None)
    f(); print("__SMOKE_OK__")
""")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("attempt", range(3))
def test_smoke_in_a_fresh_process(tmp_path, attempt):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    log = tmp_path / "smoke.log"
    with open(log, "w") as fh:  # a file, not a pipe to a tty: Python block-buffers stdout, as under the driver
        p = subprocess.run([sys.executable, "-c", DRIVER_SMOKE], cwd=ROOT, stdout=fh, stderr=subprocess.STDOUT, timeout=800)
    text = log.read_text()
    assert p.returncode == 0, f"smoke exited with {p.returncode} (negative = signal):\n{text[-4000:]}"
    m = re.search(r"fp32 engine vs oracle rel-L2 ([0-9.e+-]+) .*bf16-emulating oracle ([0-9.e+-]+) .*vs the fp32 oracle ([0-9.e+-]+)", text)
    assert m, text[-2000:]
    assert float(m.group(1)) < 1e-4 and float(m.group(2)) < 2e-2 and float(m.group(3)) < 1e-1
    assert text.rstrip().endswith("__SMOKE_OK__")


RCCL_WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %(root)r)
    import torch
    import torch.distributed as tdist
    from swift_amd import dist, ops
    from swift_amd.models.precond import PassPrecond
    from swift_amd.training.loss import TrigFlowLoss
    from swift_amd.training.trainer import GradAllReduce
    from swift_amd.data.era5 import SyntheticERA5Dataset
    from swift_amd.utils.detinit import det_normal, swinv2_state

    dev = torch.device("cuda", 0)
    nv, nf, depth, dim, heads = 69, 3, 2, 1056, 12
    mcfg = dict(_target_="swift.models.swinv2.SwinV2", window_size=[16, 16], shift_size=[8, 8], patch_size=[2, 2], depth=depth,
                dim=dim, heads=heads, logvar=True)
    net = PassPrecond(mcfg, img_resolution=[64, 64], img_channels=nv, condition_channels=nv + nf, auxiliary_dim=1)
    net.load_state_dict(swinv2_state(grid=(32, 32), in_channels=2 * nv + nf, out_channels=nv, patch_size=(2, 2), depth=depth,
                                     dim=dim, heads=heads, logvar=True, seed=5))
    net = net.to(dev)
    names = [f"v{i}" for i in range(nv)]
    ds = SyntheticERA5Dataset(names, ["f0", "f1", "f2"], img_resolution=(64, 64), length=8, seed=5, random_stats=True)
    L = TrigFlowLoss(ds, dict(dist="loguniform", sigma_min=0.02, sigma_max=200.0), sigma_data=1.0).to(dev)
    B = 2
    x, cond, z = (det_normal(s, 5, k).to(dev) for s, k in (((B, nv, 64, 64), "x"), ((B, nv + nf, 64, 64), "c"), ((B, nv, 64, 64), "z")))
    tau, aux = torch.tensor([0.3, 4.0], device=dev).view(B, 1, 1, 1), torch.tensor([0.6, 0.6], device=dev)
    ddp = GradAllReduce(net)

    def grads():
        ddp.zero_grad_flat()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = L(ddp, x, condition=cond, auxiliary=aux, _tau=tau, _z=z)
        loss.backward()
        return float(loss), ddp.sync().clone()

    assert not dist.collectives_active()
    l0, g0 = grads()                                   # no process group: graph-replayed backward, nothing reduced

    os.environ.pop("SWIFTK_DIST_BACKEND", None)
    assert dist.setup_torch(single_rank_group=True) == 0
    assert dist.collectives_active() and tdist.get_backend() == "nccl" and tdist.get_world_size() == 1
    w_before = [p.detach().clone() for p in net.parameters()]
    for p in net.parameters():                          # generate.py:277-279 / DDP's initial broadcast
        tdist.broadcast(p.data, src=0)
    assert all(torch.equal(a, b) for a, b in zip(w_before, net.parameters()))
    ck = ops.unit_checksum(x)                           # output collection: 8 bytes per unit
    ck_all = torch.zeros_like(ck)
    tdist.all_gather_into_tensor(ck_all, ck)
    assert torch.equal(ck_all, ck)
    announced = []
    orig = ddp.reduce_params
    ddp.reduce_params = lambda params: (announced.append(len(params)), orig(params))[1]
    l1, g1 = grads()                                   # eager backward: every layer's slice all-reduced (AVG) asynchronously
    assert len(announced) >= depth + 1 and ddp._pending == [] and ddp._ranges == []
    tdist.barrier()
    rel = float((g1 - g0).norm() / g0.norm())
    out = dict(backend=tdist.get_backend(), world=tdist.get_world_size(), nccl_version=list(torch.cuda.nccl.version()),
               loss_no_group=l0, loss_group=l1, grad_rel_diff=rel, announcements=len(announced), grad_norm=float(g0.norm()))
    tdist.destroy_process_group()
    print("RCCL_RESULT " + json.dumps(out), flush=True)
""")


@pytest.mark.timeout(900)
def test_rccl_single_rank_group_carries_the_collectives(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER % {"root": ROOT})
    env = {k: v for k, v in os.environ.items() if k not in ("SWIFTK_DIST_BACKEND", "MASTER_PORT", "MASTER_ADDR")}
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    rec = json.loads(next(l for l in p.stdout.splitlines() if l.startswith("RCCL_RESULT "))[len("RCCL_RESULT "):])
    print(rec)
    assert rec["backend"] == "nccl" and rec["world"] == 1 and rec["grad_norm"] > 0
    # (the loss mean is a sum of per-workgroup partials added by fp32 atomics: their order is the only freedom between two
    # evaluations of the same forward; observed up to 1.4e-6 relative)
    assert rec["loss_group"] == pytest.approx(rec["loss_no_group"], rel=5e-6)
    # same kernels, eager vs graph-replayed; the only run-to-run freedom is the order of the fp32 atomics in the loss mean
    # and the LayerNorm / bias column sums (measured 4e-5; two eager runs differ by as much, test_graph_replay_equals_eager)
    assert rec["grad_rel_diff"] < 1e-4


PROBE = textwrap.dedent("""
    import json, sys; sys.path.insert(0, ".")
    import torch
    from swift_amd.graphs import memset_node_probe
    print("PROBE " + json.dumps(memset_node_probe()))
""")


@pytest.mark.timeout(600)
def test_captured_clears_replay_clean_and_the_runtime_memset_node_bug_is_recorded():
    """Round 6: a hipMemsetAsync captured into a HIP graph and replayed on the null stream writes a stale fill
    pattern under the HIP runtime PyTorch 2.10.0+rocm7.0 bundles -- what overflowed round 5's gradients.  The library's clear is
    a kernel: replayed 120 times with eager traffic in between it must leave zeros EVERY time.  The memset form is replayed beside
    it and what the runtime does with it is printed (an affected runtime is the expected finding here, a clean one means the
    bundled runtime was fixed): recorded, not asserted -- the product must not depend on it either way."""
    p = subprocess.run([sys.executable, "-c", PROBE], cwd=ROOT, capture_output=True, text=True, timeout=500)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads(next(ln for ln in p.stdout.splitlines() if ln.startswith("PROBE "))[6:])
    print(f"  memset-node probe: {rec}")
    assert rec["kernel_clear_clean"] is True and rec["replays"] == 120
    assert isinstance(rec["memset_node_clean"], bool)
