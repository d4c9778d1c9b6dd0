"""Data-parallel path on CPU: world_size 2, backend gloo (the no-cluster stand-in for RCCL over xGMI).  The hot ops are
computed by the oracle inside the worker processes (the product has no CPU path); what is under test is the runtime:
sharding, DDP wiring through the Lightning-style training_step, gradient averaging, max-over-ranks timing."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make_batch(n, seed=7):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((n, 6, 64, 64), generator=g) * 2 - 1
    flow = torch.randn((n, 2, 64, 64), generator=g)
    occ = (torch.rand((n, 1, 64, 64), generator=g) < 0.3).float()
    return x, flow, occ


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from golden_util import rngfree_fill
    from oracle.backend import patch_ops
    patch_ops(None)
    from ocflow_amd.models.flow_occ_model import FlowOccModel
    from ocflow_amd.runtime import TrainStepper, all_reduce_max, barrier, init_distributed, shard_range, shutdown_distributed
    env = init_distributed(force_cpu=True)
    assert env.backend == "gloo" and env.world_size == world
    model = rngfree_fill(FlowOccModel({"model": "pwoc", "learning_rate": 1e-3}))  # no batch-coupled statistics
    stepper = TrainStepper(model, env, amp_dtype=None, channels_last=False)
    full = _make_batch(2 * world)
    lo, hi = shard_range(2 * world, env)
    batch = tuple(t[lo:hi] for t in full)
    stepper.optimizer.zero_grad(set_to_none=True)
    loss = stepper.module(batch, 0)
    loss.backward()  # DDP all-reduces (averages) the gradients here
    grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    barrier(env)
    tmax = all_reduce_max(env, float(rank + 1))
    assert tmax == float(world)
    torch.save({"grads": grads, "loss": loss.detach(), "range": (lo, hi)}, os.path.join(out_dir, f"rank{rank}.pt"))
    shutdown_distributed()


def _worker_steps(rank, world, port, out_dir):
    """several consecutive optimisation steps of the HEADLINE model under DDP: FlowOccNet owns parameters that never reach the
    loss, which DDP only trips over at the second forward ("Expected to have finished reduction in the prior iteration")"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from golden_util import rngfree_fill
    from oracle.backend import patch_ops
    patch_ops(None)
    from ocflow_amd.models.flow_occ_model import FlowOccModel
    from ocflow_amd.runtime import TrainStepper, init_distributed, shard_range, shutdown_distributed
    env = init_distributed(force_cpu=True)
    model = rngfree_fill(FlowOccModel({"model": "flowoccnet", "learning_rate": 1e-4}))
    stepper = TrainStepper(model, env, amp_dtype=None, channels_last=False)
    full = _make_batch(2 * world)
    lo, hi = shard_range(2 * world, env)
    batch = stepper.prepare(tuple(t[lo:hi] for t in full))
    losses = [float(stepper.step(batch, i)) for i in range(3)]
    unused = [k for k, p in model.named_parameters() if p.grad is None]
    torch.save({"losses": losses, "unused": unused, "params": {k: p.detach().clone() for k, p in model.named_parameters()}},
               os.path.join(out_dir, f"steps_rank{rank}.pt"))
    shutdown_distributed()


@pytest.mark.timeout(900)
def test_ddp_world2_consecutive_steps_headline_model(tmp_path):
    world = 2
    mp.start_processes(_worker_steps, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "steps_rank0.pt")
    r1 = torch.load(tmp_path / "steps_rank1.pt")
    assert all(torch.isfinite(torch.tensor(r["losses"])).all() for r in (r0, r1))
    assert r0["unused"] == r1["unused"]
    # replicas stay in lock-step: same parameters on both ranks after three averaged-gradient Adam updates
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k


@pytest.mark.timeout(600)
def test_ddp_world2_matches_single_process(tmp_path, monkeypatch):
    world = 2
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert r0["range"] == (0, 2) and r1["range"] == (2, 4)
    # ranks hold identical (averaged) gradients
    for k in r0["grads"]:
        assert torch.allclose(r0["grads"][k], r1["grads"][k], atol=1e-7), k

    # single process on the full global batch: mean-reduced losses => same gradients as the DDP average
    from golden_util import rngfree_fill
    from oracle_backend import patch_ops
    patch_ops(monkeypatch)
    from ocflow_amd.models.flow_occ_model import FlowOccModel
    model = rngfree_fill(FlowOccModel({"model": "pwoc", "learning_rate": 1e-3}))
    loss = model.training_step(_make_batch(2 * world), 0)
    loss.backward()
    assert abs(float(loss) - 0.5 * (float(r0["loss"]) + float(r1["loss"]))) < 1e-5
    worst = 0.0
    for k, p in model.named_parameters():
        if p.grad is not None:
            worst = max(worst, float((p.grad - r0["grads"][k]).abs().max()) / (float(p.grad.abs().max()) + 1e-12))
    assert worst < 5e-4, worst


def test_shard_range_covers_global_batch():
    from ocflow_amd.runtime import DistEnv, shard_range
    for world in (1, 2, 3, 8):
        for n in (8, 9, 64):
            spans = [shard_range(n, DistEnv(rank=r, world_size=world)) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_synthetic_batch_is_shard_reproducible():
    from ocflow_amd.runtime import synthetic_batch
    full = synthetic_batch(4, 64, 96, "cpu", seed=3)
    part = synthetic_batch(2, 64, 96, "cpu", seed=3, index_offset=2)
    for a, b in zip(full, part):
        assert torch.equal(a[2:], b)
    imgs, flow, occ = full
    assert imgs.shape == (4, 6, 64, 96) and flow.shape == (4, 2, 64, 96) and occ.shape == (4, 1, 64, 96)
    assert float(imgs.abs().max()) <= 1.0 and set(occ.unique().tolist()) <= {0.0, 1.0}
