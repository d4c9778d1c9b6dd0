"""Multi-rank rehearsal on ONE GPU (OCFLOW_DIST_REHEARSAL=1, gloo, all ranks on cuda:0): per-step wall time and replica
consistency of the three data-parallel modes of runtime.TrainStepper.
  OCFLOW_DIST_REHEARSAL=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 scripts/dp_rehearsal.py"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocflow_amd  # noqa: F401
import torch
import torch.distributed as dist
import bench
from ocflow_amd.runtime import TrainStepper, init_distributed, synthetic_batch, use_repo_miopen_db, shutdown_distributed

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=6); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--height", type=int, default=436); ap.add_argument("--width", type=int, default=1024)
a = ap.parse_args()
use_repo_miopen_db()
env = init_distributed()
for mode, graph in (("ddp", False), ("flat", False), ("flat", True)):
    torch.manual_seed(42)
    st = TrainStepper(bench.build_module(argparse.Namespace(model="flowoccnet")), env, amp_dtype=torch.bfloat16, channels_last=True,
                      fused_optimizer=True, dp_mode=mode, master_weights=(mode == "flat"))
    batch = st.prepare(synthetic_batch(a.batch, a.height, a.width, env.device, seed=42, index_offset=env.rank * a.batch))
    st.step(batch, 0)
    if graph:
        st.enable_graph(batch); batch = st._static_batch
    times, losses = [], []
    for i in range(a.steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = st.step(batch, i)
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
        losses.append(float(loss))
    # replicas must hold identical parameters
    chk = torch.stack([p.detach().double().sum() for p in st.raw.parameters()]).sum()
    both = [torch.zeros_like(chk) for _ in range(env.world_size)]
    dist.all_gather(both, chk)
    if env.is_main:
        print(f"mode={mode} graph={graph}: ms/step {[round(t, 1) for t in times]} loss {[round(l, 4) for l in losses]} "
              f"param checksum equal across ranks: {all(bool(b == both[0]) for b in both)}", flush=True)
    del st
    torch.cuda.empty_cache()
shutdown_distributed()
