#!/usr/bin/env python3
"""GPU box: which Python call sites issue the aten::copy_ / contiguous / cat / fill launches of one eager benchmark step (torch.profiler
with stacks, grouped by the innermost frames inside this repo)."""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ocflow_amd.runtime import init_distributed, synthetic_batch, use_repo_miopen_db

args = bench.parse()
use_repo_miopen_db(None)
env = init_distributed()
stepper = bench.build_stepper(args, env)
batch = stepper.prepare(synthetic_batch(args.batch, args.height, args.width, env.device, seed=42, index_offset=0))
for i in range(3):
    stepper.step(batch, i)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    stepper.step(batch, 3)
    torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0, set()])
for ev in prof.events():
    if ev.name not in ("aten::copy_", "aten::fill_", "aten::cat", "aten::add_", "aten::add", "aten::zero_", "aten::_to_copy"):
        continue
    t = ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
    if t <= 0:
        continue
    frames = [f for f in (ev.stack or []) if "/repo/" in f or "ocflow_amd" in f or "autograd" in f]
    site = " <- ".join(f.split("/")[-1][:60] for f in frames[:2]) or "(engine / backward node)"
    k = (ev.name, site)
    agg[k][0] += 1
    agg[k][1] += t
    if ev.input_shapes:
        agg[k][2].add(str(ev.input_shapes[0])[:30])
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print(f"{'op':14s} {'calls':>5s} {'us':>8s}  site")
for (name, site), (n, t, shapes) in rows[:40]:
    print(f"{name:14s} {n:5d} {t:8.1f}  {site}  {sorted(shapes)[:3]}")
