#!/usr/bin/env python3
"""GPU box: what the warp backward sees inside the benchmark model (random-init FlowOccNet at 8 x 436x1024): per call, the scaled flow's
magnitude and its coherence inside the gather kernel's 32x8 destination tiles (how many sources fall outside a +-6 pixel window around the
tile when the window is fixed, and when it is shifted by the tile's mean flow)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ocflow_amd import ops

from ocflow_amd.runtime import init_distributed, synthetic_batch, use_repo_miopen_db

args = bench.parse()
use_repo_miopen_db(None)
env = init_distributed()
stepper = bench.build_stepper(args, env)
batch = stepper.prepare(synthetic_batch(args.batch, args.height, args.width, env.device, seed=42, index_offset=0))
orig = ops._warp_bwd_raw
seen = []


def spy(img, flow, gout, flow_scale, *a):
    f = flow.detach().float() * flow_scale
    B, _, H, W = f.shape
    mag = f.abs().amax(1)
    Hp, Wp = (H + 7) // 8 * 8, (W + 31) // 32 * 32
    fp = torch.nn.functional.pad(f, (0, Wp - W, 0, Hp - H), mode="replicate")
    t = fp.view(B, 2, Hp // 8, 8, Wp // 32, 32)
    mean = t.mean((3, 5), keepdim=True)
    dev = (t - mean).abs().amax(1)
    seen.append((tuple(img.shape), float(mag.mean()), float(mag.quantile(0.9)), float((mag > 5).float().mean()), float((dev > 5).float().mean()),
                 float((dev > 12).float().mean())))
    return orig(img, flow, gout, flow_scale, *a)


ops._warp_bwd_raw = spy
for i in range(2):
    stepper.step(batch, i)
torch.cuda.synchronize()
for s in seen[-4:]:
    print("img %s | mean |flow| %.1f px, p90 %.1f | far (fixed +-5 window) %.0f %% | far (window shifted by the tile mean, +-5) %.0f %% | (+-12) %.0f %%"
          % (s[0], s[1], s[2], 100 * s[3], 100 * s[4], 100 * s[5]))
