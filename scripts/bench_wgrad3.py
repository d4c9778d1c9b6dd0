#!/usr/bin/env python3
"""GPU box: ocf_conv3x3_wgrad_bias against the framework's weight gradient (MIOpen: workspace fill + split-K kernel + cast) at the
decoder's layer shapes, 8 pairs at 436 x 1024, bf16 channels_last, interleaved rounds in one process.
usage: python scripts/bench_wgrad3.py [--levels 2,3] [--policies 0,1088,...]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ocflow_amd  # noqa: F401
from ocflow_amd import conv
from ocflow_amd.runtime import use_repo_miopen_db

ap = argparse.ArgumentParser()
ap.add_argument("--levels", default="2,3")
ap.add_argument("--policies", default="0")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
use_repo_miopen_db(None)
dev = "cuda:0"
B = 8
HW = {2: (109, 256), 3: (55, 128), 4: (28, 64), 5: (14, 32), 6: (7, 16)}
FLOW_IN = {2: 128, 3: 160, 4: 192, 5: 224, 6: 288}
OCC_IN = {2: 96, 3: 160, 4: 224, 5: 288, 6: 416}
layers = []
for lvl in [int(v) for v in args.levels.split(",")]:
    H, W = HW[lvl]
    chain = [(FLOW_IN[lvl], 128, 1), (128, 128, 1), (128, 96, 1), (96, 64, 1), (64, 32, 1), (OCC_IN[lvl], 128, 1), (32, 16, 1)]
    if lvl == 2:
        chain += [(64, 128, 1), (128, 128, 4), (128, 96, 8), (96, 64, 16)]
    seen = set()
    for ci, co, d in chain:
        if (ci, co, d) not in seen:
            seen.add((ci, co, d))
            layers.append((ci, co, d, H, W))
pols = [int(p) for p in args.policies.split(",")]


def timeit(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print(f"{'layer':28s} | lib wgrad | ours by policy {pols} (us, TF/s)")
tot_lib, tot = 0.0, [0.0] * len(pols)
for ci, co, d, H, W in layers:
    x = torch.randn(B, ci, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device=dev) * 0.05).bfloat16().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    gw = torch.zeros(co, ci, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    gb = torch.zeros(co, device=dev)
    fns = [lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (d, d), (d, d), False, (0, 0), 1, [False, True, False])]
    for p in pols:
        fns.append(lambda p=p: conv.wgrad_bias_raw(x, gy, gw, gb, d, policy=p))
    for fn in fns:
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    best = [1e30] * len(fns)
    for _ in range(args.rounds):
        for i, fn in enumerate(fns):
            best[i] = min(best[i], timeit(fn, args.iters))
    fl = 2.0 * B * H * W * 9 * ci * co
    print(f"{ci:4d}->{co:3d} d{d:2d} {H:3d}x{W:3d}      | {best[0]:7.1f}   | " + "  ".join(f"{t:7.1f} ({fl / t / 1e6:4.0f})" for t in best[1:]), flush=True)
    tot_lib += best[0]
    for i in range(len(pols)):
        tot[i] += best[1 + i]
print(f"{'sum':28s} | {tot_lib:7.1f}   | " + "  ".join(f"{t:7.1f}" for t in tot))
