#!/usr/bin/env python3
"""GPU box: ocf_concat_channels at the decoder-input shapes of the benchmark (8 pairs, level 2 = 109 x 256 and level 3), hipGraph replay."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ocflow_amd  # noqa: F401
from ocflow_amd import ops

dev = "cuda:0"


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    del g
    return best


def cl(*shape):
    return torch.randn(*shape, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)


for H, W in ((109, 256), (55, 128)):
    B = 8
    pair = cl(2 * B, 32, H, W)
    f1 = pair.view(2, B, 32, H, W)[0]
    cases = {
        "flow est: corr81 + f1 32 + 2 + 2 -> 128": ([cl(B, 81, H, W), f1, cl(B, 2, H, W), cl(B, 2, H, W)], 128),
        "occ est: f1 32 + w2 32 + 1 + 1 -> 96": ([f1, cl(B, 32, H, W), cl(B, 1, H, W), cl(B, 1, H, W)], 96),
        "aligned: 32 + 32 -> 64": ([f1, cl(B, 32, H, W)], 64),
        "aligned: 32 + 32 + zeros -> 96": ([f1, cl(B, 32, H, W)], 96),
        "context: 32 + 2 -> 64": ([cl(B, 32, H, W), cl(B, 2, H, W)], 64),
    }
    for name, (parts, cout) in cases.items():
        t = timeit(lambda: ops.concat_channels(parts, cout))
        nb = sum(p.numel() for p in parts) * 2 + B * H * W * cout * 2
        print(f"{H:3d}x{W:3d} {name:44s} {t:7.1f} us  {nb / t / 1e3:7.0f} GB/s", flush=True)
