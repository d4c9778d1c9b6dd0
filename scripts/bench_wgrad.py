#!/usr/bin/env python3
"""GPU box: ocf_conv3x3_wgrad vs the framework's (MIOpen) weight gradient on the decoder layer shapes of the benchmark
(8 pairs at 436x1024: levels 2..6), bf16 channels_last.  Prints time per call and the max relative error against an fp32 reference."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ocflow_amd import ops
from ocflow_amd.runtime import use_repo_miopen_db

use_repo_miopen_db()
dev = "cuda:0"
SHAPES = [(8, 128, 128, 109, 256, 1), (8, 128, 96, 109, 256, 1), (8, 96, 64, 109, 256, 1), (8, 64, 32, 109, 256, 1), (8, 32, 16, 109, 256, 1),
          (8, 128, 128, 109, 256, 2), (8, 128, 128, 109, 256, 4), (8, 128, 96, 109, 256, 8), (8, 96, 64, 109, 256, 16),
          (8, 128, 128, 55, 128, 1), (8, 128, 96, 28, 64, 1), (8, 96, 64, 14, 32, 1), (8, 64, 32, 7, 16, 1)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for B, Ci, Co, H, W, d in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((B, Ci, H, W), device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    gy = torch.randn((B, Co, H, W), device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn((Co, Ci, 3, 3), device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    ref = torch.ops.aten.convolution_backward(gy.float(), x.float(), w.float(), None, (1, 1), (d, d), (d, d), False, (0, 0), 1, [False, True, False])[1]
    got = ops.conv3x3_wgrad(x, gy, d)
    err = float((got - ref).abs().max() / ref.abs().max())
    acc = torch.zeros((Co, Ci, 3, 3), device=dev).contiguous(memory_format=torch.channels_last)
    with ops.kernel_policy(wgrad=2):
        got2 = ops.conv3x3_wgrad(x, gy, d)
        err2 = float((got2 - ref).abs().max() / ref.abs().max())
        t_v2 = timeit(lambda: ops.conv3x3_wgrad(x, gy, d, out=acc))
    with ops.kernel_policy(wgrad=1):
        t_ours = timeit(lambda: ops.conv3x3_wgrad(x, gy, d, out=acc))
    t_mi = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (d, d), (d, d), False, (0, 0), 1, [False, True, False]))
    flops = 2.0 * B * H * W * Ci * Co * 9
    print(f"B{B} {Ci:3d}->{Co:3d} {H}x{W} d{d:2d} | v2 {t_v2:7.1f} us ({flops / t_v2 / 1e6:6.1f} TF/s, err {err2:.1e}) | v1 {t_ours:7.1f} us ({flops / t_ours / 1e6:6.1f} TF/s) | MIOpen (kernel + helpers) {t_mi:7.1f} us | "
          f"max rel err {err:.2e}", flush=True)
