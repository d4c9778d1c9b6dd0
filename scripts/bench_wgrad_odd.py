#!/usr/bin/env python3
"""GPU box: the framework's (MIOpen) 3x3 weight gradient, bf16 channels_last, on the PWC decoder's dense-block layers, whose input channel
counts are not a multiple of 8 (117, 245, 373, 469, 533 at level 2: 81 cost-volume channels + features + flow + dense concatenation).
Time per call (kernel + its zero-fill / cast helpers) and the matrix-core rate it corresponds to."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ocflow_amd.runtime import use_repo_miopen_db

use_repo_miopen_db()
dev = "cuda:0"
SHAPES = [(8, 117, 128, 109, 256), (8, 245, 128, 109, 256), (8, 373, 96, 109, 256), (8, 469, 64, 109, 256), (8, 533, 32, 109, 256),
          (8, 248, 128, 109, 256), (8, 149, 128, 55, 128), (8, 277, 128, 55, 128), (8, 565, 32, 55, 128), (8, 181, 128, 28, 64), (8, 597, 32, 28, 64)]


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


for B, Ci, Co, H, W in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((B, Ci, H, W), device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    gy = torch.randn((B, Co, H, W), device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn((Co, Ci, 3, 3), device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    t_w = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, [False, True, False]))
    t_d = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, [True, False, False]))
    t_f = timeit(lambda: torch.ops.aten.convolution(x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1))
    flops = 2.0 * B * H * W * Ci * Co * 9
    print(f"B{B} {Ci:3d}->{Co:3d} {H}x{W} | wgrad {t_w:7.1f} us ({flops / t_w / 1e6:6.1f} TF/s) | dgrad {t_d:7.1f} us ({flops / t_d / 1e6:6.1f}) | fwd {t_f:7.1f} us ({flops / t_f / 1e6:6.1f})",
          flush=True)
