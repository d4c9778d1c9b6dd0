#!/usr/bin/env python3
"""GPU box: the decoder-input convolutions of the benchmark model (input = concatenation of cost volume, features, flow ...: 117, 66, 149,
34, 130, 181, 194 ... channels) through the framework with the input channel count as is and zero-padded to a multiple of 8 / 16 / 32
(MIOpen find mode, bf16 channels_last): forward + input gradient + weight gradient per call."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.backends.cudnn.benchmark = True
SHAPES = [(117, 128, 109, 256), (66, 128, 109, 256), (34, 128, 109, 256), (149, 128, 55, 128), (130, 128, 55, 128), (181, 128, 28, 64),
          (194, 128, 28, 64), (213, 128, 14, 32), (258, 128, 14, 32), (277, 128, 7, 16)]
B = 8


def timeit(fn, n=10):
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


for ci, co, H, W in SHAPES:
    line = f"{ci:3d}->{co} {H}x{W}:"
    for mult in (1, 8, 16, 32):
        c = (ci + mult - 1) // mult * mult
        if mult > 1 and c == (ci + mult // 2 - 1) // (mult // 2) * (mult // 2) and mult > 8:
            line += f" | /{mult} = same"
            continue
        x = torch.randn((B, c, H, W), device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        w = torch.randn((co, c, 3, 3), device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        gy = torch.randn((B, co, H, W), device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        a = ((1, 1), (1, 1), (1, 1), False, (0, 0), 1)
        tf = timeit(lambda: torch.ops.aten.convolution(x, w, None, *a))
        td = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, *a, [True, False, False]))
        tw = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, *a, [False, True, False]))
        line += f" | C={c}: fwd {tf:6.1f} dgrad {td:6.1f} wgrad {tw:6.1f} = {tf + td + tw:7.1f} us"
    print(line, flush=True)
