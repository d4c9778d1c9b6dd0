#!/usr/bin/env python3
"""GPU box: ocf_conv3x3_c16 against the framework's convolution for the encoder's 16 -> 16 layer at 16 x 218 x 512 (hipGraph replay)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ocflow_amd  # noqa: F401
from ocflow_amd import conv
from ocflow_amd.runtime import use_repo_miopen_db

use_repo_miopen_db(None)
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


for B, H, W in ((16, 218, 512), (16, 109, 256)):
    x = torch.randn(B, 16, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, 16, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(16, 16, 3, 3, device=dev) * 0.1).bfloat16().contiguous(memory_format=torch.channels_last)
    t = [timeit(lambda: conv._c16_raw(x, w, False)), timeit(lambda: torch.ops.aten.convolution(x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1)),
         timeit(lambda: conv._c16_raw(gy, w, True)),
         timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, [True, False, False]))]
    gw = torch.zeros(16, 16, 3, 3, device=dev)
    wsb = torch.empty(64 << 20, device=dev, dtype=torch.uint8)
    from ocflow_amd import _lib
    from ocflow_amd.ops import _t4
    import ctypes as C
    def wg():
        so, si, sky, skx = gw.stride()
        _lib.check(_lib.load().ocf_conv3x3_c16_wgrad(_t4(x), _t4(gy), C.c_void_p(gw.data_ptr()), so, si, sky, skx, C.c_void_p(wsb.data_ptr()), B, H, W, _lib.OCF_BF16,
                                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), "wgrad")
    tw = [timeit(wg), timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, [False, True, False]))]
    print(f"   wgrad ours {tw[0]:.1f} lib {tw[1]:.1f} us")
    print(f"{B}x16x{H}x{W}: fwd ours {t[0]:.1f} lib {t[1]:.1f} | dgrad ours {t[2]:.1f} lib {t[3]:.1f} us", flush=True)
