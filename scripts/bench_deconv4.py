"""micro-benchmark of ocf_deconv4x4s2_fwd against the library's transposed convolution at configs[2]'s four levels (8 x 384 x 512)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from ocflow_amd import conv as nconv

DEV = "cuda:0"
LEVELS = [(8, 529, 544, 6, 8), (8, 661, 672, 12, 16), (8, 629, 640, 24, 32), (8, 597, 608, 48, 64)]
ours_only = "--ours-only" in sys.argv


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) * 1000 / n
    del g  # (a live graph with its private pool at interpreter exit has hung the process on this image)
    return t


for B, ci, pitch, H, W in LEVELS:
    buf = torch.randn(B, H, W, pitch, device=DEV).to(torch.bfloat16)
    x = buf.permute(0, 3, 1, 2)[:, :ci]
    w = torch.randn(ci, 2, 4, 4, device=DEV) * ci ** -0.5
    b = torch.randn(2, device=DEV)
    t_ours = timeit(lambda: nconv.deconv4x4s2_bias(x, w, b))
    if ours_only:
        print(f"B{B}x{ci}x{H}x{W}: ours {t_ours:.1f} us")
        continue
    wb = w.to(torch.bfloat16)
    t_lib = timeit(lambda: F.conv_transpose2d(x, wb, None, 2, 1))
    print(f"B{B}x{ci}x{H}x{W}: ours {t_ours:.1f} us, library (bf16 weight given, no bias) {t_lib:.1f} us, features {B*H*W*ci*2/1e6:.1f} MB")

torch.cuda.synchronize()
sys.stdout.flush()
os._exit(0)
