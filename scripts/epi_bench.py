"""Time the fused conv epilogue kernels against the unfused torch chain on the decoder shapes of the benchmark step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocflow_amd  # noqa: F401
import torch
from ocflow_amd import ops
from oracle import ocflow_oracle as O

DEV = "cuda:0"
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

tot = {"fused_fwd": 0, "fused_bwd": 0, "ref_fwd": 0, "ref_bwd": 0}
for (H, W) in ((7, 16), (14, 32), (28, 64), (55, 128), (109, 256)):
    for C_ in (128, 96, 64, 32, 16):
        x = torch.randn((8, C_, H, W), device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        b = torch.randn(C_, device=DEV, requires_grad=True)
        go = torch.randn_like(x)
        xr = x.clone(memory_format=torch.preserve_format).requires_grad_()
        def f_fwd():
            with torch.no_grad():
                ops.bias_leaky_relu(x, b, 0.1)
        y = ops.bias_leaky_relu(xr * 1, b, 0.1)
        def f_bwd():
            torch.autograd.grad(y, (xr, b), go, retain_graph=True)
        def r_fwd():
            with torch.no_grad():
                O.bias_leaky_relu(x, b, 0.1)
        yr = O.bias_leaky_relu(xr, b, 0.1)
        def r_bwd():
            torch.autograd.grad(yr, (xr, b), go, retain_graph=True)
        t = [timeit(f_fwd), timeit(f_bwd), timeit(r_fwd), timeit(r_bwd)]
        mb = x.numel() * 2 / 1e6
        print(f"8x{C_}x{H}x{W} ({mb:6.1f} MB): fused fwd {t[0]:7.1f} us ({2*mb/t[0]*1e3/1e3:6.0f} GB/s)  bwd {t[1]:7.1f} us ({3*mb/t[1]:6.0f} GB/s) | torch fwd {t[2]:7.1f}  bwd {t[3]:7.1f}", flush=True)
        for k, v in zip(tot, t): tot[k] += v
print(tot)
