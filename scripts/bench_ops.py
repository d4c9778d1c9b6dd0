"""Op-level micro-benchmark on the GPU: every HIP kernel at the per-level shapes of BASELINE configs[1]/[4]
(B pairs of 436x1024, FPN pyramid 7x16 .. 109x256), timed with HIP events over batches of back-to-back launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ocflow_amd import ops

DEV = torch.device("cuda:0")
LEVELS = [(196, 7, 16), (128, 14, 32), (96, 28, 64), (64, 55, 128), (32, 109, 256)]


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters  # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
    cl = (sys.argv[3] if len(sys.argv) > 3 else "cl") == "cl"
    fmt = torch.channels_last if cl else torch.contiguous_format
    es = 2 if dtype == torch.bfloat16 else 4
    print(f"B={B} dtype={dtype} layout={'cl' if cl else 'nchw'}")
    tot_b = tot_t = 0.0
    rows = []
    for (C, H, W) in LEVELS:
        f1 = torch.randn(B, C, H, W, device=DEV).to(dtype).contiguous(memory_format=fmt)
        f2 = torch.randn(B, C, H, W, device=DEV).to(dtype).contiguous(memory_format=fmt)
        flow = torch.randn(B, 2, H, W, device=DEV) * 2
        occ = torch.rand(B, 1, H, W, device=DEV)
        g = torch.randn(B, 81, H, W, device=DEV).to(dtype).contiguous(memory_format=fmt)
        px = B * H * W
        with torch.no_grad():
            t_plain = timeit(lambda: ops.compute_cost_volume(f1, f2))
            t_occ = timeit(lambda: ops.corr_warp(f1, f2, None, occ))
            t_fused = timeit(lambda: ops.corr_warp(f1, f2, flow, occ, 4, 1.25, False))
            t_warp = timeit(lambda: ops.warp(f2, flow, False, 1.25))
        b_plain = px * (2 * C * es + 81 * es)
        b_occ = b_plain + px * 4
        b_fused = b_plain + px * 12
        # backward pieces through the raw launchers
        f1g, f2g = f1.clone().requires_grad_(), f2.clone().requires_grad_()
        out = ops.compute_cost_volume(f1g, f2g)
        t_cbwd = timeit(lambda: torch.autograd.grad(out, (f1g, f2g), g, retain_graph=True))
        imgg, flg = f2.clone().requires_grad_(), flow.clone().requires_grad_()
        wout = ops.warp(imgg, flg, False, 1.25)
        gw = torch.randn_like(wout)
        t_wbwd = timeit(lambda: torch.autograd.grad(wout, (imgg, flg), gw, retain_graph=True))
        rows.append((C, H, W, t_plain, b_plain / t_plain / 1e3, t_occ, b_occ / t_occ / 1e3, t_fused, b_fused / t_fused / 1e3, t_warp,
                     px * (2 * C * es + 8) / t_warp / 1e3, t_cbwd, px * (4 * C * es + 81 * es) / t_cbwd / 1e3, t_wbwd))
        tot_b += b_occ
        tot_t += t_occ
    print(f"{'level':>14} | {'corr us':>8} {'GB/s':>7} | {'corr+occ':>8} {'GB/s':>7} | {'fused':>8} {'GB/s':>7} | {'warp':>7} {'GB/s':>7} | "
          f"{'corr_bwd':>8} {'GB/s':>7} | {'warp_bwd':>8}")
    for r in rows:
        print(f"{r[0]:>4}x{r[1]:>3}x{r[2]:>4} | {r[3]:8.1f} {r[4]:7.0f} | {r[5]:8.1f} {r[6]:7.0f} | {r[7]:8.1f} {r[8]:7.0f} | {r[9]:7.1f} {r[10]:7.0f} | "
              f"{r[11]:8.1f} {r[12]:7.0f} | {r[13]:8.1f}")
    print(f"corr+occ all levels: {tot_b/1e6:.1f} MB in {tot_t:.1f} us = {tot_b/tot_t/1e3:.0f} GB/s ({tot_b/tot_t/1e3/8000*100:.1f}% of 8 TB/s)")
    # losses at full resolution
    H, W = 436, 1024
    imgs = torch.rand(B, 6, H, W, device=DEV) * 2 - 1
    flow = (torch.randn(B, 2, H, W, device=DEV) * 2).requires_grad_()
    occ = (torch.rand(B, 1, H, W, device=DEV) < 0.3).float()
    px = B * H * W
    with torch.no_grad():
        t = timeit(lambda: ops.warp_photometric_error(imgs[:, :3], imgs[:, 3:], flow, occ)); print(f"photo fwd   {t:8.1f} us {px*36/t/1e3:7.0f} GB/s")
        t = timeit(lambda: ops.smoothness_losses(imgs[:, :3], flow)); print(f"smooth fwd  {t:8.1f} us {px*20/t/1e3:7.0f} GB/s")
        t = timeit(lambda: ops.occlusion_from_flow(flow)); print(f"range+occ   {t:8.1f} us {px*16/t/1e3:7.0f} GB/s")
    L = ops.warp_photometric_error(imgs[:, :3], imgs[:, 3:], flow, occ)
    t = timeit(lambda: torch.autograd.grad(L, flow, retain_graph=True)); print(f"photo bwd   {t:8.1f} us {px*44/t/1e3:7.0f} GB/s")
    s1, s2 = ops.smoothness_losses(imgs[:, :3], flow)
    t = timeit(lambda: torch.autograd.grad(s1 + s2, flow, retain_graph=True)); print(f"smooth bwd  {t:8.1f} us {px*28/t/1e3:7.0f} GB/s")


if __name__ == "__main__":
    main()
