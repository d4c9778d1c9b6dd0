"""GPU parity: every HIP kernel, called through the C ABI (ocflow_amd.ops -> libocflow_hip.so), against
(i) the golden vectors generated from the reference and (ii) the CPU oracle on seeded inputs.

Tolerances (stated per north_star: 1e-4 fp32):  fp32 kernels 2e-5 abs/rel vs golden (different summation order
than ATen), bf16 kernels vs the oracle evaluated on the SAME bf16-rounded inputs in fp32: 2e-2 (output rounding).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ocflow_oracle as O

pytestmark = pytest.mark.gpu

T = torch.from_numpy
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from ocflow_amd import ops as _ops
    return _ops


def dev(a, grad=False):
    t = (T(a) if isinstance(a, np.ndarray) else a).to(DEV)
    return t.requires_grad_() if grad else t


def close(a, b, tol=2e-5):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol)


def layouts(t):
    """the same logical tensor as NCHW-contiguous, channels_last and a strided channel slice"""
    yield "nchw", t.contiguous()
    yield "cl", t.contiguous(memory_format=torch.channels_last)
    big = torch.zeros((t.shape[0], t.shape[1] + 3, t.shape[2], t.shape[3] + 2), device=t.device, dtype=t.dtype)
    big[:, 2:2 + t.shape[1], :, 1:1 + t.shape[3]] = t
    yield "view", big[:, 2:2 + t.shape[1], :, 1:1 + t.shape[3]]


# ------------------------------------------------------------------------------------------- cost volume
@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_corr_golden(ops, golden, tag):
    g = golden("corr")
    d = int(g[f"{tag}_d"])
    for (n1, f1), (n2, f2) in zip(layouts(dev(g[f"{tag}_f1"])), layouts(dev(g[f"{tag}_f2"]))):
        f1 = f1.detach().requires_grad_()
        f2 = f2.detach().requires_grad_()
        out = ops.compute_cost_volume(f1, f2, d)
        close(out, g[f"{tag}_out"])
        gf1, gf2 = torch.autograd.grad(out, (f1, f2), dev(g[f"{tag}_cot"]))
        close(gf1, g[f"{tag}_gf1"])
        close(gf2, g[f"{tag}_gf2"])


@pytest.mark.parametrize("shape,d", [((2, 32, 21, 45), 4), ((1, 196, 7, 16), 4), ((1, 64, 9, 70), 4), ((2, 16, 33, 33), 2),
                                      ((1, 40, 8, 32), 3), ((1, 3, 1, 1), 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cl", [False, True])
def test_corr_vs_oracle(ops, shape, d, dtype, cl):
    gen = torch.Generator().manual_seed(hash((shape, d)) & 0xffff)
    f1 = torch.randn(shape, generator=gen).to(dtype)
    f2 = torch.randn(shape, generator=gen).to(dtype)
    ref = O.compute_cost_volume(f1.float(), f2.float(), d)
    a, b = f1.to(DEV), f2.to(DEV)
    if cl:
        a, b = a.contiguous(memory_format=torch.channels_last), b.contiguous(memory_format=torch.channels_last)
    out = ops.compute_cost_volume(a, b, d)
    assert out.dtype == dtype and out.shape == ref.shape
    close(out, ref, 2e-5 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("ac", [False, True])
@pytest.mark.parametrize("with_occ", [False, True])
@pytest.mark.parametrize("cl", [False, True])
@pytest.mark.parametrize("fused", [False, True])
def test_corr_warp_fused(ops, ac, with_occ, cl, fused, monkeypatch):
    """corr_warp with a flow, both forms: warp kernel + correlation kernel (default) and the single fused launch"""
    monkeypatch.setattr(ops, "_FUSED_WARP", "1" if fused else "0")
    gen = torch.Generator().manual_seed(5)
    B, C, H, W = 2, 24, 19, 37
    f1 = torch.randn((B, C, H, W), generator=gen).requires_grad_()
    f2 = torch.randn((B, C, H, W), generator=gen).requires_grad_()
    flow = (torch.randn((B, 2, H, W), generator=gen) * 2.5).requires_grad_()
    occ = torch.rand((B, 1, H, W), generator=gen).requires_grad_() if with_occ else None
    cot = torch.randn((B, 81, H, W), generator=gen)
    ref = O.corr_warp(f1, f2, flow, occ, 4, 1.25, ac)
    ins = (f1, f2, flow) + ((occ,) if with_occ else ())
    gref = torch.autograd.grad(ref, ins, cot)

    fmt = torch.channels_last if cl else torch.contiguous_format
    a = f1.detach().to(DEV).contiguous(memory_format=fmt).requires_grad_()
    b = f2.detach().to(DEV).contiguous(memory_format=fmt).requires_grad_()
    fl = flow.detach().to(DEV).requires_grad_()
    oc = occ.detach().to(DEV).requires_grad_() if with_occ else None
    out = ops.corr_warp(a, b, fl, oc, 4, 1.25, ac)
    close(out, ref, 3e-5)
    gins = (a, b, fl) + ((oc,) if with_occ else ())
    gout = torch.autograd.grad(out, gins, cot.to(DEV))
    for x, y in zip(gout, gref):
        close(x, y, 1e-4)


@pytest.mark.parametrize("shape,d", [((2, 32, 21, 45), 4), ((1, 64, 30, 70), 4), ((1, 196, 7, 16), 4), ((2, 96, 9, 33), 4), ((1, 32, 12, 40), 2),
                                      ((1, 40, 8, 32), 3), ((1, 16, 5, 5), 1)])
@pytest.mark.parametrize("mode", ["occ", "warp", "warp+occ"])
@pytest.mark.parametrize("layout", ["cl", "nchw", "view"])
def test_corr_bf16_matrix_core_modes(ops, shape, d, mode, layout, monkeypatch):
    """bf16 runs on the MFMA band-GEMM kernel (corr_mfma.hip): every staging mode (occlusion on the outputs, fused warp gather)
    x every layout (dense channels_last fast path, NCHW, strided channel slice) x single / multi channel block / partial last
    block (C=196, C=40) / d < 4, against the oracle on the same bf16-rounded inputs.  Tolerance 2e-2: bf16 output rounding
    (+ the warp's bf16 rounding of the gathered features in the fused mode)."""
    monkeypatch.setattr(ops, "_FUSED_WARP", "1")  # this test is about the kernels' staging modes: keep the warp inside them
    gen = torch.Generator().manual_seed((hash((shape, d, mode)) & 0xffff) + 1)
    B, C, H, W = shape
    f1 = torch.randn(shape, generator=gen).bfloat16()
    f2 = torch.randn(shape, generator=gen).bfloat16()
    flow = torch.randn((B, 2, H, W), generator=gen) * 2.0 if "warp" in mode else None
    occ = torch.rand((B, 1, H, W), generator=gen) if "occ" in mode else None
    if flow is not None:
        w2 = O.warp(f2.float(), flow * 1.25, False).bfloat16().float()  # the kernel rounds the gathered features to bf16
    else:
        w2 = f2.float()
    ref = O.compute_cost_volume(f1.float(), w2 * occ if occ is not None else w2, d)

    def lay(t):
        t = t.to(DEV)
        if layout == "cl":
            return t.contiguous(memory_format=torch.channels_last)
        if layout == "view":
            big = torch.zeros((B, C + 5, H, W + 3), device=DEV, dtype=t.dtype).contiguous(memory_format=torch.channels_last)
            big[:, 3:3 + C, :, 2:2 + W] = t
            return big[:, 3:3 + C, :, 2:2 + W]
        return t.contiguous()

    out = ops.corr_warp(lay(f1), lay(f2), None if flow is None else flow.to(DEV), None if occ is None else occ.to(DEV), d, 1.25, False)
    assert out.dtype == torch.bfloat16 and out.shape == ref.shape
    close(out, ref, 2e-2)
    if flow is not None and layout != "nchw":
        # the default two-launch form (warp kernel, then the correlation kernel) produces the same cost volume bit for bit on the
        # matrix-core kernels: both round the gathered operand to bf16 before the contraction.  The unaligned channel slice ("view":
        # element loads only) is served by the vector-ALU kernel in the fused form (OCF_CORR_VECTOR_BF16) and by a matrix-core kernel
        # after the separate warp: same operands, another summation order -- equal within the output's bf16 rounding
        monkeypatch.setattr(ops, "_FUSED_WARP", "0")
        out2 = ops.corr_warp(lay(f1), lay(f2), flow.to(DEV), None if occ is None else occ.to(DEV), d, 1.25, False)
        if layout == "cl":
            assert torch.equal(out2, out)
        else:
            close(out2, out.float(), 1e-2)


def test_corr_bf16_output_into_concat_buffer(ops):
    """the stride ABI lets the kernel write the cost volume straight into a channel slice of a wider channels_last buffer"""
    gen = torch.Generator().manual_seed(77)
    f1 = torch.randn((2, 32, 16, 48), generator=gen).bfloat16()
    f2 = torch.randn((2, 32, 16, 48), generator=gen).bfloat16()
    ref = O.compute_cost_volume(f1.float(), f2.float(), 4)
    from ocflow_amd import _lib
    import ctypes as C
    a = f1.to(DEV).contiguous(memory_format=torch.channels_last)
    b = f2.to(DEV).contiguous(memory_format=torch.channels_last)
    buf = torch.zeros((2, 81 + 32 + 4, 16, 48), device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dst = buf[:, 0:81]
    T4 = lambda t: C.byref(_lib.Tensor4(t.data_ptr(), *t.stride()))
    for policy in (_lib.CORR_AUTO, _lib.CORR_BAND_ONLY):  # the wave kernel (any pixel pitch) and the band kernel's generic epilogue
        buf.zero_()
        st = _lib.load().ocf_corr_fwd(T4(a), T4(b), None, None, T4(dst), 2, 32, 16, 48, 4, 1.0, 0, 0.0, _lib.OCF_BF16, policy,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert st == 0
        torch.cuda.synchronize()
        close(buf[:, 0:81], ref, 2e-2)
        assert float(buf[:, 81:].abs().max()) == 0.0  # neighbours in the buffer untouched
    assert _lib.load().ocf_corr_fwd_variant(T4(a), T4(b), None, None, T4(dst), 2, 32, 16, 48, 4, _lib.OCF_BF16, 0) == _lib.CORR_VARIANT_WAVE
    # the operator surface: out= writes there (forward only: refused while a gradient is being recorded)
    buf.zero_()
    res = ops.corr_warp(a, b, None, None, 4, out=buf[:, 0:81])
    assert res.data_ptr() == buf.data_ptr() and float(buf[:, 81:].abs().max()) == 0.0
    close(res, ref, 2e-2)
    with pytest.raises(RuntimeError, match="forward-only"):
        ops.corr_warp(a.clone().requires_grad_(), b, None, None, 4, out=buf[:, 0:81])


@pytest.mark.parametrize("shape", [(2, 32, 21, 45), (1, 64, 12, 40), (1, 128, 14, 32), (1, 196, 7, 16)])
@pytest.mark.parametrize("cl", [True, False])
def test_corr_backward_bf16(ops, shape, cl):
    """bf16 backward (LDS-tiled kernels on channels_last, gather fallback otherwise) vs fp32 autograd of the oracle on the same
    bf16-rounded inputs; gradients are written in bf16 => 2e-2 of the gradient scale."""
    gen = torch.Generator().manual_seed(31)
    f1 = torch.randn(shape, generator=gen).bfloat16()
    f2 = torch.randn(shape, generator=gen).bfloat16()
    cot = torch.randn((shape[0], 81, shape[2], shape[3]), generator=gen).bfloat16()
    a, b = f1.float().requires_grad_(), f2.float().requires_grad_()
    g1r, g2r = torch.autograd.grad(O.compute_cost_volume(a, b, 4), (a, b), cot.float())
    fmt = torch.channels_last if cl else torch.contiguous_format
    x = f1.to(DEV).contiguous(memory_format=fmt).requires_grad_()
    y = f2.to(DEV).contiguous(memory_format=fmt).requires_grad_()
    g1, g2 = torch.autograd.grad(ops.compute_cost_volume(x, y, 4), (x, y), cot.to(DEV).contiguous(memory_format=fmt))
    for got, want in ((g1, g1r), (g2, g2r)):
        scale = float(want.abs().max())
        np.testing.assert_allclose(got.float().cpu().numpy(), want.numpy(), rtol=2e-2, atol=2e-2 * scale)


def test_corr_occ_only(ops):
    """flow_occ_net.py:167-168:  warped2 *= occ ; corr(feat1, warped2)  with warped2 already materialised"""
    gen = torch.Generator().manual_seed(6)
    f1 = torch.randn((1, 32, 14, 32), generator=gen)
    f2 = torch.randn((1, 32, 14, 32), generator=gen)
    occ = torch.rand((1, 1, 14, 32), generator=gen)
    close(ops.corr_warp(f1.to(DEV), f2.to(DEV), None, occ.to(DEV)), O.compute_cost_volume(f1, f2 * occ, 4))


def test_corr_errors(ops):
    f = torch.zeros((1, 4, 8, 8), device=DEV)
    with pytest.raises(ValueError):
        ops.compute_cost_volume(f, f, 5)
    with pytest.raises(ValueError):
        ops.compute_cost_volume(f, f[:, :2], 4)
    with pytest.raises(RuntimeError):
        ops.compute_cost_volume(f.cpu(), f.cpu(), 4)


# --------------------------------------------------------------------------------------------------- warp
@pytest.mark.parametrize("tag", ["main", "h1", "w1", "s22", "s11"])
@pytest.mark.parametrize("ac", [0, 1])
def test_warp_golden(ops, golden, tag, ac):
    g = golden("warp")
    for name, img in layouts(dev(g[f"{tag}_img"])):
        img = img.detach().requires_grad_()
        flow = dev(g[f"{tag}_flow"], True)
        out = ops.warp(img, flow, bool(ac))
        close(out, g[f"{tag}_out_ac{ac}"])
        gi, gf = torch.autograd.grad(out, (img, flow), dev(g[f"{tag}_cot"]))
        close(gi, g[f"{tag}_gimg_ac{ac}"])
        close(gf, g[f"{tag}_gflow_ac{ac}"], 5e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cl", [False, True])
def test_warp_vs_oracle(ops, dtype, cl):
    gen = torch.Generator().manual_seed(9)
    img = torch.randn((2, 40, 23, 31), generator=gen).to(dtype)
    flow = torch.randn((2, 2, 23, 31), generator=gen) * 4
    ref = O.warp(img.float(), flow * 0.625, False)
    a = img.to(DEV)
    if cl:
        a = a.contiguous(memory_format=torch.channels_last)
    out = ops.warp(a, flow.to(DEV), False, flow_scale=0.625)
    close(out, ref, 2e-5 if dtype == torch.float32 else 2e-2)


def test_warp_zero_flow_identity(ops):
    img = torch.randn((1, 3, 9, 11), device=DEV)
    close(ops.warp(img, torch.zeros((1, 2, 9, 11), device=DEV), True), img, 1e-6)


# ------------------------------------------------------------------------------------------------ losses
def test_losses_golden(ops, golden):
    g = golden("losses")
    img1, img2, occ = dev(g["img1"]), dev(g["img2"]), dev(g["occ"])
    for tag, o in (("none", None), ("occ", occ)):
        pred = dev(g["pred"], True)
        L = ops.photometric_error(pred, img1, o)
        close(L, g[f"photo_{tag}"], 1e-5)
        close(torch.autograd.grad(L, pred)[0], g[f"photo_{tag}_gpred"], 1e-5)
        flow = dev(g["flow"], True)
        L = ops.warp_photometric_error(img1, img2, flow, o)
        close(L, g[f"warpphoto_{tag}"], 1e-5)
        close(torch.autograd.grad(L, flow)[0], g[f"warpphoto_{tag}_gflow"], 1e-5)
    for itag, im in (("noise", img1), ("smooth", dev(g["simg"]))):
        flow = dev(g["flow"], True)
        s1, s2 = ops.smoothness_losses(im, flow)
        close(s1, g[f"smooth1_{itag}"], 1e-5)
        close(s2, g[f"smooth2_{itag}"], 1e-5)
        close(torch.autograd.grad(s1, flow, retain_graph=True)[0], g[f"smooth1_{itag}_gflow"], 1e-5)
        close(torch.autograd.grad(s2, flow)[0], g[f"smooth2_{itag}_gflow"], 1e-5)
    close(ops.photometric_error(img1, img1), np.float32(0.001), 1e-6)


def test_losses_strided_views(ops):
    """the training step passes views: imgs[:, 0:3], imgs[:, 3:6] of the [B,6,H,W] batch (models/model.py:322)"""
    gen = torch.Generator().manual_seed(12)
    imgs = torch.rand((2, 6, 20, 28), generator=gen) * 2 - 1
    flow = (torch.randn((2, 2, 20, 28), generator=gen) * 1.5).requires_grad_()
    occ = (torch.rand((2, 1, 20, 28), generator=gen) < 0.3).float()
    ref = O.photometric_error(O.warp(imgs[:, 3:6], flow, True), imgs[:, 0:3], occ)
    (gref,) = torch.autograd.grad(ref, flow)
    di = imgs.to(DEV)
    fl = flow.detach().to(DEV).requires_grad_()
    L = ops.warp_photometric_error(di[:, 0:3], di[:, 3:6], fl, occ.to(DEV))
    close(L, ref, 1e-5)
    close(torch.autograd.grad(L, fl)[0], gref, 1e-5)


@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 2, 3), (2, 5, 70), (1, 9, 64), (1, 33, 129), (3, 8, 65), (1, 17, 2)])
@pytest.mark.parametrize("layout", ["nchw", "pair_cl"])
def test_loss_kernels_odd_shapes(ops, shape, layout):
    """the column-walking smoothness kernels (8-row strips of 64 columns, register window, one-row lookahead), the photometric kernels and
    the fixed-point range map at shapes with partial strips, fewer rows / columns than the window, one-pixel maps -- both image layouts --
    against the oracle: losses, flow gradients, range map"""
    B_, H, W = shape
    gen = torch.Generator().manual_seed(H * 131 + W)
    img1, img2 = torch.rand((B_, 3, H, W), generator=gen) * 2 - 1, torch.rand((B_, 3, H, W), generator=gen) * 2 - 1
    flow = torch.randn((B_, 2, H, W), generator=gen) * 1.5
    occ = (torch.rand((B_, 1, H, W), generator=gen) < 0.3).float()
    fr = flow.clone().requires_grad_()
    s1r, s2r = O.first_order_smoothness_loss(img1, fr), O.second_order_smoothness_loss(img1, fr)
    pr = O.photometric_error(O.warp(img2, fr, True), img1, occ)
    d1, d2 = img1.to(DEV), img2.to(DEV)
    if layout == "pair_cl":
        pair = torch.cat((d1, d2), 1).contiguous(memory_format=torch.channels_last)
        d1, d2 = pair[:, 0:3], pair[:, 3:6]
    fd = flow.to(DEV).requires_grad_()
    s1, s2 = ops.smoothness_losses(d1, fd)
    p = ops.warp_photometric_error(d1, d2, fd, occ.to(DEV))
    for got, ref in ((s1, s1r), (s2, s2r), (p, pr)):
        if torch.isnan(ref):  # the mean over an empty set of differences (one-pixel axis): nan in the reference too
            assert torch.isnan(got.cpu())
        else:
            close(got, ref, 2e-5)
    terms = [(a, b) for a, b in ((s1, s1r), (s2, s2r), (p, pr)) if not torch.isnan(b)]
    if terms:
        (gd,) = torch.autograd.grad(sum(a for a, _ in terms), fd)
        (gr,) = torch.autograd.grad(sum(b for _, b in terms), fr)
        close(gd, gr, 5e-5)
    close(ops.compute_range_map(flow.to(DEV)), O.compute_range_map(flow), 1e-5)


def test_loss_arguments_off_their_defaults(ops):
    """edge-weight exponent `alpha` other than 100, a mask that occludes everything (the 1e-16 of the denominator decides), a mask that
    occludes nothing, fractional mask values"""
    gen = torch.Generator().manual_seed(77)
    img1, img2 = torch.rand((2, 3, 20, 31), generator=gen) * 2 - 1, torch.rand((2, 3, 20, 31), generator=gen) * 2 - 1
    flow = torch.randn((2, 2, 20, 31), generator=gen)
    for alpha in (10.0, 1.0, 250.0):
        fr, fd = flow.clone().requires_grad_(), flow.to(DEV).requires_grad_()
        ref = O.first_order_smoothness_loss(img1, fr, alpha) + 0.5 * O.second_order_smoothness_loss(img1, fr, alpha)
        s1, s2 = ops.smoothness_losses(img1.to(DEV), fd, alpha)
        close(s1 + 0.5 * s2, ref, 2e-5)
        close(torch.autograd.grad(s1 + 0.5 * s2, fd)[0], torch.autograd.grad(ref, fr)[0], 5e-5)
        close(ops.first_order_smoothness_loss(img1.to(DEV), flow.to(DEV), alpha), O.first_order_smoothness_loss(img1, flow, alpha), 2e-5)
    for occ in (torch.ones((2, 1, 20, 31)), torch.zeros((2, 1, 20, 31)), torch.rand((2, 1, 20, 31), generator=gen)):
        fr, fd = flow.clone().requires_grad_(), flow.to(DEV).requires_grad_()
        ref = O.photometric_error(O.warp(img2, fr, True), img1, occ)
        got = ops.warp_photometric_error(img1.to(DEV), img2.to(DEV), fd, occ.to(DEV))
        close(got, ref, 2e-5)
        close(torch.autograd.grad(got, fd)[0], torch.autograd.grad(ref, fr)[0], 5e-5)
        close(ops.photometric_error(img2.to(DEV), img1.to(DEV), occ.to(DEV)), O.photometric_error(img2, img1, occ), 2e-5)


def test_range_map_and_warp_at_exact_boundaries(ops):
    """targets exactly on pixel centres, on the last row / column, one pixel outside on every side, half a pixel outside, and non-finite
    flows: the range map (floor + four taps, taps outside the image dropped: models/model.py:223-305) and both warps (zeros padding)
    against the oracle"""
    H, W = 9, 13
    flow = torch.zeros((1, 2, H, W))
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    flow[0, 0] = (W - 1) - xs            # every pixel of a row lands exactly on the last column
    flow[0, 1, 0] = -1.0                 # first row: one pixel above the image
    flow[0, 1, 1] = float(H - 1) - 1.0   # second row: exactly the last row
    flow[0, 1, 2] = float(H) - 2.0       # third row: one pixel below the image
    flow[0, 1, 3] = -0.5 - 3.0           # fourth row: half a pixel above row 0
    flow[0, 0, 4] = -xs[4] - 0.5         # fifth row: half a pixel left of column 0
    flow[0, 0, 5] = -xs[5] - 1.0         # sixth row: exactly one pixel left
    flow[0, 0, 6] = 0.0
    flow[0, 1, 6] = 0.0                  # seventh row: identity
    flow[0, 0, 7] = 1e6                  # far outside
    close(ops.compute_range_map(flow.to(DEV)), O.compute_range_map(flow), 1e-6)
    img = torch.rand((1, 5, H, W), generator=torch.Generator().manual_seed(2))
    cot = torch.randn((1, 5, H, W), generator=torch.Generator().manual_seed(3))
    for ac in (False, True):
        close(ops.warp(img.to(DEV), flow.to(DEV), align_corners=ac), O.warp(img, flow, ac), 1e-5)
        # gradients at the same positions (integer sample coordinates: the one-sided derivative grid_sample takes)
        ir, fr = img.clone().requires_grad_(), flow.clone().requires_grad_()
        gi_ref, gf_ref = torch.autograd.grad(O.warp(ir, fr, ac), (ir, fr), cot)
        idv, fdv = img.to(DEV).requires_grad_(), flow.to(DEV).requires_grad_()
        gi, gf = torch.autograd.grad(ops.warp(idv, fdv, align_corners=ac), (idv, fdv), cot.to(DEV))
        close(gi, gi_ref, 1e-5)
        close(gf, gf_ref, 1e-4)
    nf = flow.clone()
    nf[0, 0, 8, 3] = float("nan")
    nf[0, 1, 8, 5] = float("inf")
    got, ref = ops.compute_range_map(nf.to(DEV)).cpu(), O.compute_range_map(nf)
    assert torch.isfinite(got).all() and torch.allclose(got, ref, atol=1e-6)  # non-finite sources contribute nothing (floor of nan / inf is dropped)


@pytest.mark.parametrize("C_", [1, 2, 5])
def test_loss_kernels_other_channel_counts(ops, C_):
    """images that are not RGB take the generic (runtime channel count) photometric and smoothness kernels"""
    gen = torch.Generator().manual_seed(40 + C_)
    img1, img2 = torch.rand((2, C_, 19, 37), generator=gen) * 2 - 1, torch.rand((2, C_, 19, 37), generator=gen) * 2 - 1
    flow = torch.randn((2, 2, 19, 37), generator=gen) * 1.5
    occ = (torch.rand((2, 1, 19, 37), generator=gen) < 0.3).float()
    fr = flow.clone().requires_grad_()
    ref = O.first_order_smoothness_loss(img1, fr) + O.second_order_smoothness_loss(img1, fr) + O.photometric_error(O.warp(img2, fr, True), img1, occ)
    (gr,) = torch.autograd.grad(ref, fr)
    fd = flow.to(DEV).requires_grad_()
    s1, s2 = ops.smoothness_losses(img1.to(DEV), fd)
    got = s1 + s2 + ops.warp_photometric_error(img1.to(DEV), img2.to(DEV), fd, occ.to(DEV))
    close(got, ref, 2e-5)
    close(torch.autograd.grad(got, fd)[0], gr, 5e-5)


# --------------------------------------------------------------------------------------------- range map
@pytest.mark.parametrize("tag", ["a", "b", "zero"])
def test_range_map_golden(ops, golden, tag):
    g = golden("range_map")
    flow = dev(g[f"{tag}_flow"])
    close(ops.compute_range_map(flow), g[f"{tag}_range"], 1e-5)
    close(ops.occlusion_from_flow(flow), g[f"{tag}_occ"], 1e-5)


def test_range_map_mass_conservation(ops):
    """size-independent property at the full Sintel size: sum(range) == sum of in-bounds bilinear weights"""
    gen = torch.Generator().manual_seed(3)
    flow = torch.randn((2, 2, 436, 1024), generator=gen) * 3
    rm = ops.compute_range_map(flow.to(DEV))
    ref = O.compute_range_map(flow)
    assert abs(float(rm.double().sum()) - float(ref.double().sum())) <= 1e-5 * float(ref.double().sum())
    close(rm, ref, 1e-4)


# ------------------------------------------------------------------------------------ feature normalise
@pytest.mark.parametrize("form", ["kernels", "kernels_channels_last", "composed", "strided_view"])
def test_normalize_golden(ops, golden, form, monkeypatch):
    """normalize_features (correlation_layer.py:59-82) against the reference golden, value and the exact gradient through mu and var:
    the kernel pair of csrc/norm.hip (dense NCHW and channels_last maps), the composed form (ocf_moments + framework passes) forced by
    OCFLOW_COMPOSED_NORMALIZE=1, and a strided view, which the kernels refuse and the composed form takes"""
    g = golden("norm")
    f1, f2 = dev(g["f1"], True), dev(g["f2"], True)
    a, b = f1, f2
    if form == "kernels_channels_last":
        a, b = (t.detach().contiguous(memory_format=torch.channels_last).requires_grad_() for t in (f1, f2))
    if form == "composed":
        monkeypatch.setenv("OCFLOW_COMPOSED_NORMALIZE", "1")
    if form == "strided_view":
        big = [torch.zeros((t.shape[0], t.shape[1] + 3, t.shape[2], t.shape[3]), device=DEV) for t in (f1, f2)]
        for bg, t in zip(big, (f1, f2)):
            bg[:, 1:1 + t.shape[1]] = t.detach()
        a, b = (bg[:, 1:1 + t.shape[1]].requires_grad_() for bg, t in zip(big, (f1, f2)))
    assert ops._norm_kernels_take([a.detach(), b.detach()]) == (form in ("kernels", "kernels_channels_last", "composed"))
    n1, n2 = ops.normalize_features([a, b])
    close(n1, g["n1"], 1e-5)
    close(n2, g["n2"], 1e-5)
    gf1, gf2 = torch.autograd.grad((n1 * dev(g["c1"])).sum() + (n2 * dev(g["c2"])).sum(), (a, b))
    close(gf1, g["gf1"], 5e-5)
    close(gf2, g["gf2"], 5e-5)


@pytest.mark.parametrize("tag,kw", [("per_image", dict(moments_across_images=False)), ("per_channel", dict(moments_across_channels=False)),
                                    ("per_channel_image", dict(moments_across_channels=False, moments_across_images=False)),
                                    ("center_only", dict(normalize=False)), ("scale_only", dict(center=False)),
                                    ("scale_only_per_image", dict(center=False, moments_across_images=False)),
                                    ("identity", dict(normalize=False, center=False))])
def test_normalize_flag_combinations(ops, golden, tag, kw):
    """normalize_features with the non-default flags of correlation_layer.py:42-82 against the reference's goldens (values, gradients)"""
    g = golden("norm_flags")
    a, b = dev(g["f1"], True), dev(g["f2"], True)
    n1, n2 = ops.normalize_features([a, b], **kw)
    close(n1, g[f"{tag}_n1"], 1e-5)
    close(n2, g[f"{tag}_n2"], 1e-5)
    ga, gb = torch.autograd.grad((n1 * dev(g["c1"])).sum() + (n2 * dev(g["c2"])).sum(), (a, b), allow_unused=True)
    close(ga, g[f"{tag}_gf1"], 1e-4)
    close(gb, g[f"{tag}_gf2"], 1e-4)


def test_normalize_kernels_bf16_and_batch_coupling(ops):
    """bf16 maps at a PWC pyramid shape: kernel pair vs the fp32 oracle on the same bf16-rounded inputs (forward to bf16 resolution,
    gradients to 2e-2 of their scale), and the batch coupling of F6 (ONE mean / variance for both maps and all samples)"""
    gen = torch.Generator().manual_seed(44)
    f1 = (torch.randn((4, 64, 24, 32), generator=gen) * 1.5 + 0.3).bfloat16()
    f2 = (torch.randn((4, 64, 24, 32), generator=gen) * 0.7 - 0.2).bfloat16()
    c1, c2 = torch.randn(f1.shape, generator=gen), torch.randn(f1.shape, generator=gen)
    a, b = f1.float().requires_grad_(), f2.float().requires_grad_()
    r1, r2 = O.normalize_features([a, b])
    g1, g2 = torch.autograd.grad((r1 * c1).sum() + (r2 * c2).sum(), (a, b))
    x = f1.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = f2.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    n1, n2 = ops.normalize_features([x, y])
    assert n1.dtype == torch.bfloat16
    close(n1, r1, 2e-2)
    close(n2, r2, 2e-2)
    gx, gy = torch.autograd.grad((n1.float() * c1.to(DEV)).sum() + (n2.float() * c2.to(DEV)).sum(), (x, y))
    for got, want in ((gx, g1), (gy, g2)):
        np.testing.assert_allclose(got.float().cpu().numpy(), want.numpy(), rtol=5e-2, atol=2e-2 * float(want.abs().max()))


# ------------------------------------------------------------------- full-size, size-independent properties
def test_full_size_properties(ops):
    """BASELINE configs[1] level-2 shape (B=1, C=32, 109x256, d=4): linearity in f2, centre channel identity,
    zero-flow fused warp == plain correlation (AC-True), bf16 path close to fp32 path."""
    gen = torch.Generator(device=DEV).manual_seed(0)
    shape = (1, 32, 109, 256)
    f1 = torch.randn(shape, device=DEV, generator=gen).contiguous(memory_format=torch.channels_last)
    f2 = torch.randn(shape, device=DEV, generator=gen).contiguous(memory_format=torch.channels_last)
    f3 = torch.randn(shape, device=DEV, generator=gen).contiguous(memory_format=torch.channels_last)
    cv = ops.compute_cost_volume(f1, f2)
    close(cv[:, 40], (f1 * f2).mean(1), 1e-5)
    close(ops.compute_cost_volume(f1, 2.0 * f2 + f3), 2.0 * cv + ops.compute_cost_volume(f1, f3), 2e-5)
    zf = torch.zeros((1, 2, 109, 256), device=DEV)
    close(ops.corr_warp(f1, f2, zf, None, 4, 1.0, True), cv, 1e-5)
    cvb = ops.compute_cost_volume(f1.bfloat16(), f2.bfloat16())
    close(cvb, ops.compute_cost_volume(f1.bfloat16().float(), f2.bfloat16().float()), 2e-2)
    # NCHW and channels_last agree bit for bit on the same data (same arithmetic order)
    assert torch.equal(ops.compute_cost_volume(f1.contiguous(), f2.contiguous()), cv.contiguous())


# --------------------------------------------------------------------------------------------------- SSIM
def test_ssim_golden(ops, golden):
    g = golden("ssim")
    a, b = dev(g["a"]), dev(g["b"])
    for ws in (4, 11):
        close(ops.ssim(a, b, ws, True), g[f"ssim_ws{ws}_mean"], 1e-5)
        close(ops.ssim(a, b, ws, False), g[f"ssim_ws{ws}_per_sample"], 1e-5)
    close(ops.ssim(a, a, 11), np.float32(1.0), 1e-6)
    # views of the [B,6,H,W] batch and a Sintel-sized frame against the oracle
    gen = torch.Generator().manual_seed(4)
    imgs = torch.rand((1, 6, 109, 256), generator=gen) * 2 - 1
    close(ops.ssim(imgs.to(DEV)[:, :3], imgs.to(DEV)[:, 3:], 11), O.ssim(imgs[:, :3], imgs[:, 3:], 11), 1e-5)
    with pytest.raises(ValueError):
        ops.ssim(a, b, 17)


def test_ssim_gradients_golden(ops, golden):
    """SSIM as a differentiable term: gradients w.r.t. both images against autograd of the reference's _ssim (odd window 11, the even
    window 4 with its (H+1) x (W+1) map), per-sample weights, and a Sintel-sized frame against autograd of the oracle"""
    g = golden("ssim")
    for ws in (4, 11):
        a, b = dev(g["a"], True), dev(g["b"], True)
        ga, gb = torch.autograd.grad(ops.ssim(a, b, ws, True), (a, b))
        close(ga, g[f"ssim_ws{ws}_grad_a"], 1e-5)
        close(gb, g[f"ssim_ws{ws}_grad_b"], 1e-5)
    a = dev(g["a"], True)
    (ga,) = torch.autograd.grad((ops.ssim(a, dev(g["b"]), 11, False) * dev(g["ssim_ws11_weights"])).sum(), a)
    close(ga, g["ssim_ws11_grad_a_weighted"], 1e-5)
    gen = torch.Generator().manual_seed(14)
    imgs = torch.rand((2, 6, 109, 256), generator=gen) * 2 - 1
    xa = imgs[:, :3].clone().requires_grad_()
    (gref,) = torch.autograd.grad(O.ssim(xa, imgs[:, 3:], 11), xa)
    da = imgs.to(DEV)[:, :3].clone().requires_grad_()
    (got,) = torch.autograd.grad(ops.ssim(da, imgs.to(DEV)[:, 3:], 11), da)
    np.testing.assert_allclose(got.cpu().numpy(), gref.numpy(), rtol=1e-4, atol=1e-9)


# ------------------------------------------------------------------------------------- convolution epilogue
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,cl", [((2, 32, 33, 47), True), ((2, 96, 17, 30), True), ((1, 12, 9, 21), True), ((3, 2, 16, 16), True),
                                      ((2, 32, 33, 47), False), ((2, 5, 20, 31), False), ((1, 1, 30, 40), False)])
def test_bias_leaky_relu_matches_unfused_chain(ops, shape, cl, dtype):
    """bias_leaky_relu == the oracle's `leaky_relu(x + bias)` (the two torch ops the reference's conv() epilogue runs): forward
    bit for bit in fp32 AND bf16 (same roundings), input gradient bit for bit, bias gradient within the reordering error of a
    fp32 sum (the oracle's bias gradient is additionally rounded to the activation dtype)"""
    gen = torch.Generator().manual_seed(shape[1] * 7 + shape[2])
    x0 = torch.randn(shape, generator=gen) * 2
    b0 = torch.randn(shape[1], generator=gen)
    go = torch.randn(shape, generator=gen)
    fmt = torch.channels_last if cl else torch.contiguous_format
    xr = x0.to(DEV).to(dtype).contiguous(memory_format=fmt).requires_grad_()
    br = b0.to(DEV).requires_grad_()
    ref = O.bias_leaky_relu(xr, br, 0.1)
    ref.backward(go.to(DEV).to(dtype).contiguous(memory_format=fmt))
    x = x0.to(DEV).to(dtype).contiguous(memory_format=fmt).requires_grad_()
    b = b0.to(DEV).requires_grad_()
    out = ops.bias_leaky_relu(x * 1, b, 0.1)  # `x * 1`: a fresh non-leaf tensor, as a convolution output is
    assert out.dtype == dtype and out.stride() == ref.stride()
    assert torch.equal(out, ref)
    out.backward(go.to(DEV).to(dtype).contiguous(memory_format=fmt))
    assert torch.equal(x.grad, xr.grad)
    n = shape[0] * shape[2] * shape[3]
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    np.testing.assert_allclose(b.grad.cpu().numpy(), br.grad.cpu().numpy(), rtol=tol, atol=tol * np.sqrt(n))
    # the fp32 bias gradient against an fp64 sum of the (exact) input gradient
    exact = xr.grad.double().sum((0, 2, 3)).cpu().numpy()
    np.testing.assert_allclose(b.grad.double().cpu().numpy(), exact, rtol=2e-5, atol=2e-5 * np.sqrt(n))


def test_bias_leaky_relu_full_size_and_errors(ops):
    """level-2 decoder shape of the benchmark (8 x 128 x 109 x 256, bf16, channels_last) + argument checking"""
    gen = torch.Generator().manual_seed(5)
    x0 = torch.randn((8, 128, 109, 256), generator=gen).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(128, generator=gen).to(DEV)
    ref = O.bias_leaky_relu(x0, b, 0.1)
    out = ops.bias_leaky_relu(x0.clone(memory_format=torch.preserve_format), b, 0.1)
    assert torch.equal(out, ref)
    with pytest.raises(ValueError):
        ops.bias_leaky_relu(x0, b[:5])
    with pytest.raises(RuntimeError, match="no CPU fallback|HIP"):
        ops.bias_leaky_relu(torch.zeros(1, 4, 8, 8), torch.zeros(4))
    with pytest.raises(RuntimeError):  # a strided view is not a convolution output: refused, not copied
        big = torch.zeros((2, 16, 8, 12), device=DEV)
        ops.bias_leaky_relu(big[:, :8], torch.zeros(8, device=DEV))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,cl", [((2, 2, 33, 47), True), ((2, 1, 17, 30), True), ((2, 32, 9, 21), True), ((2, 2, 20, 31), False)])
def test_bias_add_matches_reference(ops, shape, cl, dtype):
    """bias_add == `x + bias` (nn.Conv2d / nn.ConvTranspose2d bias term): forward bit for bit, the input gradient IS the incoming
    gradient, bias gradient = fp32 sum of it.  Two calls in a row check that the pooled zero accumulators are never reused."""
    gen = torch.Generator().manual_seed(shape[1] * 11 + shape[3])
    fmt = torch.channels_last if cl else torch.contiguous_format
    for rep in range(2):
        x0 = torch.randn(shape, generator=gen)
        b0 = torch.randn(shape[1], generator=gen)
        go = torch.randn(shape, generator=gen).to(DEV).to(dtype).contiguous(memory_format=fmt)
        xr = x0.to(DEV).to(dtype).contiguous(memory_format=fmt).requires_grad_()
        br = b0.to(DEV).requires_grad_()
        ref = O.bias_add(xr, br)
        ref.backward(go)
        x = x0.to(DEV).to(dtype).contiguous(memory_format=fmt).requires_grad_()
        b = b0.to(DEV).requires_grad_()
        out = ops.bias_add(x * 1, b)
        assert torch.equal(out, ref)
        out.backward(go)
        assert torch.equal(x.grad, xr.grad)
        exact = go.double().sum((0, 2, 3)).cpu().numpy()
        n = shape[0] * shape[2] * shape[3]
        np.testing.assert_allclose(b.grad.double().cpu().numpy(), exact, rtol=2e-5, atol=2e-5 * np.sqrt(n))


def test_bias_epilogue_inside_graph_capture(ops):
    """the pooled zero accumulators of the epilogue backward are re-zeroed by the graph they were captured in: replaying a
    captured forward+backward three times gives the same bias gradient every time (an accumulator zeroed outside the capture
    would keep growing)"""
    gen = torch.Generator().manual_seed(9)
    x0 = torch.randn((2, 32, 24, 40), generator=gen).to(DEV).contiguous(memory_format=torch.channels_last)
    b = torch.randn(32, generator=gen).to(DEV).requires_grad_()
    go = torch.randn((2, 32, 24, 40), generator=gen).to(DEV).contiguous(memory_format=torch.channels_last)
    def run():
        x = x0.clone(memory_format=torch.preserve_format).requires_grad_()
        out = ops.bias_leaky_relu(x * 1, b, 0.1)
        gx, gb = torch.autograd.grad(out, (x, b), go)
        return gx, gb
    ref_gx, ref_gb = run()       # eager: takes a slice of the eager pool, which the capture below must not continue
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gx, gb = run()
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(gx, ref_gx)
        np.testing.assert_allclose(gb.cpu().numpy(), ref_gb.cpu().numpy(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("shape", [(2, 32, 21, 45), (1, 96, 12, 40), (1, 196, 7, 16), (2, 64, 9, 33)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cl", [True, False])
def test_corr_occlusion_backward(ops, shape, dtype, cl):
    """out = corr(f1, f2 * occ) (flow_occ_net.py:167-168): gradients w.r.t. f1, f2 AND the occlusion map.  channels_last goes
    through ocf_corr_occ_bwd (mask applied while staging, d/docc as a by-product of the g2 pass, split channel blocks at the
    small shapes, 8-byte granularity at C=196); NCHW composes the same thing from ocf_corr_bwd + elementwise ops.  Reference:
    fp32 autograd of the oracle on the same (rounded) inputs; bf16 gradients are written in bf16 => 2e-2 of the scale."""
    gen = torch.Generator().manual_seed(shape[1] + 3)
    f1 = torch.randn(shape, generator=gen).to(dtype)
    f2 = torch.randn(shape, generator=gen).to(dtype)
    occ = torch.rand((shape[0], 1, shape[2], shape[3]), generator=gen)
    cot = torch.randn((shape[0], 81, shape[2], shape[3]), generator=gen).to(dtype)
    a, b, o = f1.float().requires_grad_(), f2.float().requires_grad_(), occ.clone().requires_grad_()
    ref = O.compute_cost_volume(a, b * o, 4)
    g1r, g2r, gor = torch.autograd.grad(ref, (a, b, o), cot.float())
    fmt = torch.channels_last if cl else torch.contiguous_format
    x = f1.to(DEV).contiguous(memory_format=fmt).requires_grad_()
    y = f2.to(DEV).contiguous(memory_format=fmt).requires_grad_()
    oc = occ.to(DEV).requires_grad_()
    out = ops.corr_warp(x, y, None, oc, 4)
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    close(out, ref, tol)
    g1, g2, go = torch.autograd.grad(out, (x, y, oc), cot.to(DEV).contiguous(memory_format=fmt))
    for got, want in ((g1, g1r), (g2, g2r), (go, gor)):
        scale = float(want.abs().max())
        np.testing.assert_allclose(got.float().cpu().numpy(), want.numpy(), rtol=tol * 5, atol=tol * scale)


def test_empty_batch(ops):
    """an empty batch goes through like in the reference (torch ops on empty tensors): empty outputs, no launch, no error"""
    f = torch.zeros((0, 32, 12, 20), device=DEV, requires_grad=True)
    fl = torch.zeros((0, 2, 12, 20), device=DEV)
    out = ops.compute_cost_volume(f, f, 4)
    assert out.shape == (0, 81, 12, 20)
    out.sum().backward()
    assert f.grad.shape == f.shape
    assert ops.warp(f, fl).shape == f.shape
    assert ops.corr_warp(f, f, fl, None, 4).shape == (0, 81, 12, 20)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_occ", [False, True])
def test_corr_backward_from_concat_slice(ops, dtype, with_occ):
    """in the decoders the cost volume is concatenated with other features, so its gradient arrives as a channel slice of the
    concat gradient (81 contiguous taps per pixel at a 117-element pixel pitch, 2-byte aligned): the tiled backward kernels
    consume that view directly (per-pixel tap windows, unaligned 8-byte loads) -- no dense copy"""
    gen = torch.Generator().manual_seed(17)
    B, C, H, W = 2, 32, 21, 45
    f1 = torch.randn((B, C, H, W), generator=gen).to(dtype)
    f2 = torch.randn((B, C, H, W), generator=gen).to(dtype)
    occ = torch.rand((B, 1, H, W), generator=gen) if with_occ else None
    extra = torch.randn((B, 36, H, W), generator=gen).to(dtype)
    cot = torch.randn((B, 81 + 36, H, W), generator=gen).to(dtype)
    a, b = f1.float().requires_grad_(), f2.float().requires_grad_()
    ref = torch.cat([O.compute_cost_volume(a, b * occ if with_occ else b, 4), extra.float()], 1)
    g1r, g2r = torch.autograd.grad(ref, (a, b), cot.float())
    x = f1.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = f2.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    out = torch.cat([ops.corr_warp(x, y, None, occ.to(DEV) if with_occ else None, 4), extra.to(DEV).contiguous(memory_format=torch.channels_last)], 1)
    g1, g2 = torch.autograd.grad(out, (x, y), cot.to(DEV).contiguous(memory_format=torch.channels_last))
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    for got, want in ((g1, g1r), (g2, g2r)):
        scale = float(want.abs().max())
        np.testing.assert_allclose(got.float().cpu().numpy(), want.numpy(), rtol=tol * 5, atol=tol * scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C_,slope", [(32, 0.1), (96, 0.1), (12, 0.1), (2, 1.0), (32, 1.0)])
def test_bias_epilogue_backward_from_concat_slice(ops, dtype, C_, slope):
    """the decoders concatenate every conv output with its input (dense blocks), so the epilogue's incoming gradient is a channel
    slice of the concat gradient: channel-contiguous pixels at a larger, 2-byte aligned pitch.  The backward kernel reads that
    view with unaligned 16-byte loads (no dense copy) -- same input gradient bit for bit, same bias gradient"""
    gen = torch.Generator().manual_seed(C_ * 3 + int(slope * 10))
    B, H, W, E = 2, 19, 23, 37
    x0 = torch.randn((B, C_, H, W), generator=gen).to(dtype)
    b0 = torch.randn(C_, generator=gen)
    extra = torch.randn((B, E, H, W), generator=gen).to(dtype)
    cot = torch.randn((B, E + C_, H, W), generator=gen).to(dtype)
    cl = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last)
    res = []
    for fn in (lambda t, bb: O.bias_leaky_relu(t, bb, slope), lambda t, bb: ops.bias_leaky_relu(t * 1, bb, slope)):
        x, b = cl(x0).requires_grad_(), b0.to(DEV).requires_grad_()
        out = torch.cat([cl(extra), fn(x, b)], 1)  # the slice starts at channel 37: odd element offset
        res.append(torch.autograd.grad(out, (x, b), cl(cot)))
    (gxr, gbr), (gx, gb) = res
    assert torch.equal(gx, gxr)
    n = B * H * W
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    np.testing.assert_allclose(gb.cpu().numpy(), gbr.float().cpu().numpy(), rtol=tol, atol=tol * np.sqrt(n))


@pytest.mark.parametrize("shape,d", [((1, 8, 5, 7), 1), ((2, 24, 9, 33), 2), ((1, 40, 8, 31), 3), ((2, 32, 4, 70), 4), ((1, 64, 1, 50), 4),
                                      ((1, 16, 50, 1), 4), ((1, 100, 13, 17), 4), ((3, 32, 17, 16), 4)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_occ", [False, True])
def test_corr_backward_odd_shapes(ops, shape, d, dtype, with_occ):
    """the tiled backward's staging (per-pixel tap windows, clamped branch-free loads, edge tiles, 1-pixel-wide / 1-pixel-high maps,
    d < 4, channel counts that are not a multiple of the tile's 32 / of the 16-byte unit) against fp32 autograd of the oracle"""
    gen = torch.Generator().manual_seed(sum(shape) * 7 + d)
    n = (2 * d + 1) ** 2
    f1 = torch.randn(shape, generator=gen).to(dtype)
    f2 = torch.randn(shape, generator=gen).to(dtype)
    occ = torch.rand((shape[0], 1, shape[2], shape[3]), generator=gen) if with_occ else None
    cot = torch.randn((shape[0], n, shape[2], shape[3]), generator=gen).to(dtype)
    a, b = f1.float().requires_grad_(), f2.float().requires_grad_()
    ref = O.compute_cost_volume(a, b * occ if with_occ else b, d)
    g1r, g2r = torch.autograd.grad(ref, (a, b), cot.float())
    cl = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last)
    x, y = cl(f1).requires_grad_(), cl(f2).requires_grad_()
    out = ops.corr_warp(x, y, None, occ.to(DEV) if with_occ else None, d)
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    close(out, ref, tol)
    g1, g2 = torch.autograd.grad(out, (x, y), cl(cot))
    for got, want in ((g1, g1r), (g2, g2r)):
        scale = float(want.abs().max()) + 1e-12
        np.testing.assert_allclose(got.float().cpu().numpy(), want.numpy(), rtol=tol * 5, atol=tol * scale)


@pytest.mark.parametrize("shape", [(2, 32, 21, 45), (1, 64, 12, 40), (1, 196, 7, 16), (1, 100, 13, 17), (1, 64, 1, 50), (1, 16, 50, 1),
                                   (3, 32, 17, 16), (2, 32, 4, 70), (1, 36, 9, 33), (2, 32, 70, 300), (3, 64, 40, 330)])
@pytest.mark.parametrize("with_occ", [False, True])
@pytest.mark.parametrize("sliced", [False, True])
def test_corr_backward_matrix_core_kernel(ops, shape, with_occ, sliced):
    """the band-GEMM backward (corr_bwd_mfma.hip, policy OCF_CORR_BWD_POLICY_MFMA = what the automatic policy picks whenever the layout
    allows) against the vector kernel (OCF_CORR_BWD_POLICY_VECTOR): edge tiles in both directions, 1-pixel-wide / 1-pixel-high maps, channel counts that are not a multiple of the 32-channel
    block (196, 100, 36), the gradient arriving as a 2-byte aligned channel slice (first and last pixels of the tensor: the clamped
    and shifted operand windows), occlusion (d/docc by-product); the last two shapes have more (tile, channel block) items than the
    pipelined kernel has workgroups (720 and 660 > 512: every workgroup runs its prefetch loop, across a channel-block change in the
    second).  Against fp32 autograd of the oracle, and against the vector kernel
    (same products, different summation order: within one bf16 rounding of the gradient scale)."""
    from ocflow_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(sum(shape) * 5 + with_occ)
    B, C_, H, W = shape
    f1 = torch.randn(shape, generator=gen).bfloat16()
    f2 = torch.randn(shape, generator=gen).bfloat16()
    occ = torch.rand((B, 1, H, W), generator=gen) if with_occ else None
    E = 36 if sliced else 0
    extra = torch.randn((B, E, H, W), generator=gen).bfloat16()
    cot = torch.randn((B, 81 + E, H, W), generator=gen).bfloat16()
    a, b = f1.float().requires_grad_(), f2.float().requires_grad_()
    o = occ.clone().requires_grad_() if with_occ else None
    ref = torch.cat([O.compute_cost_volume(a, b * o if with_occ else b, 4), extra.float()], 1)
    want = torch.autograd.grad(ref, (a, b, o) if with_occ else (a, b), cot.float())
    cl = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last)

    def run(policy):
        with ops.kernel_policy(corr_bwd=policy):
            x, y = cl(f1).requires_grad_(), cl(f2).requires_grad_()
            oc = occ.to(DEV).requires_grad_() if with_occ else None
            out = torch.cat([ops.corr_warp(x, y, None, oc, 4), cl(extra)], 1)
            return torch.autograd.grad(out, (x, y, oc) if with_occ else (x, y), cl(cot))

    got, vec = run(_lib.CORR_BWD_MFMA), run(_lib.CORR_BWD_VECTOR)
    for g_, v_, w_ in zip(got, vec, want):
        scale = float(w_.abs().max()) + 1e-12
        np.testing.assert_allclose(g_.float().cpu().numpy(), w_.numpy(), rtol=0.1, atol=2e-2 * scale)
        np.testing.assert_allclose(g_.float().cpu().numpy(), v_.float().cpu().numpy(), rtol=0, atol=1e-2 * scale)


def test_torch_library_ops(ops):
    """ocflow_amd/torch_ops.py: torch.ops.ocflow.cost_volume / warp are the same launches as ops.compute_cost_volume / ops.warp, pass
    torch.library.opcheck (schema, fake kernel, autograd registration) and trace under torch.compile without graph breaks"""
    import ocflow_amd.torch_ops  # noqa: F401
    gen = torch.Generator().manual_seed(12)
    f1 = torch.randn((2, 32, 12, 20), generator=gen).to(DEV).requires_grad_()
    f2 = torch.randn((2, 32, 12, 20), generator=gen).to(DEV).requires_grad_()
    fl = (torch.randn((2, 2, 12, 20), generator=gen) * 2).to(DEV).requires_grad_()
    cot = torch.randn((2, 81, 12, 20), generator=gen).to(DEV)
    out = torch.ops.ocflow.cost_volume(f1, f2, 4)
    ref = ops.compute_cost_volume(f1, f2, 4)
    assert torch.equal(out, ref)
    for a, b in zip(torch.autograd.grad(out, (f1, f2), cot), torch.autograd.grad(ref, (f1, f2), cot)):
        assert torch.equal(a, b)
    w, wr = torch.ops.ocflow.warp(f1, fl, 1.25, False), ops.warp(f1, fl, False, 1.25)
    assert torch.equal(w, wr)
    ga, gb = torch.autograd.grad(w, (f1, fl), torch.ones_like(w)), torch.autograd.grad(wr, (f1, fl), torch.ones_like(wr))
    np.testing.assert_allclose(ga[0].cpu().numpy(), gb[0].cpu().numpy(), rtol=1e-5, atol=1e-5)  # (float atomics: order differs)
    np.testing.assert_allclose(ga[1].cpu().numpy(), gb[1].cpu().numpy(), rtol=1e-4, atol=1e-4)
    torch.library.opcheck(torch.ops.ocflow.cost_volume.default, (f1.detach(), f2.detach(), 4), test_utils=("test_schema", "test_faketensor"))
    torch.library.opcheck(torch.ops.ocflow.warp.default, (f1.detach(), fl.detach(), 1.25, False), test_utils=("test_schema", "test_faketensor"))

    def fn(a, b, flow):
        return torch.ops.ocflow.cost_volume(a, torch.ops.ocflow.warp(b, flow, 1.25, False), 4).square().mean()

    compiled = torch.compile(fn, backend="aot_eager", fullgraph=True)
    assert abs(float(compiled(f1, f2, fl)) - float(fn(f1, f2, fl))) < 1e-6


def test_torch_library_loss_and_fused_ops(ops):
    """the other registered operators: ocflow::corr_warp (flow and occlusion optional), ocflow::photo_loss,
    ocflow::smooth_loss, ocflow::range_map -- same values and gradients as the reference-named entry points, opcheck (schema + fake
    kernels), and ocflow::cost_volume's backward is a stateless operator on the saved tensors (no forward re-run)"""
    import ocflow_amd.torch_ops  # noqa: F401
    gen = torch.Generator().manual_seed(13)
    B, C_, H, W = 2, 32, 12, 20
    f1 = torch.randn((B, C_, H, W), generator=gen).to(DEV).requires_grad_()
    f2 = torch.randn((B, C_, H, W), generator=gen).to(DEV).requires_grad_()
    fl = (torch.randn((B, 2, H, W), generator=gen) * 2).to(DEV).requires_grad_()
    oc = torch.rand((B, 1, H, W), generator=gen).to(DEV).requires_grad_()
    cot = torch.randn((B, 81, H, W), generator=gen).to(DEV)
    for flow, occ in ((fl, oc), (fl, None), (None, oc), (None, None)):
        out, _ = torch.ops.ocflow.corr_warp(f1, f2, flow, occ, 4, 1.25, False)
        ref = ops.corr_warp(f1, f2, flow, occ, 4, 1.25, False)
        assert torch.equal(out, ref)
        ins = [t for t in (f1, f2, flow, occ) if t is not None]
        for a, b in zip(torch.autograd.grad(out, ins, cot), torch.autograd.grad(ref, ins, cot)):
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-5)  # (warp backward: float atomics)
        torch.library.opcheck(torch.ops.ocflow.corr_warp.default, (f1.detach(), f2.detach(), None if flow is None else flow.detach(),
                                                                   None if occ is None else occ.detach(), 4, 1.25, False),
                              test_utils=("test_schema", "test_faketensor"))
    g1, g2 = torch.ops.ocflow.cost_volume_backward(f1.detach(), f2.detach(), cot, 4)
    r1, r2 = torch.autograd.grad(ops.compute_cost_volume(f1, f2, 4), (f1, f2), cot)
    assert torch.equal(g1, r1) and torch.equal(g2, r2)
    img = (torch.rand((B, 6, H, W), generator=gen) * 2 - 1).to(DEV)
    img1, img2 = img[:, :3], img[:, 3:]
    mask = (torch.rand((B, 1, H, W), generator=gen) < 0.3).float().to(DEV)
    for flow, occ in ((fl, mask), (fl, None)):
        s = torch.ops.ocflow.photo_loss(img1, img2, flow, occ)
        ref = ops.warp_photometric_error(img1, img2, flow, occ)
        got = s[0] / (s[1] * 3 + 1e-16) if occ is not None else s[0] / float(img1.numel())
        assert abs(float(got) - float(ref)) < 1e-6
        (ga,), (gb,) = torch.autograd.grad(got, (flow,)), torch.autograd.grad(ref, (flow,))
        np.testing.assert_allclose(ga.cpu().numpy(), gb.cpu().numpy(), rtol=1e-5, atol=1e-8)
        torch.library.opcheck(torch.ops.ocflow.photo_loss.default, (img1, img2, flow.detach(), occ), test_utils=("test_schema", "test_faketensor"))
    s = torch.ops.ocflow.smooth_loss(img1, fl, 100.0)
    r = ops._SmoothFn.apply(img1.float(), fl, 100.0)
    assert torch.equal(s, r)
    w = torch.tensor([0.3, 0.2, 0.4, 0.1], device=DEV)
    np.testing.assert_allclose(torch.autograd.grad((s * w).sum(), fl)[0].cpu().numpy(), torch.autograd.grad((r * w).sum(), fl)[0].cpu().numpy(),
                               rtol=1e-6, atol=1e-9)
    torch.library.opcheck(torch.ops.ocflow.smooth_loss.default, (img1, fl.detach(), 100.0), test_utils=("test_schema", "test_faketensor"))
    rng, occl = torch.ops.ocflow.range_map(fl.detach(), True)
    np.testing.assert_allclose(rng.cpu().numpy(), ops.compute_range_map(fl).cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(occl.cpu().numpy(), ops.occlusion_from_flow(fl).cpu().numpy(), rtol=1e-5, atol=1e-6)
    torch.library.opcheck(torch.ops.ocflow.range_map.default, (fl.detach(), True), test_utils=("test_schema", "test_faketensor"))


def test_compiled_flowoccnet_level_and_loss_on_registered_ops(ops):
    """torch.compile(fullgraph=True) of one whole level of FlowOccNet (flow_occ_net.py:151-180: warp -> occlusion estimator -> mask ->
    cost volume -> flow estimator) plus the unsupervised loss side (image warp + photometric + smoothness + range-map occlusion): under
    the tracer the reference-named entry points dispatch to torch.ops.ocflow.*, so the graph has no break; values and gradients equal
    the eager run"""
    import ocflow_amd.torch_ops  # noqa: F401
    from ocflow_amd.models.networks._epilogue import cat_aligned
    from ocflow_amd.models.networks.flow_net import FLOW_IN_CHANNELS, LEVELS, OpticalFlowEstimator
    from ocflow_amd.models.networks.flow_occ_net import OCC_IN_CHANNELS, OcclusionEstimator
    i = LEVELS.index(4)
    torch.manual_seed(5)
    occ_est = OcclusionEstimator(OCC_IN_CHANNELS[i], level=4).to(DEV)
    flow_est = OpticalFlowEstimator(FLOW_IN_CHANNELS[i], level=4).to(DEV)
    gen = torch.Generator().manual_seed(14)
    B, C_, H, W = 2, 96, 16, 24
    f1 = torch.randn((B, C_, H, W), generator=gen).to(DEV)
    f2 = torch.randn((B, C_, H, W), generator=gen).to(DEV)
    flow_up = (torch.randn((B, 2, H, W), generator=gen) * 0.5).to(DEV)
    extras = [torch.randn((B, 1, H, W), generator=gen).to(DEV) for _ in range(2)] + [torch.randn((B, 2, H, W), generator=gen).to(DEV)]
    imgs = (torch.rand((B, 6, 2 * H, 2 * W), generator=gen) * 2 - 1).to(DEV)

    def level(f1, f2, flow_up, occ_feat_up, occ_up, flow_feat_up, imgs):
        warped2 = ops.warp(f2, flow_up, False, flow_scale=20.0 / 16)
        occ, occ_up2, occ_feat_up2 = occ_est(cat_aligned([f1, warped2, occ_feat_up, occ_up]), 2 * H, 2 * W)
        corr = ops.corr_warp(f1, warped2, None, occ, 4)
        flow, flow_up2, _ = flow_est(cat_aligned([corr, f1, flow_up, flow_feat_up]), 2 * H, 2 * W)
        with torch.no_grad():
            mask = ops.occlusion_from_flow(-flow_up2)
        photo = ops.warp_photometric_error(imgs[:, :3], imgs[:, 3:], flow_up2, mask)
        s1, s2 = ops.smoothness_losses(imgs[:, :3], flow_up2)
        return 4.0 * photo + 0.5 * s1 + 0.1 * s2 + occ_up2.mean()

    params = list(occ_est.parameters()) + list(flow_est.parameters())
    args = (f1, f2, flow_up, *extras, imgs)
    eager = level(*args)
    g_eager = torch.autograd.grad(eager, params, allow_unused=True)
    compiled = torch.compile(level, backend="aot_eager", fullgraph=True)
    out = compiled(*args)
    assert abs(float(out) - float(eager)) < 1e-5 * max(1.0, abs(float(eager)))
    g_comp = torch.autograd.grad(out, params, allow_unused=True)
    for a, b in zip(g_comp, g_eager):
        assert (a is None) == (b is None)
        if a is not None:
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-3, atol=2e-4 * float(b.abs().max()) + 1e-9)


def test_psnr_golden(ops, golden):
    """ops.psnr (one reduction kernel over the batch) against the reference's per-image numpy psnr, incl. the identical-pair case
    (100) and a strided view of a [B,6,H,W] batch at a Sintel-sized frame against the oracle"""
    g = golden("psnr")
    a, b = (dev(g["a"]) + 1) * 127.5, (dev(g["b"]) + 1) * 127.5
    np.testing.assert_allclose(ops.psnr(a, b).cpu().numpy(), g["psnr"], rtol=1e-5, atol=1e-4)
    assert abs(float(ops.psnr(a, b, per_sample=False)) - float(g["psnr"].mean())) < 1e-3
    gen = torch.Generator().manual_seed(8)
    imgs = torch.rand((2, 6, 109, 256), generator=gen) * 255
    got = ops.psnr(imgs.to(DEV)[:, :3], imgs.to(DEV)[:, 3:]).cpu().numpy()
    want = [O.psnr(imgs[i, :3], imgs[i, 3:]) for i in range(2)]
    np.testing.assert_allclose(got, want, rtol=1e-5)
    cl = imgs.to(DEV).contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(ops.psnr(cl[:, :3], cl[:, 3:]).cpu().numpy(), want, rtol=1e-5)
    with pytest.raises(ValueError):
        ops.psnr(a, b[:, :2])


# ----------------------------------------------------------------- BatchNorm2d + LeakyReLU of the feature pyramid (csrc/bn.hip)
@pytest.mark.parametrize("shape", [(2, 16, 21, 45), (2, 196, 7, 16), (8, 32, 28, 64), (1, 96, 13, 17), (3, 4, 5, 6), (2, 128, 14, 32),
                                   (1, 1024, 3, 5)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batchnorm_lrelu_vs_framework(ops, shape, dtype):
    """ops.batchnorm_lrelu (two HIP passes per direction, last-block fp64 reduction of the statistics) against the reference's op chain
    nn.BatchNorm2d -> nn.LeakyReLU(0.1) (feature_pyramid_net.py:5-95) evaluated in fp32 on the same (rounded) input: output, running
    statistics and batch counter after two training calls, gradients w.r.t. input / gamma / beta, evaluation mode; 16-byte units
    (C % 8 == 0 bf16, C % 4 == 0 fp32), 8-byte units (C = 196 bf16, C = 4), 245-of-256-thread blocks (G = 49), more channels than
    threads (1024)."""
    import torch.nn as nn
    import torch.nn.functional as F
    B, C_, H, W = shape
    gen = torch.Generator().manual_seed(C_ + H)
    x0 = (torch.randn(shape, generator=gen) * 1.7 + 0.4).to(dtype)
    cot = torch.randn(shape, generator=gen).to(dtype)
    bn_ref, bn = nn.BatchNorm2d(C_), nn.BatchNorm2d(C_).to(DEV)
    with torch.no_grad():
        w = torch.rand(C_, generator=gen) + 0.5
        b = torch.randn(C_, generator=gen) * 0.3
        for m in (bn_ref, bn):
            m.weight.copy_(w); m.bias.copy_(b)
    tol = 2e-5 if dtype == torch.float32 else 1.6e-2
    for it in range(2):  # twice: the ticket counter of the workspace must come back to zero, the running statistics accumulate
        xr = x0.float().requires_grad_()
        yr = F.leaky_relu(bn_ref(xr), 0.1)
        gxr, gwr, gbr = torch.autograd.grad(yr, (xr, bn_ref.weight, bn_ref.bias), cot.float())
        x = x0.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
        assert ops.batchnorm_lrelu_supported(x, bn)
        y = ops.batchnorm_lrelu(x, bn, 0.1)
        assert y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
        gx, gw, gb = torch.autograd.grad(y, (x, bn.weight, bn.bias), cot.to(DEV).contiguous(memory_format=torch.channels_last))
        close(y, yr, tol)
        np.testing.assert_allclose(gx.float().cpu().numpy(), gxr.numpy(), rtol=tol * 5, atol=tol * float(gxr.abs().max()) * 2)
        n = B * H * W
        np.testing.assert_allclose(gw.cpu().numpy(), gwr.numpy(), rtol=tol * 5, atol=tol * np.sqrt(n) * 3)
        np.testing.assert_allclose(gb.cpu().numpy(), gbr.numpy(), rtol=tol * 5, atol=tol * np.sqrt(n) * 3)
        np.testing.assert_allclose(bn.running_mean.cpu().numpy(), bn_ref.running_mean.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(bn.running_var.cpu().numpy(), bn_ref.running_var.numpy(), rtol=1e-5, atol=1e-6)
        assert int(bn.num_batches_tracked) == it + 1 == int(bn_ref.num_batches_tracked)
    bn.eval(); bn_ref.eval()
    with torch.no_grad():
        close(ops.batchnorm_lrelu(x0.to(DEV).contiguous(memory_format=torch.channels_last), bn, 0.1), F.leaky_relu(bn_ref(x0.float()), 0.1), tol)
    assert int(bn.num_batches_tracked) == 2


@pytest.mark.parametrize("shape,groups", [((2, 16, 21, 45), 1), ((4, 196, 7, 16), 2), ((8, 32, 28, 64), 2), ((3, 4, 5, 6), 1), ((6, 64, 55, 128), 2)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batchnorm_backward_without_the_output_is_the_same_backward(ops, shape, groups, dtype, monkeypatch):
    """ocf_bn_act_bwd_x (the LeakyReLU mask recomputed from x with the forward's coefficients) against ocf_bn_act_bwd (the mask read from the
    saved output): bit-identical input and parameter gradients, column kernel and two-pass kernels, one and two statistics groups"""
    import torch.nn as nn
    B, C_, H, W = shape
    gen = torch.Generator().manual_seed(C_ * 3 + H)
    x0 = (torch.randn(shape, generator=gen) * 1.7 + 0.4).to(dtype)
    cot = torch.randn(shape, generator=gen).to(dtype).to(DEV).contiguous(memory_format=torch.channels_last)
    res = []
    for with_y in ("1", "0"):
        monkeypatch.setenv("OCFLOW_BN_BWD_Y", with_y)
        bn = nn.BatchNorm2d(C_).to(DEV)
        with torch.no_grad():
            bn.weight.copy_(torch.rand(C_, generator=torch.Generator().manual_seed(1)) + 0.5)
            bn.bias.copy_(torch.randn(C_, generator=torch.Generator().manual_seed(2)) * 0.3)
        x = x0.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
        y = ops.batchnorm_lrelu(x, bn, 0.1, groups)
        res.append(torch.autograd.grad(y, (x, bn.weight, bn.bias), cot))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("shape", [(4, 16, 21, 45), (4, 196, 7, 16), (16, 32, 28, 64), (2, 128, 14, 32), (6, 64, 55, 128)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batchnorm_lrelu_two_statistics_groups(ops, shape, dtype):
    """ops.batchnorm_lrelu(groups=2) -- both images of a pair as one batch -- is the module applied to the two halves in turn: per-half
    statistics, running buffers updated half after half, parameter gradients summed.  Bit-identical to two groups=1 calls of the same
    kernels (same partial sums, same order), and within tolerance of the framework chain on the halves."""
    import copy
    import torch.nn as nn
    import torch.nn.functional as F
    B, C_, H, W = shape
    gen = torch.Generator().manual_seed(B + C_)
    x0 = (torch.randn(shape, generator=gen) * 1.3 - 0.2).to(dtype)
    x0[B // 2:] += 0.7  # the halves differ in their statistics
    cot = torch.randn(shape, generator=gen).to(dtype)
    bn = nn.BatchNorm2d(C_).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C_, generator=gen) + 0.5); bn.bias.copy_(torch.randn(C_, generator=gen) * 0.3)
    bn_a, bn_b = copy.deepcopy(bn), copy.deepcopy(bn)
    cl = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last)
    x = cl(x0).requires_grad_()
    y = ops.batchnorm_lrelu(x, bn, 0.1, groups=2)
    gx, gw, gb = torch.autograd.grad(y, (x, bn.weight, bn.bias), cl(cot))
    # two calls of the same kernels on the halves
    xs = [cl(x0[:B // 2]).requires_grad_(), cl(x0[B // 2:]).requires_grad_()]
    ys = [ops.batchnorm_lrelu(xs[0], bn_a, 0.1), ops.batchnorm_lrelu(xs[1], bn_a, 0.1)]
    ga = [torch.autograd.grad(ys[k], (xs[k], bn_a.weight, bn_a.bias), cl(cot[k * (B // 2):(k + 1) * (B // 2)])) for k in range(2)]
    assert torch.equal(y, torch.cat(ys, 0)) and torch.equal(gx, torch.cat([ga[0][0], ga[1][0]], 0))
    assert torch.equal(gw, ga[0][1] + ga[1][1]) and torch.equal(gb, ga[0][2] + ga[1][2])
    # (the second half's statistics pass through fp32 on their way to the running update: last-bit differences)
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), bn_a.running_mean.cpu().numpy(), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), bn_a.running_var.cpu().numpy(), rtol=1e-6, atol=1e-8)
    assert int(bn.num_batches_tracked) == 2 == int(bn_a.num_batches_tracked)
    # the framework chain on the halves (fp32)
    bn_b = bn_b.cpu()
    yr = torch.cat([F.leaky_relu(bn_b(x0[:B // 2].float()), 0.1), F.leaky_relu(bn_b(x0[B // 2:].float()), 0.1)], 0)
    tol = 2e-5 if dtype == torch.float32 else 1.6e-2
    close(y, yr, tol)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), bn_b.running_var.numpy(), rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        ops.batchnorm_lrelu(x, bn, 0.1, groups=3)


def test_batchnorm_lrelu_framework_path_for_other_layouts(ops):
    """layouts / modes the kernels do not take go through the framework's own operators: same values as the module chain, bit for bit"""
    import torch.nn as nn
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(4)
    bn = nn.BatchNorm2d(6).to(DEV)
    x = torch.randn((2, 6, 5, 7), generator=gen).to(DEV)  # NCHW, C % 4 != 0
    assert not ops.batchnorm_lrelu_supported(x, bn)
    import copy
    bn2 = copy.deepcopy(bn)
    assert torch.equal(ops.batchnorm_lrelu(x, bn, 0.1), F.leaky_relu(bn2(x), 0.1))
    assert torch.equal(bn.running_var, bn2.running_var)
    bn8 = nn.BatchNorm2d(8, momentum=None).to(DEV)  # cumulative moving average: not taken
    assert not ops.batchnorm_lrelu_supported(torch.zeros((2, 8, 4, 4), device=DEV).contiguous(memory_format=torch.channels_last), bn8)


# ----------------------------------------------------------------- 3x3 convolution weight gradient (first slice)
@pytest.mark.parametrize("shape,d", [((2, 32, 16, 21, 45), 1), ((1, 128, 128, 9, 70), 1), ((2, 96, 64, 17, 33), 2), ((1, 64, 32, 40, 64), 4),
                                      ((1, 16, 8, 5, 130), 1), ((1, 128, 96, 35, 40), 16), ((3, 40, 24, 7, 9), 8), ((1, 8, 160, 6, 66), 1)])
@pytest.mark.parametrize("version", [1, 2])
def test_conv3x3_wgrad_vs_fp32_reference(ops, shape, d, version):
    """ocf_conv3x3_wgrad (both kernel versions: register-staged, and the LDS-DMA ring) against the fp32 weight gradient of the same convolution on the
    bf16-rounded operands: every dilation of the context network, partial 32-channel blocks (Ci = 40, Co = 24 / 96 / 160 > 128),
    widths that are not a multiple of the 64-pixel k-tile, accumulate-into semantics"""
    B, Ci, Co, H, W = shape
    gen = torch.Generator().manual_seed(Ci * 3 + Co + d)
    x = torch.randn((B, Ci, H, W), generator=gen).bfloat16().to(DEV).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((B, Co, H, W), generator=gen).bfloat16().to(DEV).contiguous(memory_format=torch.channels_last)
    w = torch.zeros((Co, Ci, 3, 3), device=DEV)
    ref = torch.ops.aten.convolution_backward(gy.float(), x.float(), w, None, (1, 1), (d, d), (d, d), False, (0, 0), 1, [False, True, False])[1]
    from ocflow_amd import _lib
    lib = _lib.load()
    with ops.kernel_policy(wgrad=version):
        got = ops.conv3x3_wgrad(x, gy, d)
        assert got.dtype == torch.float32 and got.shape == ref.shape
        np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=2e-4, atol=2e-4 * float(ref.abs().max()))
        again = ops.conv3x3_wgrad(x, gy, d, out=got.clone())  # accumulates
        np.testing.assert_allclose(again.cpu().numpy(), 2 * ref.cpu().numpy(), rtol=2e-4, atol=4e-4 * float(ref.abs().max()))


def test_conv3x3_autograd_path_matches_framework(ops):
    """ops.conv3x3 (framework forward + input gradient, hand-written weight gradient) against F.conv2d autograd in bf16: same output, same
    input gradient bit for bit, weight gradient within bf16 rounding of the fp32-accumulated value"""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(3)
    x0 = torch.randn((2, 64, 30, 50), generator=gen).bfloat16().to(DEV).contiguous(memory_format=torch.channels_last)
    w0 = (torch.randn((96, 64, 3, 3), generator=gen) * 0.05).bfloat16().to(DEV).contiguous(memory_format=torch.channels_last)
    go = torch.randn((2, 96, 30, 50), generator=gen).bfloat16().to(DEV).contiguous(memory_format=torch.channels_last)
    res = []
    for fn in (lambda a, b: F.conv2d(a, b, None, 1, 2, 2), lambda a, b: ops.conv3x3(a, b, 2)):
        x, w = x0.clone().requires_grad_(), w0.clone().requires_grad_()
        assert ops.conv3x3_wgrad_eligible(x, w, (1, 1), (2, 2), (2, 2), 1)
        y = fn(x, w)
        gx, gw = torch.autograd.grad(y, (x, w), go)
        res.append((y, gx, gw))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert res[1][2].dtype == torch.bfloat16 and res[1][2].stride() == w0.stride()
    scale = float(res[0][2].float().abs().max())
    np.testing.assert_allclose(res[1][2].float().cpu().numpy(), res[0][2].float().cpu().numpy(), rtol=2e-2, atol=1e-2 * scale)
    assert not ops.conv3x3_wgrad_eligible(x0.float(), w0.float(), (1, 1), (1, 1), (1, 1), 1)
    assert not ops.conv3x3_wgrad_eligible(x0, w0, (2, 2), (1, 1), (1, 1), 1)


@pytest.mark.gpu
def test_concat_channels_matches_torch_cat_and_pads_with_zeros():
    """ocf_concat_channels: the decoder-input concatenations (81 + C + 2 + 2 channels, parts that are views of wider tensors or halves of a
    paired batch) in one launch, zero-padded to a multiple of 32 channels; values are copied, so the bar is equality"""
    from ocflow_amd import ops
    torch.manual_seed(3)
    B, H, W = 3, 13, 21
    corr = torch.randn(B, 81, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    pair = torch.randn(2 * B, 32, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    f1 = pair.view(2, B, 32, H, W)[0]
    wide = torch.randn(B, 8, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    flow, feat = wide[:, 1:3], wide[:, 5:7]
    parts = [corr, f1, flow, feat]
    assert ops.concat_channels_eligible(parts)
    leaves = [p.detach().clone().requires_grad_(True) for p in parts]
    out = ops.concat_channels(leaves, 128)
    ref = torch.cat([p.detach() for p in parts], 1)
    assert out.shape == (B, 128, H, W) and out.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(out[:, :117], ref) and float(out[:, 117:].abs().max()) == 0.0
    g = torch.randn_like(out)
    out.backward(g)
    c0 = 0
    for leaf in leaves:
        assert torch.equal(leaf.grad, g[:, c0:c0 + leaf.shape[1]])
        c0 += leaf.shape[1]
    # aligned parts take the vector path; a single part; one-pixel maps
    a = torch.randn(2, 64, 5, 7, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    b = torch.randn(2, 40, 5, 7, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    assert torch.equal(ops.concat_channels([a, b])[:, :104], torch.cat([a, b], 1))
    one = torch.randn(2, 24, 1, 1, device="cuda").bfloat16()
    assert torch.equal(ops.concat_channels([one, one], 64)[:, :48], torch.cat([one, one], 1))


@pytest.mark.gpu
@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_adam_flat_matches_torch_adam(wd):
    """ocf_adam_flat (one launch over flat parameter / gradient / moment buffers, bf16 copy written along) against torch.optim.Adam on the
    same values for five steps: parameters and both moments within 2e-6 relative (fp32 rounding of a different operation order), the
    bf16 copy = the rounded fp32 parameters bit for bit"""
    import ctypes as C
    from ocflow_amd import _lib
    torch.manual_seed(3)
    n = 64 * 1000
    p0 = torch.randn(n, device=DEV)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref_p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pb = torch.zeros(n, device=DEV, dtype=torch.bfloat16)
    step = torch.zeros((), device=DEV)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    for it in range(5):
        g = torch.randn(n, device=DEV) * (0.1 + it)
        ref_p.grad = g.clone()
        opt.step()
        step += 1
        _lib.check(_lib.load().ocf_adam_flat(ptr(p), ptr(g), ptr(m), ptr(v), ptr(pb), n, 1e-3, 0.9, 0.999, 1e-8, wd, ptr(step),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "ocf_adam_flat")
        st = opt.state[ref_p]
        for got, want, what in ((p, ref_p.detach(), "param"), (m, st["exp_avg"], "exp_avg"), (v, st["exp_avg_sq"], "exp_avg_sq")):
            err = float((got - want).abs().max() / (want.abs().max() + 1e-30))
            assert err < 2e-6, (it, what, err)
        assert torch.equal(pb, p.bfloat16())


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["cl", "nchw"])
@pytest.mark.parametrize("shape", [(2, 20, 32), (3, 7, 9), (1, 436, 1024)])
def test_pair_input_is_cat_and_cast(ops, shape, layout):
    """ocf_pair_input: the two 3-channel halves of a [B, 6, H, W] fp32 tensor as one [2B, 3, H, W] bf16 channels_last batch, bit for bit
    `torch.cat((x[:, 0:3], x[:, 3:6]), 0).bfloat16()`; both input layouts, pixel counts that are / are not multiples of four"""
    B, H, W = shape
    x = torch.randn(B, 6, H, W, device=DEV)
    if layout == "cl":
        x = x.contiguous(memory_format=torch.channels_last)
    x1, x2 = x[:, 0:3], x[:, 3:6]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert ops.pair_input_eligible(x1, x2)
        got = ops.pair_input(x1, x2)
    assert not ops.pair_input_eligible(x1, x2)  # (no autocast: the encoder takes the fp32 tensors as they are)
    want = torch.cat((x1, x2), 0).bfloat16()
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("dtypes", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16)],
                         ids=["f32", "bf16", "f32flow-bf16occ"])
@pytest.mark.parametrize("shape", [(2, 5, 7), (1, 1, 1), (3, 2, 1), (2, 9, 16), (8, 109, 256)])
def test_flow_occ_head_loss_matches_upsample_l1_bce(ops, shape, dtypes):
    """ocf_flow_occ_head_fwd / _bwd against the framework's chain on the same quarter-resolution tensors:
    `F.l1_loss(up4(flow).float(), flow_gt)`, `F.binary_cross_entropy(up4(occ).float(), occ_gt)` with nn.Upsample(scale_factor=4, bilinear),
    forward (1e-5 relative: fp64 totals here, fp32 tree sums there) and the gradients w.r.t. the quarter-resolution tensors (fp32: 1e-5 of
    the scale; bf16: one rounding, 1e-2), image borders, 1-pixel maps, saturated probabilities (exactly 0 and 1: the -100 clamp of the logs
    and the 1e-12 clamp of the BCE gradient) and different loss weights included"""
    B, h, w = shape
    fdt, odt = dtypes
    g = torch.Generator().manual_seed(B * 100 + h)
    flow = (torch.randn(B, 2, h, w, generator=g) * 3).to(fdt)
    occ = torch.sigmoid(torch.randn(B, 1, h, w, generator=g) * 4).to(odt)
    occ.view(-1)[0] = 1.0
    occ.view(-1)[-1] = 0.0
    fgt = torch.randn(B, 2, 4 * h, 4 * w, generator=g) * 3
    ogt = (torch.rand(B, 1, 4 * h, 4 * w, generator=g) < 0.3).float()
    up = torch.nn.Upsample(scale_factor=4, mode="bilinear")
    fr = flow.to(DEV).requires_grad_(True)
    orr = occ.to(DEV).requires_grad_(True)
    l1r = F.l1_loss(up(fr.float()), fgt.to(DEV))
    bcr = F.binary_cross_entropy(up(orr.float()), ogt.to(DEV))
    (0.7 * l1r + 1.3 * bcr).backward()
    fd = flow.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    od = occ.to(DEV).requires_grad_(True)
    assert ops.flow_occ_head_eligible(fd, od, fgt.to(DEV), ogt.to(DEV))
    l1, bc = ops.flow_occ_head_loss(fd, od, fgt.to(DEV), ogt.to(DEV))
    (0.7 * l1 + 1.3 * bc).backward()
    assert abs(float(l1) - float(l1r)) <= 1e-5 * abs(float(l1r)) + 1e-7, (float(l1), float(l1r))
    assert abs(float(bc) - float(bcr)) <= 1e-5 * abs(float(bcr)) + 1e-7, (float(bc), float(bcr))
    for got, want, dt, what in ((fd.grad, fr.grad, fdt, "flow"), (od.grad, orr.grad, odt, "occ")):
        assert got.dtype == dt and got.shape == want.shape
        tol = 1e-5 if dt == torch.float32 else 1e-2
        scale = float(want.float().abs().max()) + 1e-30
        err = float((got.float() - want.float()).abs().max())
        assert err <= tol * scale, (what, err, scale)
