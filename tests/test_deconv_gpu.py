"""GPU parity of the small transposed-convolution kernels (ocflow_amd/csrc/deconv_small.hip) behind the estimators' `upconv1` / `upconv2`
(flow_net.py:38-48, flow_occ_net.py:46-66: ConvTranspose2d(Ci, Co, 3, 2, 1) called with output_size = the next level's map) against the
framework's fp32 transposed convolution on the SAME bf16-rounded operands.

Bar: fp32 accumulation, one bf16 rounding of the output (2^-8 relative): rtol 1e-2 + 1e-2 of the output scale; the weight and bias
gradients are fp32 sums (atomics, any order): relative L2 <= 2e-3."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [
    # B, ci, co, H, W, output_padding (y, x)
    (2, 2, 2, 7, 16, (1, 1)), (2, 16, 1, 7, 16, (1, 1)), (2, 32, 2, 14, 32, (1, 1)), (2, 1, 1, 14, 32, (1, 1)),
    (1, 2, 2, 14, 32, (0, 1)), (3, 32, 2, 28, 64, (0, 0)), (1, 16, 1, 55, 128, (0, 1)), (1, 64, 4, 5, 9, (1, 0)), (2, 3, 3, 1, 1, (1, 1)),
    (8, 32, 2, 55, 128, (0, 1)),
]


def _l2(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    return float((got - want).norm() / (want.norm() + 1e-30))


def _close(got, want, what, rtol=1e-2, atol_rel=1e-2):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = float(want.abs().max()) + 1e-30
    worst = float(((got - want).abs() - rtol * want.abs()).max())
    assert worst <= atol_rel * scale, f"{what}: worst excess {worst:.3e} at scale {scale:.3e}"


@pytest.mark.parametrize("wdt", [torch.float32, torch.bfloat16], ids=["w32", "w16"])
@pytest.mark.parametrize("B,ci,co,H,W,op", CASES)
def test_deconv_forward_backward_match_fp32_autograd(B, ci, co, H, W, op, wdt):
    from ocflow_amd import conv
    g = torch.Generator().manual_seed(ci * 100 + co)
    x = torch.randn(B, ci, H, W, generator=g).bfloat16()
    w = (torch.randn(ci, co, 3, 3, generator=g) * 0.3).to(wdt)
    b = torch.randn(co, generator=g) * 0.5
    Ho, Wo = 2 * H - 1 + op[0], 2 * W - 1 + op[1]
    gy = torch.randn(B, co, Ho, Wo, generator=g).bfloat16()
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True)
    assert conv.deconv_eligible(xd, wd, (2, 2), (1, 1), (1, 1), 1, op)
    y = conv.deconv3x3s2_bias(xd, wd, bd, op)
    assert y.shape == (B, co, Ho, Wo) and y.dtype == torch.bfloat16
    y.backward(gy.to(DEV).contiguous(memory_format=torch.channels_last))
    xr = x.float().requires_grad_(True)
    wr = w.bfloat16().float().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, br, 2, 1, op)
    yr.backward(gy.float())
    _close(y, yr, "forward")
    _close(xd.grad, xr.grad, "input gradient")
    assert wd.grad.dtype == wdt and bd.grad.dtype == torch.float32
    tol = 2e-3 if wdt == torch.float32 else 1e-2
    assert _l2(wd.grad, wr.grad) < tol, _l2(wd.grad, wr.grad)
    assert _l2(bd.grad, br.grad) < 2e-3, _l2(bd.grad, br.grad)


def test_deconv_reads_a_channel_slice_and_the_module_routes_here():
    """x as a channel slice of a wider pixel-major buffer (pitch 24 for 16 channels); `deconv_bias` of the networks' epilogue module
    takes the native path for the estimator's layer and the library's for anything else"""
    from ocflow_amd import conv, ops
    from ocflow_amd.models.networks import _epilogue
    torch.manual_seed(0)
    buf = torch.randn(2, 24, 9, 20, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    x = buf[:, 4:20]
    m = torch.nn.ConvTranspose2d(16, 1, 3, 2, 1).to(DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        seen = []
        old = conv.deconv3x3s2_bias
        conv.deconv3x3s2_bias = lambda *a: (seen.append(1), old(*a))[1]
        try:
            y = _epilogue.deconv_bias(m, x, output_size=(18, 39))
        finally:
            conv.deconv3x3s2_bias = old
    assert seen == [1] and y.shape == (2, 1, 18, 39)
    ref = F.conv_transpose2d(x.float(), m.weight.detach().bfloat16().float(), m.bias.detach(), 2, 1, (1, 0))
    _close(y, ref, "sliced input")
    m4 = torch.nn.ConvTranspose2d(16, 2, 4, 2, 1).to(DEV)  # the PWC estimators' 4x4 deconvolutions stay with the library
    assert not conv.deconv_eligible(x, m4.weight, m4.stride, m4.padding, m4.dilation, m4.groups, (0, 0))


@pytest.mark.parametrize("B,ci,co,H,W,op", [(2, 16, 1, 7, 16, (1, 1)), (2, 1, 1, 14, 32, (1, 1)), (1, 32, 2, 9, 13, (0, 1)), (8, 16, 1, 55, 128, (0, 1))])
def test_deconv_with_fused_sigmoid_matches_the_two_op_chain(B, ci, co, H, W, op):
    """`torch.sigmoid(conv_transpose2d(x) + bias)` (flow_occ_net.py:62-66) in the launches of the transposed convolution: forward = sigmoid of
    the bf16-rounded result, backward = the sigmoid derivative applied to the incoming gradient on the fly (bf16-rounded, as the framework's
    sigmoid_backward) -- against the unfused chain on the same kernels (forward within one bf16 rounding of the sigmoid's own evaluation,
    gradients 1e-2) and against fp32 autograd"""
    from ocflow_amd import conv
    g = torch.Generator().manual_seed(ci * 10 + co)
    x = torch.randn(B, ci, H, W, generator=g).bfloat16()
    w = torch.randn(ci, co, 3, 3, generator=g) * 0.3
    b = torch.randn(co, generator=g) * 0.5
    Ho, Wo = 2 * H - 1 + op[0], 2 * W - 1 + op[1]
    gy = torch.randn(B, co, Ho, Wo, generator=g).bfloat16().to(DEV).contiguous(memory_format=torch.channels_last)
    outs = []
    for fused in (True, False):
        xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        y = conv.deconv3x3s2_bias(xd, wd, bd, op, True) if fused else torch.sigmoid(conv.deconv3x3s2_bias(xd, wd, bd, op))
        y.backward(gy)
        outs.append((y, xd.grad, wd.grad, bd.grad))
    _close(outs[0][0], outs[1][0], "forward vs unfused", 1e-2, 4e-3)
    _close(outs[0][1], outs[1][1], "input gradient vs unfused")
    # (the two sigmoid evaluations may round a value to neighbouring bf16 numbers: y (1 - y) then differs by one bf16 ulp = 4e-3 there)
    assert _l2(outs[0][2], outs[1][2]) < 1e-2
    # the bias gradient is a sum of signed terms that largely cancel: its bar is relative to the sum of their magnitudes (<= |gy| / 4)
    bias_scale = float(gy.float().abs().sum()) * 0.25 / co
    assert float((outs[0][3] - outs[1][3]).abs().max()) < 2e-3 * bias_scale
    xr, wr, br = x.float().requires_grad_(True), w.bfloat16().float().requires_grad_(True), b.clone().requires_grad_(True)
    yr = torch.sigmoid(F.conv_transpose2d(xr, wr, br, 2, 1, op))
    yr.backward(gy.float().cpu())
    _close(outs[0][0], yr, "forward vs fp32")
    _close(outs[0][1], xr.grad, "input gradient vs fp32", 2e-2, 2e-2)
    assert _l2(outs[0][2], wr.grad) < 1e-2 and float((outs[0][3].cpu() - br.grad).abs().max()) < 2e-3 * bias_scale
