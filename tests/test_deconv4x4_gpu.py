"""ocf_deconv4x4s2_fwd (csrc/deconv4x4.hip): the PWC-style decoders' `upfeat` layers -- ConvTranspose2d(Ci, 2, 4, 2, 1) on a channel range of
the dense-block buffer (cost_volume_flow_net.py:63-65,182-183) -- against the framework's transposed convolution on the SAME bf16 operands in
fp32; the backward is the framework's own and must equal the library layer's gradients."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (B, Ci, pitch, Co, H, W): configs[2]'s four levels at 8 x 384 x 1280 (pitch = the dense-block buffer's), ragged tiles, Co = 1, Ci % 16 != 0
CASES = [(8, 529, 544, 2, 6, 20), (8, 661, 672, 2, 48, 160), (2, 565, 576, 2, 7, 9), (1, 32, 32, 1, 5, 3), (3, 100, 112, 2, 1, 1),
         (1, 597, 600 + 8, 2, 12, 40)]


def _operands(B, ci, pitch, co, H, W, seed, wdtype):
    g = torch.Generator(device="cpu").manual_seed(seed)
    buf = torch.randn(B, H, W, pitch, generator=g).to(DEV).to(torch.bfloat16)
    x = buf.permute(0, 3, 1, 2)[:, :ci]
    w = (torch.randn(ci, co, 4, 4, generator=g) * ci ** -0.5).to(DEV).to(wdtype)
    b = torch.randn(co, generator=g).to(DEV)
    return buf, x, w, b


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("wdtype", [torch.float32, torch.bfloat16])
def test_forward_matches_fp32_transposed_convolution(case, wdtype):
    from ocflow_amd import conv as nconv
    B, ci, pitch, co, H, W = case
    buf, x, w, b = _operands(*case, seed=ci + H, wdtype=wdtype)
    assert nconv.deconv4x4_eligible(x, w, (2, 2), (1, 1), (1, 1), 1, (0, 0))
    keep = buf.clone()
    y = nconv.deconv4x4s2_bias(x, w, b)
    torch.cuda.synchronize()
    assert torch.equal(buf, keep)  # reads only
    assert y.shape == (B, co, 2 * H, 2 * W) and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    want = F.conv_transpose2d(x.float(), w.to(torch.bfloat16).float(), b, 2, 1)
    # one bf16 rounding of an fp32 sum of <= 4 Ci products (accumulation order differs from the library's: 2^-8 relative + fp32 noise)
    err = (y.float() - want).abs()
    assert bool((err <= want.abs() * 2.0 ** -8 + 2e-5 * ci ** 0.5).all()), float(err.max())


def test_garbage_behind_the_slice_is_multiplied_by_zero():
    from ocflow_amd import conv as nconv
    B, ci, pitch, co, H, W = 2, 565, 576, 2, 7, 9
    buf, x, w, b = _operands(B, ci, pitch, co, H, W, seed=1, wdtype=torch.float32)
    y0 = nconv.deconv4x4s2_bias(x, w, b).clone()
    buf[..., ci:] = 3.0e4  # the alignment channels (finite: 0 * inf would be NaN -- the buffer's own are activations or zeros)
    y1 = nconv.deconv4x4s2_bias(x, w, b)
    assert torch.equal(y0, y1)


def test_gradients_equal_the_library_layer():
    from ocflow_amd import conv as nconv
    B, ci, pitch, co, H, W = 2, 565, 576, 2, 7, 9
    buf, x, w, b = _operands(B, ci, pitch, co, H, W, seed=2, wdtype=torch.float32)
    gy = torch.randn(B, co, 2 * H, 2 * W, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    grads = []
    for ours in (True, False):
        leaf = buf.detach().clone().requires_grad_(True)  # (the layer reads a channel range of a wider buffer, as in the decoder)
        full = leaf.permute(0, 3, 1, 2)
        xx = full if ours else full[:, :ci]  # (ours reads the channel range of the whole buffer and returns the buffer's gradient)
        ww, bb = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
        if ours:
            y = nconv.deconv4x4s2_bias(xx, ww, bb)
        else:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = F.conv_transpose2d(xx, ww, bb, 2, 1)
        y.backward(gy)
        grads.append((leaf.grad.float(), ww.grad.float(), bb.grad.float()))
    for a, r in zip(*grads):
        assert a.shape == r.shape
        assert torch.allclose(a, r, rtol=2e-2, atol=2e-2 * float(r.abs().max())), float((a - r).abs().max())


def test_ineligible_shapes_fall_to_the_library_and_refusals_are_loud():
    from ocflow_amd import conv as nconv
    _, x, w, b = _operands(1, 64, 64, 2, 4, 4, seed=3, wdtype=torch.float32)
    assert not nconv.deconv4x4_eligible(x.float(), w, (2, 2), (1, 1), (1, 1), 1, (0, 0))  # fp32 activations
    assert not nconv.deconv4x4_eligible(x, w, (2, 2), (1, 1), (1, 1), 1, (1, 1))  # output padding
    assert not nconv.deconv4x4_eligible(x, w[:, :, :3, :3], (2, 2), (1, 1), (1, 1), 1, (0, 0))
    _, x2, w2, _ = _operands(1, 60, 60, 2, 4, 4, seed=4, wdtype=torch.float32)  # pitch 60 < 64: the fragment loads would leave the pixel
    assert not nconv.deconv4x4_eligible(x2, w2, (2, 2), (1, 1), (1, 1), 1, (0, 0))
    with pytest.raises(RuntimeError):
        nconv.deconv4x4s2_bias(x2, w2, None)
