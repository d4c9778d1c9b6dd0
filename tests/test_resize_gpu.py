"""ocf_resize_bilinear against torch.nn.functional.interpolate (reference call site: models/model.py:63-69, the quarter-resolution image of
the PWC unsupervised step): same fp32 formula, so the results agree to the last bits (tolerance 2 ulp of the image range: the compilers
may contract different multiply-adds)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("shape,size", [((8, 3, 384, 512), (96, 128)), ((2, 3, 37, 53), (9, 13)), ((1, 1, 5, 7), (1, 1)), ((2, 2, 8, 8), (16, 24))])
@pytest.mark.parametrize("align", [True, False])
def test_matches_interpolate_on_a_channel_slice(shape, size, align):
    from ocflow_amd import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(H * W)
    pair = torch.rand((B, 2 * C, H, W), generator=g).to(DEV)
    x = pair[:, :C]  # the reference's `inputs[:, 0:3]`
    got = ops.resize_bilinear(x, size, align_corners=align)
    want = F.interpolate(x, size=size, mode="bilinear", align_corners=align)
    assert got.shape == want.shape and got.dtype == torch.float32
    assert float((got - want).abs().max()) <= 2.4e-7
    want_cpu = F.interpolate(x.cpu(), size=size, mode="bilinear", align_corners=align)
    assert float((got.cpu() - want_cpu).abs().max()) <= 2.4e-7


def test_scale_factor_quarter_is_the_size_the_model_asks_for():
    from ocflow_amd import ops
    x = torch.rand((2, 3, 384, 512)).to(DEV)
    want = F.interpolate(x, scale_factor=0.25, mode="bilinear", align_corners=True)
    got = ops.resize_bilinear(x, (int(384 * 0.25), int(512 * 0.25)), align_corners=True)
    assert got.shape == want.shape and float((got - want).abs().max()) <= 2.4e-7


def test_refuses_what_needs_a_gradient():
    from ocflow_amd import ops
    with pytest.raises(ValueError):
        ops.resize_bilinear(torch.rand((1, 3, 8, 8), device=DEV, requires_grad=True), (2, 2))
    with pytest.raises(ValueError):
        ops.resize_bilinear(torch.rand((1, 3, 8, 8), device=DEV).to(torch.bfloat16), (2, 2))
