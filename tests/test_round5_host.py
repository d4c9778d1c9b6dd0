"""Host-side logic of the round-5 paths on CPU tensors: the new routing code must hand everything that is not a bf16 HIP tensor to the
reference's own operators, with the reference's values.  The product has no CPU path: as in tests/test_host_models.py the elementwise hot
ops are computed by the oracle (monkeypatched in)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle_backend import patch_ops


@pytest.fixture
def cpu_ops(monkeypatch):
    patch_ops(monkeypatch)


def test_pwc_encoder_routing_equals_the_plain_layer_sequence_on_cpu(cpu_ops):
    """cost_volume_flow_net.encode (padded level 6 in the bf16 HIP configuration) is the reference's layer sequence everywhere else
    (cost_volume_flow_net.py:158-169)"""
    from ocflow_amd.models.networks.cost_volume_flow_net import FlowNetCV, encode
    torch.manual_seed(0)
    net = FlowNetCV().eval()
    im = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        got = encode(net, im)
        x, want = im, []
        for lvl in range(1, 7):
            for suffix in (("a", "aa", "b") if lvl < 6 else ("aa", "a", "b")):
                layer = getattr(net, f"conv{lvl}{suffix}")
                x = F.leaky_relu(layer[0](x), layer[1].negative_slope)  # (nn.Sequential(nn.Conv2d, nn.LeakyReLU(0.1)) of the reference)
            want.append(x)
    assert [tuple(t.shape) for t in got] == [tuple(t.shape) for t in want]
    assert [int(t.shape[1]) for t in got] == [16, 32, 64, 96, 128, 196]
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_padding_predicates_are_false_off_the_device(cpu_ops):
    from ocflow_amd import conv as nconv
    from ocflow_amd.models.networks._epilogue import conv_lrelu, conv_lrelu_padded, padded_channels_ok
    x = torch.randn(1, 128, 6, 8)
    assert not padded_channels_ok(x)
    conv = nn.Conv2d(128, 196, 3, 2, 1)
    with torch.no_grad():
        assert torch.equal(conv_lrelu_padded(conv, x, 0.1), conv_lrelu(conv, x, 0.1))  # (196 channels, no padding)
    w = torch.randn(600, 2, 4, 4)
    assert not nconv.deconv4x4_eligible(torch.randn(1, 600, 4, 4), w, (2, 2), (1, 1), (1, 1), 1, (0, 0))


def test_deconv_bias_range_reads_the_leading_channels_on_cpu(cpu_ops):
    from ocflow_amd.models.networks._epilogue import deconv_bias_range
    torch.manual_seed(1)
    deconv = nn.ConvTranspose2d(37, 2, 4, 2, 1)
    x = torch.randn(2, 48, 5, 7)
    with torch.no_grad():
        got = deconv_bias_range(deconv, x)
        want = F.conv_transpose2d(x[:, :37], deconv.weight, deconv.bias, 2, 1)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6)


def test_unpad_channels_and_its_gradient_on_cpu():
    from ocflow_amd import ops
    x = torch.randn(2, 8, 3, 4, requires_grad=True)
    y = ops.unpad_channels(x, 5)
    assert y.shape == (2, 5, 3, 4) and torch.equal(y, x[:, :5])
    g = torch.randn_like(y)
    y.backward(g)
    assert torch.equal(x.grad[:, :5], g) and float(x.grad[:, 5:].abs().max()) == 0.0
    assert ops.unpad_channels(x, 8) is x


def test_quarter_resolution_image_is_the_frameworks_on_cpu(cpu_ops):
    """FlowStageModel._predict: ocf_resize_bilinear serves fp32 HIP tensors only; the CPU path is F.interpolate (models/model.py:63-69)"""
    from ocflow_amd.models.model import FlowStageModel
    m = FlowStageModel({"model": "pwc", "learning_rate": 1e-4})
    imgs = torch.rand(1, 6, 64, 64)
    flow = (torch.zeros(1, 2, 64, 64), torch.zeros(1, 2, 16, 16))
    _, _, img_s = m._predict(imgs, flow)
    assert torch.equal(img_s, F.interpolate(imgs[:, :3], scale_factor=0.25, mode="bilinear", align_corners=True))
