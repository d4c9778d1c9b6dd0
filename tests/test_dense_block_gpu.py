"""The PWC-style dense blocks (cost_volume_flow_net.py:58-75,176-180; cost_volume_flow_occ_net.py:188-213) on ONE buffer
(ocflow_amd.conv.conv3x3_dense_block: concatenation into a channel slice, slice-in / slice-out convolutions, input gradients accumulated
by the residual epilogue of ocf_conv3x3_igemm_ex, LeakyReLU of the cost volume inside ocf_corr_fwd) against
 (1) the reference's op chain  x = cat(lrelu(conv(x)), x)  in fp32 autograd on the same bf16-rounded operands, and
 (2) the layer-by-layer path of this repo (cat_aligned + conv3x3_bias_act), value for value."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cl(t):
    return t.to(DEV).contiguous(memory_format=torch.channels_last)


def _l2(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    return float((got - want).norm() / (want.norm() + 1e-30))


def _convs(cin, widths, seed):
    torch.manual_seed(seed)
    convs = []
    for w in widths:
        c = nn.Conv2d(cin, w, 3, 1, 1)
        with torch.no_grad():
            c.weight.copy_((torch.randn_like(c.weight) * (2.0 / (9 * cin)) ** 0.5).bfloat16().float())
            c.bias.copy_(torch.randn_like(c.bias) * 0.1)
        convs.append(c)
        cin += w
    return convs


@pytest.mark.parametrize("B,H,W,part_ch,widths,lead", [
    (2, 12, 20, (81, 32, 2, 2), (128, 128, 96, 64, 32), 81),     # a PWC level: cost volume | c1 | up_flow | up_feat
    (2, 6, 8, (81,), (128, 128, 96, 64, 32), 81),                 # level 6: the cost volume alone (small-map kernel)
    (1, 33, 70, (81, 64, 2, 1, 2), (64, 32, 16), 0),              # the pwoc form (5 parts), odd map, no fused activation
    (8, 24, 32, (81, 96, 2, 2), (128, 128, 96, 64, 32), 81),      # 6144 pixels: the staged kernel
])
def test_dense_block_matches_the_reference_chain(B, H, W, part_ch, widths, lead):
    from ocflow_amd import conv as nconv
    g = torch.Generator().manual_seed(B + H + W)
    parts = [torch.randn(B, c, H, W, generator=g).bfloat16() for c in part_ch]
    if lead:  # a LeakyReLU output: its negative values are 0.1 x something
        parts[0] = F.leaky_relu(parts[0].float(), 0.1).bfloat16()
    cin = sum(part_ch)
    convs = _convs(cin, widths, seed=H)
    dconvs = [nn.Conv2d(c.in_channels, c.out_channels, 3, 1, 1).to(DEV) for c in convs]
    for d, c in zip(dconvs, convs):
        d.load_state_dict(c.state_dict())
        d.weight.data = d.weight.data.bfloat16().contiguous(memory_format=torch.channels_last)
    dparts = [_cl(p).requires_grad_(True) for p in parts]
    assert nconv.dense_block_eligible(dparts, dconvs)
    out = nconv.conv3x3_dense_block(dparts, dconvs, 0.1, lead_mask_channels=lead)
    tot, cin_al = sum(widths), (cin + 31) // 32 * 32
    assert out.shape == (B, tot + cin_al, H, W)
    # reference: fp32 autograd of the reference's chain on the same operands, every layer's output rounded to bf16 (what the buffer holds)
    rparts = [p.float().requires_grad_(True) for p in parts]
    pre = None
    if lead:  # the gradient the node returns for part 0 is w.r.t. the UN-activated volume: model it as lrelu(pre), pre = its preimage
        pre = torch.where(parts[0].float() > 0, parts[0].float(), parts[0].float() / 0.1).requires_grad_(True)
        rparts[0] = F.leaky_relu(pre, 0.1)
    rws = [c.weight.detach().clone().requires_grad_(True) for c in convs]
    rbs = [c.bias.detach().clone().requires_grad_(True) for c in convs]
    x = torch.cat(rparts, 1)
    for w_, b_ in zip(rws, rbs):
        y = F.leaky_relu(F.conv2d(x, w_, b_, 1, 1), 0.1)
        y = y + (y.detach().bfloat16().float() - y.detach())  # bf16 storage, straight-through
        x = torch.cat((y, x), 1)
    assert _l2(out[:, :tot + cin], x) < 6e-3, _l2(out[:, :tot + cin], x)
    assert float(out[:, tot + cin:].abs().max()) == 0.0 if cin_al > cin else True
    gy = torch.randn(x.shape, generator=g).bfloat16()
    gfull = torch.zeros(out.shape, dtype=torch.bfloat16)
    gfull[:, :tot + cin] = gy
    out.backward(_cl(gfull))
    x.backward(gy.float())
    # bars: a five-layer LeakyReLU stack is ill-conditioned in its masks (tests/test_conv_igemm_gpu.py::test_chain_...): relative L2
    for i, (dp, rp) in enumerate(zip(dparts, [pre if (lead and k == 0) else r for k, r in enumerate(rparts)])):
        assert _l2(dp.grad, rp.grad) < 6e-2, (i, _l2(dp.grad, rp.grad))
    for j, (d, w_, b_) in enumerate(zip(dconvs, rws, rbs)):
        assert _l2(d.weight.grad, w_.grad) < 6e-2, (j, _l2(d.weight.grad, w_.grad))
        assert _l2(d.bias.grad, b_.grad) < 6e-2, (j, _l2(d.bias.grad, b_.grad))


def test_dense_block_equals_the_layer_by_layer_path():
    """same values as cat_aligned + conv3x3_bias_act layer by layer (this repo's round-4 path): forward bit for bit on the real
    channels, gradients within the bf16 roundings of the accumulation order"""
    from ocflow_amd import conv as nconv
    from ocflow_amd.models.networks._epilogue import cat_aligned
    B, H, W, part_ch, widths = 2, 20, 36, (81, 64, 2, 2), (128, 128, 96, 64, 32)
    g = torch.Generator().manual_seed(3)
    parts = [torch.randn(B, c, H, W, generator=g).bfloat16() for c in part_ch]
    cin = sum(part_ch)
    convs = [c.to(DEV) for c in _convs(cin, widths, seed=1)]
    for c in convs:
        c.weight.data = c.weight.data.bfloat16().contiguous(memory_format=torch.channels_last)
    runs = {}
    for mode in ("dense", "layers"):
        dp = [_cl(p).requires_grad_(True) for p in parts]
        for c in convs:
            c.weight.grad = c.bias.grad = None
        if mode == "dense":
            out = nconv.conv3x3_dense_block(dp, convs, 0.1, 0)
            tot = sum(widths)
            real = out[:, :tot + cin]
        else:
            x = cat_aligned(dp)
            for c in convs:
                x = torch.cat((nconv.conv3x3_bias_act(x, c.weight, c.bias, 0.1, 1), x), 1)
            real = x[:, :sum(widths) + cin]
        gy = torch.randn(real.shape, generator=torch.Generator().manual_seed(9)).bfloat16()
        real.backward(_cl(gy))
        runs[mode] = (real.detach().clone(), [p.grad.clone() for p in dp], [c.weight.grad.clone() for c in convs], [c.bias.grad.clone() for c in convs])
    assert torch.equal(runs["dense"][0], runs["layers"][0])
    for k in (1, 2, 3):
        for a, b in zip(runs["dense"][k], runs["layers"][k]):
            assert _l2(a, b) < 2e-2, (k, _l2(a, b))


def test_cost_volume_leaky_slope_in_the_kernel():
    """ocf_corr_fwd(leaky_slope) against leaky_relu(ocf_corr_fwd): positive taps bit for bit, negative ones within one bf16 rounding
    (the fused epilogue scales the fp32 mean and rounds ONCE, the two-op chain rounds the mean and then the product) -- bf16: every
    level's kernel; fp32 -- and the standalone operator's backward applies the activation's derivative"""
    from ocflow_amd import ops
    for (B, C_, H, W), dt in [((8, 32, 24, 64), torch.bfloat16), ((2, 64, 14, 40), torch.bfloat16), ((2, 96, 12, 20), torch.bfloat16),
                              ((2, 196, 6, 8), torch.bfloat16), ((1, 32, 9, 17), torch.float32)]:
        g = torch.Generator().manual_seed(C_)
        f1 = torch.randn(B, C_, H, W, generator=g).to(dt)
        f2 = torch.randn(B, C_, H, W, generator=g).to(dt)
        a, b = (_cl(f1), _cl(f2)) if dt == torch.bfloat16 else (f1.to(DEV), f2.to(DEV))
        plain = ops.compute_cost_volume(a, b, 4)
        fused = ops.compute_cost_volume(a, b, 4, leaky_slope=0.1)
        want = F.leaky_relu(plain.float(), 0.1).to(dt)
        if dt == torch.bfloat16:
            pos = plain > 0
            assert torch.equal(fused[pos], want[pos])
            assert torch.allclose(fused.float(), want.float(), rtol=2.0 ** -7, atol=1e-30), float((fused.float() - want.float()).abs().max())
            assert float((fused[~pos].float() > 0).sum()) == 0
        else:
            assert float((fused - want).abs().max()) < 1e-6
    a1, b1 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    a2, b2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    gy = torch.randn(plain.shape, device=DEV)
    ops.compute_cost_volume(a1, b1, 4, leaky_slope=0.1).backward(gy)
    F.leaky_relu(ops.compute_cost_volume(a2, b2, 4), 0.1).backward(gy)
    assert float((a1.grad - a2.grad).abs().max()) < 1e-5 and float((b1.grad - b2.grad).abs().max()) < 1e-5


@pytest.mark.parametrize("model", ["pwc", "pwoc"])
def test_pwc_nets_train_on_the_dense_block_and_match_the_unfused_run(model, monkeypatch):
    """whole net, bf16 channels_last: the dense-block path against OCFLOW_NATIVE_DENSE=0 (the round-4 path: cat_aligned + per-layer
    kernels) on the same weights -- outputs and a late / an early gradient"""
    from ocflow_amd import ops
    res = {}
    for dense in ("1", "0"):
        monkeypatch.setenv("OCFLOW_NATIVE_DENSE", dense)
        torch.manual_seed(11)
        if model == "pwc":
            from ocflow_amd.models.networks.cost_volume_flow_net import FlowNetCV
            net = FlowNetCV()
        else:
            from ocflow_amd.models.networks.cost_volume_flow_occ_net import FlowOccNetCV
            net = FlowOccNetCV()
        net = net.to(DEV).to(memory_format=torch.channels_last).train()
        x = _cl(torch.rand(2, 6, 128, 192, generator=torch.Generator().manual_seed(5)) * 2 - 1)
        calls = []

        class Spy:
            def launch(self, name, meta, fn):
                calls.append(name)
                return fn()

        ops.set_kernel_timer(Spy())
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = net(x)
            loss = sum(o.float().abs().mean() for o in out)
            loss.backward()
        finally:
            ops.set_kernel_timer(None)
        if dense == "1":
            assert "concat" in calls, sorted(set(calls))
        res[dense] = ([o.detach().float().clone() for o in out], net.conv2_0[0].weight.grad.float().clone(), net.conv6_4[0].weight.grad.float().clone(),
                      net.conv1a[0].weight.grad.float().clone())
    for a, b in zip(res["1"][0], res["0"][0]):
        assert _l2(a, b) < 2e-2, _l2(a, b)
    for k in (1, 2, 3):
        assert _l2(res["1"][k], res["0"][k]) < 8e-2, (k, _l2(res["1"][k], res["0"][k]))
