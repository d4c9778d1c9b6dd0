"""GPU parity ON THE TIMED PATH: the exact shapes `bench.py` runs (8 pairs per GPU at 436x1024 => the five FPN pyramid levels
B=8 x (C, h, w) below, and 8 x 3 x 436 x 1024 for the loss kernels), HIP kernels through the C ABI against the CPU oracle on the
same seeded inputs.  Each correlation test also asserts WHICH kernel instantiation ran (ocf_corr_fwd_variant), under the automatic
policy, without the wave kernel and with the band-GEMM kernels forced (per-call policies of ABI version 2), so that every variant the
benchmark can time is value-checked at its own shape: row kernel K1 / K2, wave kernel (3 / 4 / 7 channel blocks, one and two strips),
direct kernel (3 / 4 / 7 channel blocks), band TH=8 with the persistent multi-tile walk (896 tiles > 512 workgroups), band MULTI
TH=4 / TH=2 / TH=1.

bf16 bar: inputs are bf16-rounded once, the oracle runs in fp32 on those values; the HIP result differs from it by the output's own
bf16 rounding (<= 2^-8 relative) plus fp32 summation order => rtol 8e-3 (two bf16 ulps), atol 1e-3 of the value scale."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ocflow_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B = 8
# (C, h, w) of FeaturePyramidNet at 436x1024, level 6 .. 2 and the warp displacement scale of the level (20 / 2^level)
LEVELS = {6: (196, 7, 16), 5: (128, 14, 32), 4: (96, 28, 64), 3: (64, 55, 128), 2: (32, 109, 256)}


@pytest.fixture(scope="module")
def ops():
    from ocflow_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def lib():
    from ocflow_amd import _lib
    return _lib


def _t4(lib, t):
    return C.byref(lib.Tensor4(t.data_ptr(), *t.stride()))


def cl(t):
    return t.to(DEV).contiguous(memory_format=torch.channels_last)


def assert_close(got, want, rtol, atol_rel, what="", outliers=0.0):
    """|got - want| <= atol_rel * max|want| + rtol * |want| everywhere, except for at most `outliers` (a fraction) of the elements:
    the flow gradient of a bilinear warp is DISCONTINUOUS where a sample coordinate crosses an integer, and the oracle (normalised
    grid -> grid_sample) and the kernel (closed form) round that coordinate differently in the last bit."""
    got = got.detach().float().cpu().numpy()
    want = want.detach().float().cpu().numpy()
    scale = float(np.abs(want).max()) + 1e-30
    if outliers > 0.0:
        bad = np.abs(got - want) > atol_rel * scale + rtol * np.abs(want)
        assert bad.mean() <= outliers, f"{what}: {int(bad.sum())} of {bad.size} elements out of tolerance (allowed fraction {outliers})"
        return
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol_rel * scale, err_msg=what)


def level_inputs(level, seed=0):
    Cc, h, w = LEVELS[level]
    gen = torch.Generator().manual_seed(1000 * level + seed)
    f1 = torch.randn((B, Cc, h, w), generator=gen).bfloat16()
    f2 = torch.randn((B, Cc, h, w), generator=gen).bfloat16()
    # a smooth-ish flow of a few pixels at this level plus out-of-range spikes at the borders
    flow = torch.randn((B, 2, h, w), generator=gen) * 1.5
    occ = torch.rand((B, 1, h, w), generator=gen)
    return f1, f2, flow, occ


# variant the automatic policy / the automatic policy without the wave kernel / the forced band policy must pick at B = 8 on a 256-CU MI355X
EXPECTED = {
    2: ("OCF_CORR_ROWS_K1", "OCF_CORR_ROWS_K1", "OCF_CORR_BAND_TH8"),
    3: ("OCF_CORR_ROWS_K2", "OCF_CORR_ROWS_K2", "OCF_CORR_BAND_MULTI_TH4"),
    4: ("OCF_CORR_WAVE", "OCF_CORR_DIRECT", "OCF_CORR_BAND_MULTI_TH2"),
    5: ("OCF_CORR_WAVE", "OCF_CORR_DIRECT", "OCF_CORR_BAND_MULTI_TH1"),
    6: ("OCF_CORR_WAVE", "OCF_CORR_DIRECT", "OCF_CORR_BAND_MULTI_TH1"),
}
VARIANT_CODE = {"OCF_CORR_ROWS_K1": 1, "OCF_CORR_ROWS_K2": 2, "OCF_CORR_DIRECT": 3, "OCF_CORR_WAVE": 4, "OCF_CORR_BAND_TH8": 11,
                "OCF_CORR_BAND_MULTI_TH4": 12, "OCF_CORR_BAND_MULTI_TH2": 13, "OCF_CORR_BAND_MULTI_TH1": 14}
POLICY_CODE = {"auto": 0, "no_wave": 2, "band": 1, "wave": 3}


@pytest.mark.parametrize("level", [2, 3, 4, 5, 6])
@pytest.mark.parametrize("policy", ["auto", "no_wave", "band"])
@pytest.mark.parametrize("with_occ", [True, False])
def test_corr_forward_at_bench_shapes(ops, lib, level, policy, with_occ):
    """flow_occ_net.py:155-168 at the benchmark's shapes: warp(f2, flow * 20/2^level) -> * occ -> cost volume"""
    f1, f2, flow, occ = level_inputs(level)
    Cc, h, w = LEVELS[level]
    fs = 20.0 / 2 ** level
    L = lib.load()
    a, b2, fl, oc = cl(f1), cl(f2), flow.to(DEV), occ.to(DEV) if with_occ else None
    with ops.kernel_policy(corr_fwd=POLICY_CODE[policy]):
        w2 = ops.warp(b2, fl, False, flow_scale=fs) if level != 6 else b2  # level 6 is not warped (flow_occ_net.py:151)
        out = ops.corr_warp(a, w2, None, oc, 4)
    got_variant = L.ocf_corr_fwd_variant(_t4(lib, a), _t4(lib, w2), None, _t4(lib, oc) if with_occ else None, _t4(lib, out), B, Cc, h, w,
                                         4, lib.OCF_BF16, POLICY_CODE[policy])
    assert got_variant == VARIANT_CODE[EXPECTED[level][("auto", "no_wave", "band").index(policy)]], got_variant
    assert out.dtype == torch.bfloat16 and out.shape == (B, 81, h, w) and out.stride(1) == 1
    if level != 6:
        assert_close(w2, O.warp(f2.float(), flow * fs, False), 8e-3, 4e-3, "warp forward")
    # the correlation against the oracle on exactly the operand the kernel consumed (the bf16 warped tensor)
    w2c = w2.float().cpu()
    ref = O.compute_cost_volume(f1.float(), w2c * occ if with_occ else w2c, 4)
    assert_close(out, ref, 8e-3, 1e-3, f"corr level {level} {policy}")


@pytest.mark.parametrize("level", [2, 3, 4, 5, 6])
def test_row_kernel_equals_band_kernel_bitwise(ops, lib, level):
    """same products, same fp32 accumulation order inside the MFMA, same scaling order: the two kernels agree bit for bit"""
    f1, f2, _, occ = level_inputs(level, seed=3)
    a, b2, oc = cl(f1), cl(f2), occ.to(DEV)
    with ops.kernel_policy(corr_fwd=POLICY_CODE["band"]):
        band = ops.corr_warp(a, b2, None, oc, 4)
        band_plain = ops.corr_warp(a, b2, None, None, 4)
    for policy in ("auto", "no_wave", "wave"):  # row / wave / direct kernels: all bit-identical to the band kernel
        with ops.kernel_policy(corr_fwd=POLICY_CODE[policy]):
            assert torch.equal(ops.corr_warp(a, b2, None, oc, 4), band), policy
            assert torch.equal(ops.corr_warp(a, b2, None, None, 4), band_plain), policy


@pytest.mark.parametrize("level", [2, 3, 4, 5, 6])
def test_cost_volume_written_into_the_decoder_input_buffer(ops, lib, level):
    """`out=`: the cost volume lands in the first 81 channels of a 128-channel channels_last buffer (the decoder input of
    flow_occ_net.py:171-176: [cost volume | features | flow | flow features | alignment]) -- the wave kernel takes any pixel pitch.
    Same bits as the dense result, the rest of the buffer untouched."""
    f1, f2, _, occ = level_inputs(level, seed=5)
    Cc, h, w = LEVELS[level]
    a, b2, oc = cl(f1), cl(f2), occ.to(DEV)
    dense = ops.corr_warp(a, b2, None, oc, 4)
    buf = torch.full((B, 128, h, w), 7.0, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    v = lib.load().ocf_corr_fwd_variant(_t4(lib, a), _t4(lib, b2), None, _t4(lib, oc), _t4(lib, buf[:, :81]), B, Cc, h, w, 4, lib.OCF_BF16, 0)
    assert v == VARIANT_CODE["OCF_CORR_WAVE"], v
    res = ops.corr_warp(a, b2, None, oc, 4, out=buf[:, :81])
    assert res.data_ptr() == buf.data_ptr() and torch.equal(res, dense)
    assert float((buf[:, 81:].float() - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("level", [2, 3, 4, 5])
@pytest.mark.parametrize("with_occ", [True, False])
def test_fused_warp_correlation_equals_two_launches(ops, lib, level, with_occ, monkeypatch):
    """K1+K2 as ONE launch (flow_net.py:110-112 / the no-grad second forward of models/model.py:381-388): the row-tiled kernels (levels
    2, 3) sample warp(f2, flow * 20/2^level) while they stage the second operand, the wave kernel (levels 4, 5) in registers --
    bit-identical to ocf_warp_fwd followed by the correlation kernel; taken under OCFLOW_FUSED_WARP=gather when no gradient is recorded"""
    f1, f2, flow, occ = level_inputs(level, seed=6)
    Cc, h, w = LEVELS[level]
    fs = 20.0 / 2 ** level
    flow[:, :, :2] *= 30.0  # sources far outside the image
    a, b2, fl, oc = cl(f1), cl(f2), flow.to(DEV), occ.to(DEV) if with_occ else None
    monkeypatch.setattr(ops, "_FUSED_WARP", "0")
    two = ops.corr_warp(a, b2, fl, oc, 4, fs, False)
    monkeypatch.setattr(ops, "_FUSED_WARP", "gather")
    out = torch.empty_like(two)
    v = lib.load().ocf_corr_fwd_variant(_t4(lib, a), _t4(lib, b2), _t4(lib, fl), _t4(lib, oc) if with_occ else None, _t4(lib, out), B, Cc, h, w,
                                        4, lib.OCF_BF16, 0)
    assert v == VARIANT_CODE[EXPECTED[level][0]], v
    timer_calls = []

    class Spy:
        def launch(self, name, meta, fn):
            timer_calls.append(name)
            return fn()

    ops.set_kernel_timer(Spy())
    try:
        with torch.no_grad():
            one = ops.corr_warp(a, b2, fl, oc, 4, fs, False)
        assert timer_calls == ["corr_warp_fwd"], timer_calls  # a single launch, no warp kernel
        timer_calls.clear()
        x = a.clone().requires_grad_()
        ops.corr_warp(x, b2, fl, oc, 4, fs, False)
        assert timer_calls == ["warp_fwd", "corr_fwd"], timer_calls  # a recorded forward keeps the warped tensor for backward
    finally:
        ops.set_kernel_timer(None)
    assert torch.equal(one, two)
    w2 = O.warp(f2.float(), flow * fs, False).bfloat16().float()
    assert_close(one, O.compute_cost_volume(f1.float(), w2 * occ if with_occ else w2, 4), 8e-3, 2e-3, f"fused level {level}")


@pytest.mark.parametrize("shape", [(1, 32, 5, 40), (2, 24, 9, 33), (1, 64, 7, 70), (3, 40, 13, 31), (1, 8, 1, 8), (2, 32, 4, 64), (1, 56, 3, 100),
                                   (1, 96, 5, 40), (2, 128, 3, 24), (1, 196, 6, 8), (1, 100, 9, 16), (2, 72, 4, 33), (1, 224, 2, 16)])
@pytest.mark.parametrize("with_occ", [True, False])
def test_row_kernel_edge_tiles(ops, lib, shape, with_occ):
    """partial tiles in x and y, widths that are not a multiple of 8 (scalar row tail), channel counts below a full 32-channel
    block, one-row / one-tile maps -- the row kernel (asserted) against the oracle"""
    Bb, Cc, h, w = shape
    gen = torch.Generator().manual_seed(sum(shape))
    f1 = torch.randn(shape, generator=gen).bfloat16()
    f2 = torch.randn(shape, generator=gen).bfloat16()
    occ = torch.rand((Bb, 1, h, w), generator=gen) if with_occ else None
    a, b2 = cl(f1), cl(f2)
    oc = occ.to(DEV) if with_occ else None
    out = ops.corr_warp(a, b2, None, oc, 4)
    if (w * 81 * 2) % 16 == 0:  # rows of the dense output start on 16-byte boundaries: the row kernel's precondition
        v = lib.load().ocf_corr_fwd_variant(_t4(lib, a), _t4(lib, b2), None, _t4(lib, oc) if with_occ else None, _t4(lib, out), Bb, Cc, h, w, 4,
                                            lib.OCF_BF16, 0)
        assert v in (1, 2, 3, 4), v
        if v == 4:  # the direct kernel the wave kernel replaced is still value-checked at these edge shapes
            with ops.kernel_policy(corr_fwd=POLICY_CODE["no_wave"]):
                assert torch.equal(ops.corr_warp(a, b2, None, oc, 4), out)
    with ops.kernel_policy(corr_fwd=POLICY_CODE["wave"]):  # and the wave kernel at every edge shape (C <= 64 included, odd widths: any pitch)
        assert torch.equal(ops.corr_warp(a, b2, None, oc, 4), out)
    ref = O.compute_cost_volume(f1.float(), f2.float() * occ if with_occ else f2.float(), 4)
    assert_close(out, ref, 8e-3, 1e-3)


@pytest.mark.parametrize("with_occ", [True, False])
@pytest.mark.parametrize("level", [2, 3, 4, 5, 6])
def test_corr_occ_backward_at_bench_shapes(ops, lib, level, with_occ):
    """The correlation backward AS THE BENCHMARK RUNS IT: the cotangent is channels [0:81] of an 8 x 128 x h x w channels_last tensor (the
    gradient of the decoder-input concatenation: 256-byte pixel pitch), with the occlusion mask folded in (FlowOccNet) and without
    (FlowNet / the unsupervised step); the kernel identity is asserted -- ocf_corr_bwd_variant must answer the matrix-core kernel
    (corr_bwd_mfma.hip: `corr_bwd_mfma_par_lds`, both gradients in one launch, the gradient operand staged through LDS) at all five
    levels -- and its values are compared with fp32 autograd of the oracle; gradients leave the kernel in bf16."""
    f1, f2, _, occ = level_inputs(level, seed=1)
    Cc, h, w = LEVELS[level]
    gen = torch.Generator().manual_seed(77 + level)
    wide = torch.randn((B, 128, h, w), generator=gen).bfloat16()
    cot = wide[:, :81]
    a, b2, o = f1.float().requires_grad_(), f2.float().requires_grad_(), occ.clone().requires_grad_()
    if with_occ:
        g1r, g2r, gor = torch.autograd.grad(O.compute_cost_volume(a, b2 * o, 4), (a, b2, o), cot.float())
    else:
        g1r, g2r = torch.autograd.grad(O.compute_cost_volume(a, b2, 4), (a, b2), cot.float())
    x, y, oc = cl(f1).requires_grad_(), cl(f2).requires_grad_(), occ.to(DEV).requires_grad_()
    wide_d = cl(wide)
    cot_d = wide_d[:, :81]
    assert cot_d.stride(3) == 128 and cot_d.stride(1) == 1  # the in-model layout: a channel slice at a 256-byte pitch
    g1b, g2b = torch.empty_like(x), torch.empty_like(y)
    v = lib.load().ocf_corr_bwd_variant(_t4(lib, x.detach()), _t4(lib, y.detach()), _t4(lib, cot_d), _t4(lib, g1b), _t4(lib, g2b), B, Cc, h, w, 4,
                                        lib.OCF_BF16, 0)
    assert v == 1, f"level {level}: ocf_corr_bwd_variant = {v}, expected OCF_CORR_BWD_MFMA (1)"
    if with_occ:
        g1, g2, go = torch.autograd.grad(ops.corr_warp(x, y, None, oc, 4), (x, y, oc), cot_d)
        assert_close(go, gor, 2e-2, 4e-3, "grad occ")
    else:
        g1, g2 = torch.autograd.grad(ops.corr_warp(x, y, None, None, 4), (x, y), cot_d)
    assert_close(g1, g1r, 2e-2, 4e-3, "grad f1")
    assert_close(g2, g2r, 2e-2, 4e-3, "grad f2")


@pytest.mark.parametrize("level", [2, 3, 4, 5])
@pytest.mark.parametrize("deterministic", [False, True])
def test_warp_backward_at_bench_shapes(ops, level, deterministic):
    """ocf_warp_fwd / ocf_warp_bwd (AC-False feature warp) at the bench shapes vs autograd of the oracle's grid_sample warp"""
    f1, f2, flow, _ = level_inputs(level, seed=2)
    Cc, h, w = LEVELS[level]
    fs = 20.0 / 2 ** level
    gen = torch.Generator().manual_seed(5 + level)
    cot = torch.randn((B, Cc, h, w), generator=gen).bfloat16()
    img, fl = f2.float().requires_grad_(), flow.clone().requires_grad_()
    gir, gfr = torch.autograd.grad(O.warp(img, fl * fs, False), (img, fl), cot.float())
    x, f = cl(f2).requires_grad_(), flow.to(DEV).requires_grad_()
    ops.set_deterministic(deterministic)
    try:
        out = ops.warp(x, f, False, flow_scale=fs)
        gi, gf = torch.autograd.grad(out, (x, f), cl(cot))
        if deterministic:
            gi2, gf2 = torch.autograd.grad(ops.warp(x, f, False, flow_scale=fs), (x, f), cl(cot))
            assert torch.equal(gi, gi2) and torch.equal(gf, gf2)
    finally:
        ops.set_deterministic(False)
    assert_close(gi, gir, 2e-2, 4e-3, "grad img")
    # the flow gradient is a sum over C channels of products with bf16 features: relative to its own scale
    assert_close(gf, gfr, 2e-2, 4e-3, "grad flow", outliers=1e-4)


def _loss_inputs():
    gen = torch.Generator().manual_seed(11)
    H, W = 436, 1024
    base = torch.rand((B, 3, H // 4 + 2, W // 4 + 2), generator=gen) * 2 - 1
    img1 = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
    img2 = (img1 + 0.1 * torch.randn((B, 3, H, W), generator=gen)).clamp(-1, 1)
    flow = torch.nn.functional.interpolate(torch.randn((B, 2, H // 16, W // 16), generator=gen) * 4, size=(H, W), mode="bilinear",
                                           align_corners=False)
    occ = (torch.rand((B, 1, H, W), generator=gen) < 0.3).float()
    return img1, img2, flow, occ


@pytest.mark.parametrize("pair_layout", [False, True])
def test_loss_kernels_full_size_vs_oracle(ops, pair_layout):
    """photometric (fused AC-True warp + mask), first/second-order smoothness and SSIM at 8 x 3 x 436 x 1024: values and flow
    gradients vs the oracle (multi-block fp64-atomic reductions, 1024-block grids) -- fp32 bar.  pair_layout: the images as the model
    hands them over -- the two halves of ONE six-channel channels_last tensor (channel stride 1, pixel pitch 24 bytes: the packed
    12-byte-load forms of the photometric and smoothness kernels)"""
    img1, img2, flow, occ = _loss_inputs()
    # the photometric reference in float64: at W = 1024 an fp32 sample coordinate x + u only resolves 6e-5 px (and the reference's
    # normalise / un-normalise round trip of the grid loses more), so two fp32 evaluations of the flow gradient -- a piecewise
    # constant image difference times bilinear weights -- legitimately differ by ~1e-4 relative and flip at integer crossings
    fl = flow.double().requires_grad_()
    ref_p = O.photometric_error(O.warp(img2.double(), fl, True), img1.double(), occ.double())
    (gp_ref,) = torch.autograd.grad(ref_p, fl)
    fl2 = flow.clone().requires_grad_()
    s1_ref, s2_ref = O.first_order_smoothness_loss(img1, fl2), O.second_order_smoothness_loss(img1, fl2)
    (gs_ref,) = torch.autograd.grad(0.7 * s1_ref + 0.3 * s2_ref, fl2)

    d1, d2, dfl, docc = img1.to(DEV), img2.to(DEV), flow.to(DEV).requires_grad_(), occ.to(DEV)
    if pair_layout:
        pair = torch.cat((d1, d2), 1).contiguous(memory_format=torch.channels_last)
        d1, d2 = pair[:, 0:3], pair[:, 3:6]
        assert d1.stride(1) == 1 and d1.stride(3) == 6
    p = ops.warp_photometric_error(d1, d2, dfl, docc)
    assert abs(float(p) - float(ref_p)) <= 2e-5 * abs(float(ref_p))
    (gp,) = torch.autograd.grad(p, dfl)
    assert_close(gp, gp_ref, 2e-3, 2e-4, "photometric grad_flow", outliers=1e-3)
    p_nomask = ops.warp_photometric_error(d1, d2, dfl.detach(), None)
    ref_nomask = O.photometric_error(O.warp(img2.double(), flow.double(), True), img1.double(), None)
    assert abs(float(p_nomask) - float(ref_nomask)) <= 2e-5 * abs(float(ref_nomask))
    dfl2 = flow.to(DEV).requires_grad_()
    s1, s2 = ops.smoothness_losses(d1, dfl2)
    assert abs(float(s1) - float(s1_ref)) <= 2e-5 * abs(float(s1_ref))
    assert abs(float(s2) - float(s2_ref)) <= 2e-5 * abs(float(s2_ref))
    (gs,) = torch.autograd.grad(0.7 * s1 + 0.3 * s2, dfl2)
    assert_close(gs, gs_ref, 1e-4, 1e-5, "smoothness grad_flow")
    for ws in (11, 4):
        assert abs(float(ops.ssim(d1, d2, ws)) - float(O.ssim(img1, img2, ws))) <= 2e-5


def test_range_map_full_batch_vs_oracle(ops):
    """the LDS-window splat at the benchmark's 8 x 436 x 1024 (3488 workgroups), incl. flows that leave the window and the image"""
    gen = torch.Generator().manual_seed(21)
    flow = torch.nn.functional.interpolate(torch.randn((B, 2, 28, 64), generator=gen) * 6, size=(436, 1024), mode="bilinear")
    flow[:, :, :3, :] *= 20.0  # far out of the window / out of the image
    from ocflow_amd import _lib
    for det, policy in ((False, _lib.RANGE_SPLAT), (False, _lib.RANGE_GATHER), (True, _lib.RANGE_AUTO)):  # splat kernel, gather + far kernels, fixed-point deterministic kernel
        ops.set_deterministic(det)
        try:
            with ops.kernel_policy(range_map=policy):
                rm = ops.compute_range_map(flow.to(DEV))
                oc = ops.occlusion_from_flow(flow.to(DEV))
                if det:
                    assert torch.equal(rm, ops.compute_range_map(flow.to(DEV)))
        finally:
            ops.set_deterministic(False)
        ref = O.compute_range_map(flow)
        assert_close(rm, ref, 1e-4, 1e-6, "range map")
        assert_close(oc, O.occlusion_from_range(ref), 1e-4, 1e-5, "occlusion")


def test_warp_backward_binned_overflow(ops, lib):
    """a CONVERGENT flow: every source of a 64 x 96 map samples near the centre, so the bins of the few central tiles (896 entries each)
    overflow and the surplus (source, tile) pairs take the overflow list -> float-atomic path of the per-source kernel; against autograd
    of the oracle's grid_sample warp"""
    Bb, Cc, h, w = 2, 32, 64, 96
    gen = torch.Generator().manual_seed(91)
    img = torch.randn((Bb, Cc, h, w), generator=gen)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    flow = torch.stack([(w / 2 - xx) * 0.93, (h / 2 - yy) * 0.93], 0).unsqueeze(0).repeat(Bb, 1, 1, 1) + torch.randn((Bb, 2, h, w), generator=gen) * 0.3
    cot = torch.randn((Bb, Cc, h, w), generator=gen)
    a, f = img.clone().requires_grad_(), flow.clone().requires_grad_()
    gir, gfr = torch.autograd.grad(O.warp(a, f, True), (a, f), cot)
    with ops.kernel_policy(warp_bwd=lib.WARP_BWD_BINNED):
        x, fl = cl(img).requires_grad_(), flow.to(DEV).requires_grad_()
        gi, gf = torch.autograd.grad(ops.warp(x, fl, True), (x, fl), cl(cot))
    assert_close(gi, gir, 1e-4, 2e-5, "grad img (overflowing bins)")
    assert_close(gf, gfr, 1e-4, 2e-5, "grad flow", outliers=2e-3)


@pytest.mark.parametrize("shape", [(2, 40, 23, 45), (1, 196, 7, 16), (3, 8, 9, 33), (1, 32, 40, 70), (2, 64, 5, 130)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ac", [False, True])
@pytest.mark.parametrize("form", ["window", "binned"])
def test_warp_backward_gather_kernels_small_shapes(ops, lib, shape, dtype, ac, form):
    """the gather + per-source kernel pair forced on (policy OCF_WARP_BWD_POLICY_GATHER) at shapes the automatic policy would give to the scatter
    kernel: partial tiles, channel counts that are not a multiple of 32 (40, 196: 8-byte aligned pixels), fp32, both sampling
    conventions, flows with many FAR sources (|displacement| > 6 px: the float-atomic path inside the per-source kernel) and sources
    that leave the image -- against autograd of the oracle's grid_sample warp, and bit-identical grad_img on a second run"""
    Bb, Cc, h, w = shape
    gen = torch.Generator().manual_seed(sum(shape) + int(ac))
    img = torch.randn(shape, generator=gen).to(dtype)
    flow = torch.randn((Bb, 2, h, w), generator=gen) * 5.0
    cot = torch.randn(shape, generator=gen).to(dtype)
    a, f = img.float().requires_grad_(), flow.clone().requires_grad_()
    gir, gfr = torch.autograd.grad(O.warp(a, f, ac), (a, f), cot.float())
    # window form: sources within +-6 px gathered, the rest scattered with float atomics; binned form (round 3): every source listed in
    # the bins of the tiles its taps land in, nothing is "far"
    with ops.kernel_policy(warp_bwd=lib.WARP_BWD_GATHER if form == "window" else lib.WARP_BWD_BINNED):
        x, fl = cl(img).requires_grad_(), flow.to(DEV).requires_grad_()
        gi, gf = torch.autograd.grad(ops.warp(x, fl, ac), (x, fl), cl(cot))
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert_close(gi, gir, tol * 5, tol, "grad img")
    assert_close(gf, gfr, tol * 5, tol, "grad flow", outliers=2e-3)


@pytest.mark.parametrize("shape", [(8, 96, 28, 64), (8, 128, 14, 32), (8, 196, 7, 16), (2, 40, 23, 45), (3, 8, 9, 33), (1, 64, 5, 130)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ac", [False, True])
def test_warp_backward_single_launch_small_maps(ops, lib, shape, dtype, ac):
    """maps of at most 1792 pixels per image (the three coarsest pyramid levels of the benchmark and odd small shapes): ONE launch, a whole
    image as the gather window -- the plan says so (bit 1, no zero-fill), grad_img comes back in the feature dtype without a cast launch,
    large displacements and sources leaving the image included; against autograd of the oracle's grid_sample warp, and the same values as
    the scatter kernel up to float summation order"""
    Bb, Cc, h, w = shape
    gen = torch.Generator().manual_seed(sum(shape) + int(ac))
    img = torch.randn(shape, generator=gen).to(dtype)
    flow = torch.randn((Bb, 2, h, w), generator=gen) * 5.0
    cot = torch.randn(shape, generator=gen).to(dtype)
    a, f = img.float().requires_grad_(), flow.clone().requires_grad_()
    gir, gfr = torch.autograd.grad(O.warp(a, f, ac), (a, f), cot.float())
    x, fl = cl(img).requires_grad_(), flow.to(DEV).requires_grad_()
    from ocflow_amd.ops import _t4, _dtype_code
    gi32 = torch.empty(shape, device=DEV, dtype=torch.float32).contiguous(memory_format=torch.channels_last)
    gfl = torch.empty((Bb, 2, h, w), device=DEV, dtype=torch.float32)
    plan = lib.load().ocf_warp_bwd_plan(_t4(x.detach()), _t4(cl(cot)), _t4(gi32), _t4(gfl), Bb, Cc, h, w, _dtype_code(x), lib.WARP_BWD_AUTO)
    assert plan == 2, plan
    gi, gf = torch.autograd.grad(ops.warp(x, fl, ac), (x, fl), cl(cot))
    assert gi.dtype == dtype
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert_close(gi, gir, tol * 5, tol, "grad img")
    assert_close(gf, gfr, tol * 5, tol, "grad flow", outliers=2e-3)
    gi2, gf2 = torch.autograd.grad(ops.warp(x, fl, ac), (x, fl), cl(cot))
    assert torch.equal(gf, gf2)  # (grad_img: the order of a pixel's list, hence of its sum, is the order the LDS exchanges landed in)
    assert_close(gi2, gi.float(), tol * 5, tol, "grad img, second run")
    with ops.kernel_policy(warp_bwd=lib.WARP_BWD_SCATTER):
        gis, gfs = torch.autograd.grad(ops.warp(x, fl, ac), (x, fl), cl(cot))
    assert_close(gi, gis.float(), tol * 5, tol, "grad img vs scatter kernel")
    assert_close(gf, gfs, 1e-4, 2e-5, "grad flow vs scatter kernel", outliers=2e-3)
