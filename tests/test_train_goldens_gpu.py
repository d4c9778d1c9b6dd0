"""TRAIN-mode parity of the feature-pyramid models the benchmark trains (FlowOccModel('flowoccnet'), FlowStageModel('flownet')) against
reference goldens (tests/golden/train_fpn.npz, made by make_golden.py::gold_train_fpn from /root/reference): batch-statistics BatchNorm
through the HIP kernels with the two images of a pair as ONE batch of two statistics groups (`FeaturePyramidNet.forward_pair`), i.e. the
regrouped launches the benchmark times -- outputs, losses, parameter gradients (BatchNorm weight / bias among them) and the running
statistics after the step.  fp32, bar 1e-4 (north_star); reference: feature_pyramid_net.py:5-95, flow_occ_net.py:128-189,
flow_net.py:89-131, flow_occ_model.py:43-55, models/model.py:366-409."""
import numpy as np
import pytest
import torch

from golden_util import rngfree_fill, smooth_field

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy


@pytest.fixture(autouse=True)
def _deterministic():
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    yield


def _inputs(g):
    x = T(smooth_field((2, 6, 128, 192), seed=101, amp=1.0)).float().clamp(-1, 1)
    gt_flow = torch.randn((2, 2, 128, 192), generator=torch.Generator().manual_seed(102)) * 2
    gt_occ = (torch.rand((2, 1, 128, 192), generator=torch.Generator().manual_seed(103)) < 0.3).float()
    # the fixture stores checksums of its inputs: the regeneration here is the generator's
    assert abs(float(x.abs().mean()) - float(g["x_absmean"])) < 1e-6
    assert abs(float(gt_flow.abs().mean()) - float(g["gt_flow_absmean"])) < 1e-6
    assert abs(float(gt_occ.mean()) - float(g["gt_occ_mean"])) < 1e-7
    return x, gt_flow, gt_occ


def _epe(a, b):
    return float(torch.sqrt(((a.detach().float().cpu() - T(b)) ** 2).sum(1)).mean())


def _check_grads(params, g, prefix, tol=1e-4):
    """every stored gradient within `tol` of the reference, relative to that gradient's own scale"""
    keys = [k for k in g if k.startswith(prefix)]
    assert len(keys) >= 8
    worst = []
    for k in keys:
        name = k[len(prefix):]
        got = params[name].grad
        assert got is not None, name
        got = got.detach().float().cpu()
        ref = T(g[k])
        got = got[:8, :16] if got.dim() >= 2 else got
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        scale = float(ref.abs().max()) + 1e-12
        worst.append((float((got - ref).abs().max()) / scale, name))
    worst.sort(reverse=True)
    assert worst[0][0] < tol, worst[:4]


def _check_buffers(sd, g, prefix, root):
    for k in [k for k in g if k.startswith(prefix)]:
        name = root + k[len(prefix):]
        got, ref = sd[name].detach().cpu(), T(g[k])
        if ref.dtype in (torch.int64, torch.int32):
            assert int(got) == int(ref), name  # two encoder passes per forward = two updates
        else:
            np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=1e-5, err_msg=name)


@pytest.mark.parametrize("fmt", ["nchw", "channels_last"])
def test_flowoccnet_supervised_train_step_matches_reference(golden, fmt):
    from ocflow_amd.models.flow_occ_model import FlowOccModel
    g = golden("train_fpn")
    x, gt_flow, gt_occ = _inputs(g)
    mf = torch.channels_last if fmt == "channels_last" else torch.contiguous_format
    m = rngfree_fill(FlowOccModel({"learning_rate": 1e-3, "model": "flowoccnet"})).to(DEV).to(memory_format=mf).train()
    seen = []
    h = m.model.register_forward_hook(lambda mod, inp, out: seen.append(out))
    fl_loss, oc_loss = m.general_step((x.to(DEV).contiguous(memory_format=mf), gt_flow.to(DEV), gt_occ.to(DEV)), 0, "train")
    h.remove()
    (fl_loss + oc_loss).backward()
    fl, oc = seen[0]
    assert _epe(fl, g["occ_flow"]) < 1e-4
    assert float((oc.detach().float().cpu() - T(g["occ_occ"])).abs().mean()) < 1e-4
    assert abs(float(fl_loss) - float(g["occ_flow_loss"])) < 1e-4 * max(1.0, abs(float(g["occ_flow_loss"])))
    assert abs(float(oc_loss) - float(g["occ_occ_loss"])) < 1e-4 * max(1.0, abs(float(g["occ_occ_loss"])))
    _check_grads(dict(m.named_parameters()), g, "occ_grad__")
    _check_buffers(m.state_dict(), g, "occ_buf__", "model.")


@pytest.mark.parametrize("fmt", ["nchw", "channels_last"])
def test_flownet_occlusion_aware_train_step_matches_reference(golden, fmt):
    from ocflow_amd.models.model import FlowStageModel
    g = golden("train_fpn")
    x, gt_flow, gt_occ = _inputs(g)
    mf = torch.channels_last if fmt == "channels_last" else torch.contiguous_format
    hp = dict(learning_rate=1e-5, model="flownet", displacement=4, photo_weight=4.0, smooth1_weight=0.5, smooth2_weight=0.0,
              with_occ=False, occ_aware=True)
    m = rngfree_fill(FlowStageModel(hp)).to(DEV).to(memory_format=mf).train()
    seen = []
    h = m.flow_pred.register_forward_hook(lambda mod, inp, out: seen.append(out))
    batch = (x.to(DEV).contiguous(memory_format=mf), gt_flow.to(DEV), gt_occ.to(DEV))
    res = m.general_step_occ_aware(batch, 0, "train")
    h.remove()
    assert _epe(seen[0], g["stage_flow"]) < 1e-4 and _epe(seen[1], g["stage_back_flow"]) < 1e-4
    np.testing.assert_allclose(torch.stack([r.detach() for r in res]).float().cpu().numpy(), g["stage_scalars"], rtol=1e-4, atol=1e-5)
    loss = hp["photo_weight"] * res[0] + hp["smooth1_weight"] * res[1] + hp["smooth2_weight"] * res[2]
    assert abs(float(loss) - float(g["stage_loss"])) < 1e-4 * max(1.0, abs(float(g["stage_loss"])))
    loss.backward()
    _check_grads(dict(m.named_parameters()), g, "stage_grad__")
    _check_buffers(m.state_dict(), g, "stage_buf__", "flow_pred.")


def test_bf16_pairing_and_padding_track_their_own_fp32_run(golden):
    """the throughput configuration (bf16 autocast, channels_last, paired encoder, 32-channel aligned concatenations) against an fp32 run of
    the SAME regrouping: outputs and the stored gradients differ only by bf16 rounding (a statistics or channel-mapping error would be O(1))."""
    from ocflow_amd.models.flow_occ_model import FlowOccModel
    g = golden("train_fpn")
    x, gt_flow, gt_occ = _inputs(g)
    runs = []
    for amp in (False, True):
        m = rngfree_fill(FlowOccModel({"learning_rate": 1e-3, "model": "flowoccnet"})).to(DEV).to(memory_format=torch.channels_last).train()
        seen = []
        h = m.model.register_forward_hook(lambda mod, inp, out: seen.append(out))
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            fl_loss, oc_loss = m.general_step((x.to(DEV).contiguous(memory_format=torch.channels_last), gt_flow.to(DEV), gt_occ.to(DEV)), 0, "train")
        h.remove()
        (fl_loss.float() + oc_loss.float()).backward()
        runs.append((seen[0][0].detach().float(), seen[0][1].detach().float(), float(fl_loss), float(oc_loss),
                     {k: p.grad.detach().float() for k, p in m.named_parameters() if p.grad is not None}))
    (f0, o0, l0, c0, g0), (f1, o1, l1, c1, g1) = runs
    assert float((f0 - f1).abs().mean()) < 0.02 * float(f0.abs().mean()) + 0.02
    assert float((o0 - o1).abs().mean()) < 0.02
    assert abs(l0 - l1) < 0.02 * abs(l0) + 1e-3 and abs(c0 - c1) < 0.03 * abs(c0) + 1e-3
    # gradients: direction and size.  (bf16 element errors are amplified through five warped pyramid levels; cosine similarity and the
    # norm ratio are what a wrong statistics group or a shifted channel would break)
    bad = []
    for k in g0:
        a, b = g0[k].flatten(), g1[k].flatten()
        if float(a.norm()) < 1e-8:
            continue
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        ratio = float(b.norm() / a.norm())
        if cos < 0.9 or not (0.8 < ratio < 1.25):
            bad.append((k, round(cos, 3), round(ratio, 3)))
    assert len(bad) <= max(2, len(g0) // 50), bad[:10]
