#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REFERENCE on CPU.

Run in the build container only (the reference lives at /root/reference and never
travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What this script does
  * puts /root/reference on sys.path (read-only use, no bytecode written),
  * installs in-process stand-ins for packages the container lacks and that the
    reference imports at module level (pytorch_lightning, torchvision, cv2, imageio,
    PIL).  They only have to make the `import` lines succeed; none of their code is on
    the hot path,
  * installs the 6-line `models.networks.cost_volume_net.CostVolumeLayer` stand-in
    (that module is imported by the reference but missing from it):
    CostVolumeLayer(d).forward(f1, f2) == compute_cost_volume(f1, f2, d).  THIS IS OUR
    DEFINITION and is recorded in every fixture's metadata,
  * runs the hot-path ops on small seeded inputs and writes `*.npz` files next to this
    script: inputs, outputs and gradients (data only - no reference source).

Whole-network fixtures use an RNG-free weight rule (`rngfree_fill`, restated in
tests/golden_util.py) so the 10-40 MB of weights never have to be stored.
"""
import sys
import os
import types
import math

sys.dont_write_bytecode = True
REF = os.environ.get("OCFLOW_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(HERE))  # tests/ for golden_util

import numpy as np
import torch
import torch.nn as nn

from golden_util import rngfree_fill, seeded_fill, smooth_field  # noqa: E402

torch.set_num_threads(4)


# ----------------------------------------------------------------------------- stubs
def _install_stubs():
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        global_step = 0
        current_epoch = 0
        logger = None

        def log(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    pl.Trainer = object
    pl.seed_everything = lambda s: torch.manual_seed(s)
    sys.modules["pytorch_lightning"] = pl
    for name in ("pytorch_lightning.loggers", "pytorch_lightning.callbacks"):
        sys.modules[name] = types.ModuleType(name)

    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.utils = types.ModuleType("torchvision.utils")
    tv.transforms.Compose = tv.transforms.ToTensor = tv.transforms.Normalize = object
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tv.models
    sys.modules["torchvision.transforms"] = tv.transforms
    sys.modules["torchvision.utils"] = tv.utils

    for name in ("cv2", "imageio"):
        m = types.ModuleType(name)
        m.imread = None
        sys.modules[name] = m
    try:
        import PIL  # noqa: F401
    except Exception:
        pil = types.ModuleType("PIL")
        pil.Image = types.ModuleType("PIL.Image")
        sys.modules["PIL"] = pil
        sys.modules["PIL.Image"] = pil.Image

    # the module the reference imports but does not ship
    from models.networks.correlation_layer import compute_cost_volume

    cvn = types.ModuleType("models.networks.cost_volume_net")

    class CostVolumeLayer(nn.Module):
        def __init__(self, max_displacement=4):
            super().__init__()
            self.max_displacement = max_displacement

        def forward(self, f1, f2):
            return compute_cost_volume(f1, f2, self.max_displacement)

    cvn.CostVolumeLayer = CostVolumeLayer
    sys.modules["models.networks.cost_volume_net"] = cvn


_install_stubs()

from models.networks.correlation_layer import compute_cost_volume, normalize_features  # noqa: E402
from models.networks.cost_volume_flow_net import FlowNetCV  # noqa: E402
import models.model as ref_model  # noqa: E402
from models.flow_model import FlowModel  # noqa: E402
from models.flow_occ_model import FlowOccModel  # noqa: E402
from models.data.utils import flow_utils as ref_flow_utils  # noqa: E402

META = dict(
    reference="dongliangcao/OCFlow @ /root/reference",
    torch=torch.__version__,
    cost_volume_layer="CostVolumeLayer(d).forward(f1,f2) := compute_cost_volume(f1,f2,d) (our stand-in for the missing module)",
)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    out["_meta"] = np.array(repr(META))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in out.items() if k != "_meta"})


def g(seed):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------ 1. cost volume (a1)
def gold_corr():
    cases = {"a": ((2, 8, 6, 7), 1), "b": ((2, 8, 6, 7), 4), "c": ((1, 32, 12, 20), 4), "d": ((1, 5, 3, 4), 2)}
    arrs = {}
    for tag, (shape, d) in cases.items():
        f1 = torch.randn(shape, generator=g(11)).requires_grad_()
        f2 = torch.randn(shape, generator=g(12)).requires_grad_()
        out = compute_cost_volume(f1, f2, d)
        cot = torch.randn(out.shape, generator=g(13))
        gf1, gf2 = torch.autograd.grad(out, (f1, f2), cot)
        arrs.update({f"{tag}_f1": f1, f"{tag}_f2": f2, f"{tag}_d": np.int64(d), f"{tag}_out": out,
                     f"{tag}_cot": cot, f"{tag}_gf1": gf1, f"{tag}_gf2": gf2})
    save("corr", **arrs)


# ------------------------------------------------------------ 2. feature normalise (a3)
NORM_FLAG_CASES = {
    "per_image": dict(moments_across_images=False),
    "per_channel": dict(moments_across_channels=False),
    "per_channel_image": dict(moments_across_channels=False, moments_across_images=False),
    "center_only": dict(normalize=False),
    "scale_only": dict(center=False),
    "scale_only_per_image": dict(center=False, moments_across_images=False),
    "identity": dict(normalize=False, center=False),
}


def gold_norm():
    f1 = (torch.randn((2, 16, 5, 6), generator=g(21)) * 2 + 0.3).requires_grad_()
    f2 = (torch.randn((2, 16, 5, 6), generator=g(22)) * 0.5 - 1.0).requires_grad_()
    n1, n2 = normalize_features([f1, f2])
    c1 = torch.randn(n1.shape, generator=g(23))
    c2 = torch.randn(n2.shape, generator=g(24))
    gf1, gf2 = torch.autograd.grad((n1 * c1).sum() + (n2 * c2).sum(), (f1, f2))
    save("norm", f1=f1, f2=f2, n1=n1, n2=n2, c1=c1, c2=c2, gf1=gf1, gf2=gf2)
    # the non-default flag combinations of correlation_layer.py:42-82 (no call site of the reference uses them; the operator accepts them)
    arrs = dict(f1=f1.detach(), f2=f2.detach(), c1=c1, c2=c2)
    for tag, kw in NORM_FLAG_CASES.items():
        a, b = f1.detach().clone().requires_grad_(), f2.detach().clone().requires_grad_()
        m1, m2 = normalize_features([a, b], **kw)
        ga, gb = torch.autograd.grad((m1 * c1).sum() + (m2 * c2).sum(), (a, b), allow_unused=True)
        arrs.update({f"{tag}_n1": m1, f"{tag}_n2": m2, f"{tag}_gf1": ga if ga is not None else torch.zeros_like(a),
                     f"{tag}_gf2": gb if gb is not None else torch.zeros_like(b)})
    save("norm_flags", **arrs)


# ------------------------------------------------------------------- 3. warps (a4, a5)
def gold_warp():
    arrs = {}
    shapes = {"main": (2, 5, 13, 17), "h1": (1, 3, 1, 9), "w1": (1, 3, 7, 1), "s22": (2, 4, 2, 2), "s11": (1, 2, 1, 1)}
    for tag, shape in shapes.items():
        B, C, H, W = shape
        img = torch.randn(shape, generator=g(31)).requires_grad_()
        flow = (torch.randn((B, 2, H, W), generator=g(32)) * 3.0).requires_grad_()
        cot = torch.randn(shape, generator=g(33))
        # AC-False: the in-network feature warp  (cost_volume_flow_net.py:121-151)
        o0 = FlowNetCV.warp(None, img, flow)
        gi0, gf0 = torch.autograd.grad(o0, (img, flow), cot)
        # AC-True: the loss-side image warp       (flow_model.py:49-79)
        o1 = FlowModel.warp(None, img, flow)
        gi1, gf1 = torch.autograd.grad(o1, (img, flow), cot)
        arrs.update({f"{tag}_img": img, f"{tag}_flow": flow, f"{tag}_cot": cot,
                     f"{tag}_out_ac0": o0, f"{tag}_gimg_ac0": gi0, f"{tag}_gflow_ac0": gf0,
                     f"{tag}_out_ac1": o1, f"{tag}_gimg_ac1": gi1, f"{tag}_gflow_ac1": gf1})
    save("warp", **arrs)


# ---------------------------------------------------------------- 4. losses (a6, a7, a8)
def gold_losses():
    B, H, W = 2, 16, 24
    img1 = torch.rand((B, 3, H, W), generator=g(41)) * 2 - 1
    img2 = torch.rand((B, 3, H, W), generator=g(42)) * 2 - 1
    # image-like (smooth) variants so that the edge weights exp(-(100 dI)^2) are not all ~0
    simg = torch.from_numpy(smooth_field((B, 3, H, W), seed=43, amp=0.02)).float()
    flow = (torch.randn((B, 2, H, W), generator=g(44)) * 2.0).requires_grad_()
    occ = (torch.rand((B, 1, H, W), generator=g(45)) < 0.3).float()
    arrs = dict(img1=img1, img2=img2, simg=simg, flow=flow, occ=occ)

    pred = (img2 + 0.1 * torch.randn(img2.shape, generator=g(46))).requires_grad_()
    for tag, o in (("none", None), ("occ", occ)):
        L = ref_model.photometric_error(pred, img1, o)
        (gp,) = torch.autograd.grad(L, pred)
        arrs[f"photo_{tag}"] = L
        arrs[f"photo_{tag}_gpred"] = gp
    arrs["pred"] = pred

    # fused chain the training step uses: warp(AC-True) -> photometric(+occ)   (model.py:379,394)
    wrapper = types.SimpleNamespace()
    for tag, o in (("none", None), ("occ", occ)):
        warped = ref_model.FlowStageModel.warp(wrapper, img2, flow)
        L = ref_model.photometric_error(warped, img1, o)
        (gfl,) = torch.autograd.grad(L, flow)
        arrs[f"warpphoto_{tag}"] = L
        arrs[f"warpphoto_{tag}_gflow"] = gfl

    for name, fn in (("smooth1", ref_model.first_order_smoothness_loss), ("smooth2", ref_model.second_order_smoothness_loss)):
        for itag, im in (("noise", img1), ("smooth", simg)):
            L = fn(im, flow)
            (gfl,) = torch.autograd.grad(L, flow)
            arrs[f"{name}_{itag}"] = L
            arrs[f"{name}_{itag}_gflow"] = gfl
    save("losses", **arrs)


# ------------------------------------------------------------------- 5. range map (a9)
def gold_range():
    arrs = {}
    for tag, shape, amp in (("a", (2, 2, 8, 9), 2.0), ("b", (1, 2, 5, 16), 6.0), ("zero", (1, 2, 4, 4), 0.0)):
        flow = torch.randn(shape, generator=g(51)) * amp
        rm = FlowModel.compute_range_map(types.SimpleNamespace(flow_to_warp=lambda f: FlowModel.flow_to_warp(None, f)), flow)
        occ = 1.0 - torch.clamp(rm, min=0.0, max=1.0)
        arrs.update({f"{tag}_flow": flow, f"{tag}_range": rm, f"{tag}_occ": occ})
    save("range_map", **arrs)


# ------------------------------------------------------- 6./7. whole networks + train step
def _net_fixture(model, x, mode):
    rngfree_fill(model)
    model.train(mode == "train")
    with torch.no_grad():
        out = model(x)
    return out


def gold_nets():
    arrs = {}
    x64 = torch.rand((1, 6, 64, 64), generator=g(61)) * 2 - 1
    x64b2 = torch.rand((2, 6, 64, 64), generator=g(62)) * 2 - 1
    x96 = torch.from_numpy(smooth_field((1, 6, 64, 96), seed=63, amp=1.0)).float().clamp(-1, 1)
    arrs.update(x64=x64, x64b2=x64b2, x96=x96)

    manifests = {}
    # Cfg1: models.flow_model forward on 2 random 64x64x3 frames (BASELINE.json configs[0])
    for name, x, mode in (("simple", x64b2, "train"), ("simple", x64, "eval"), ("pwc", x64, "train"), ("flownet", x96, "eval")):
        m = FlowModel({"learning_rate": 1e-3, "model": name, "displacement": 4})
        out = _net_fixture(m, x, mode)
        arrs[f"flowmodel_{name}_{mode}"] = out
        manifests[f"flowmodel_{name}"] = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    # pwc second output (quarter-res flow * 5)
    m = FlowModel({"learning_rate": 1e-3, "model": "pwc"})
    rngfree_fill(m)
    with torch.no_grad():
        f_full, f_q = m.flow_pred(x64)
    arrs["flownetcv_full"], arrs["flownetcv_quarter"] = f_full, f_q

    for name, x, mode in (("flowoccnet", x96, "eval"), ("pwoc", x64, "train"), ("simple", x64b2, "train")):
        m = FlowOccModel({"learning_rate": 1e-3, "model": name})
        fl, oc = _net_fixture(m, x, mode)
        arrs[f"flowoccmodel_{name}_{mode}_flow"] = fl
        arrs[f"flowoccmodel_{name}_{mode}_occ"] = oc
        manifests[f"flowoccmodel_{name}"] = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]

    # supervised steps with gradients (a12)
    m = FlowOccModel({"learning_rate": 1e-3, "model": "pwoc"})
    rngfree_fill(m)
    m.train()
    gt_flow = torch.randn((1, 2, 64, 64), generator=g(64))
    gt_occ = (torch.rand((1, 1, 64, 64), generator=g(65)) < 0.3).float()
    fl_loss, oc_loss = m.general_step((x64, gt_flow, gt_occ), 0, "train")
    (fl_loss + oc_loss).backward()
    arrs.update(pwoc_step_gt_flow=gt_flow, pwoc_step_gt_occ=gt_occ, pwoc_step_flow_loss=fl_loss, pwoc_step_occ_loss=oc_loss,
                pwoc_step_grad_conv2_0=m.model.conv2_0[0].weight.grad[:4, :8], pwoc_step_grad_conv1a=m.model.conv1a[0].weight.grad)

    # unsupervised step (a10): FlowStageModel pwc, occ_aware, config weights
    hp = dict(learning_rate=1e-5, model="pwc", displacement=4, photo_weight=4.0, smooth1_weight=0.5, smooth2_weight=0.0,
              with_occ=False, occ_aware=True)
    xs = torch.from_numpy(smooth_field((2, 6, 64, 128), seed=66, amp=1.0)).float().clamp(-1, 1)
    gt_flow2 = torch.randn((2, 2, 64, 128), generator=g(67)) * 2
    gt_occ2 = (torch.rand((2, 1, 64, 128), generator=g(68)) < 0.3).float()
    for occ_aware, tag in ((True, "occaware"), (False, "plain")):
        hp["occ_aware"] = occ_aware
        m = ref_model.FlowStageModel(dict(hp))
        rngfree_fill(m)
        m.train()
        m.logger = None
        m.global_step = 1  # skip tensorboard
        if occ_aware:
            res = m.general_step_occ_aware((xs, gt_flow2, gt_occ2), 0, "train")
        else:
            res = m.general_step((xs, gt_flow2, gt_occ2), 0, "train")
        loss = hp["photo_weight"] * res[0] + hp["smooth1_weight"] * res[1] + hp["smooth2_weight"] * res[2]
        loss.backward()
        arrs[f"stage_{tag}_scalars"] = torch.stack([r.detach() for r in res])
        arrs[f"stage_{tag}_loss"] = loss
        arrs[f"stage_{tag}_grad_predict_flow2"] = m.flow_pred.predict_flow2.weight.grad[:, :16]
        arrs[f"stage_{tag}_grad_conv1a"] = m.flow_pred.conv1a[0].weight.grad
    arrs.update(stage_x=xs, stage_gt_flow=gt_flow2, stage_gt_occ=gt_occ2)
    manifests["flowstage_pwc"] = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    save("nets", **arrs)

    import json
    with open(os.path.join(HERE, "state_dict_manifest.json"), "w") as f:
        json.dump(manifests, f)
    print("wrote state_dict_manifest.json", {k: len(v) for k, v in manifests.items()})


# -------------------------------------------------------------------------- 8. EPE
def gold_epe():
    rs = np.random.RandomState(81)
    tu, tv, u, v = [rs.randn(8, 9).astype(np.float32) * 3 for _ in range(4)]
    tu[0, 0] = 1e9  # unknown-flow sentinel path
    occ = (rs.rand(8, 9) < 0.3).astype(np.float32)
    e0 = ref_flow_utils.flow_error(tu.copy(), tv.copy(), u.copy(), v.copy())
    # the occ path indexes the EPE map with a flattened mask (flow_utils.py:227-230) => 1-D inputs
    e1 = ref_flow_utils.flow_error(tu.reshape(-1).copy(), tv.reshape(-1).copy(), u.reshape(-1).copy(), v.reshape(-1).copy(), occ)
    # .flo wire format written by the reference's save_flow, read back by its read_flow (flow_utils.py:10-68)
    import tempfile
    flo = (rs.randn(7, 11, 2) * 4).astype(np.float32)
    with tempfile.TemporaryDirectory() as td:
        fn = os.path.join(td, "x.flo")
        ref_flow_utils.save_flow(fn, flo)
        raw = np.frombuffer(open(fn, "rb").read(), dtype=np.uint8)
        back = ref_flow_utils.read_flow(fn)
    save("epe", tu=tu, tv=tv, u=u, v=v, occ=occ, epe=np.float64(e0), epe_occ=np.float64(e1), flo=flo, flo_bytes=raw, flo_read=back)


# ------------------------------------------------------------- 9. SSIM metric
def gold_ssim():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_ssim", os.path.join(REF, "inpainting_metrics", "ssim", "ssim.py"))
    ref_ssim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_ssim)  # the package __init__ pulls FID/Inception; the file itself needs torch only
    a = torch.rand((2, 3, 16, 20), generator=g(91)) * 2 - 1
    b = (a + 0.2 * torch.randn(a.shape, generator=g(92))).clamp(-1, 1)
    arrs = dict(a=a, b=b)
    for ws in (4, 11):  # 4 = what calculate_ssim uses (inpainting_metrics/__init__.py:23): even window => (H+1)x(W+1) map
        arrs[f"ssim_ws{ws}_mean"] = ref_ssim.ssim(a, b, window_size=ws, size_average=True)
        arrs[f"ssim_ws{ws}_per_sample"] = ref_ssim.ssim(a, b, window_size=ws, size_average=False)
    arrs["ssim_identical"] = ref_ssim.ssim(a, a, window_size=11)
    # gradients (SSIM as a differentiable photometric term): autograd of the reference's own _ssim
    for ws in (4, 11):
        ag, bg = a.clone().requires_grad_(), b.clone().requires_grad_()
        ga, gb = torch.autograd.grad(ref_ssim.ssim(ag, bg, window_size=ws, size_average=True), (ag, bg))
        arrs[f"ssim_ws{ws}_grad_a"], arrs[f"ssim_ws{ws}_grad_b"] = ga, gb
    cw = torch.tensor([0.3, -1.7])
    ag = a.clone().requires_grad_()
    (ga,) = torch.autograd.grad((ref_ssim.ssim(ag, b, window_size=11, size_average=False) * cw).sum(), ag)
    arrs["ssim_ws11_weights"], arrs["ssim_ws11_grad_a_weighted"] = cw, ga
    save("ssim", **arrs)


# ------------------------------------------------------------- 10. PSNR metric (sibling of SSIM)
def gold_psnr():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_psnr", os.path.join(REF, "inpainting_metrics", "psnr", "psnr.py"))
    ref_psnr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_psnr)
    a = torch.rand((3, 3, 16, 20), generator=g(93)) * 2 - 1
    b = (a + 0.1 * torch.randn(a.shape, generator=g(94))).clamp(-1, 1)
    b[2] = a[2]  # identical pair: the reference returns 100
    # exactly what calculate_psnr feeds the function (inpainting_metrics/__init__.py:76-78): per image, 0..255 scale, numpy
    vals = [ref_psnr.psnr((x.numpy() + 1) * 127.5, (y.numpy() + 1) * 127.5) for x, y in zip(a, b)]
    save("psnr", a=a, b=b, psnr=np.asarray(vals, dtype=np.float64))


# ------------------------------------------- 11. whole networks at the benchmark's resolution (BASELINE configs[1])
def _sub(t, k=4):
    """every k-th pixel in y and x: the fixture keeps 1/16 of a full-resolution map (the input is regenerated by the test from
    golden_util.smooth_field, the weights from rngfree_fill, so nothing else has to be stored)"""
    return t[..., ::k, ::k].contiguous()


def gold_fullsize():
    arrs = {}
    # FlowOccModel('flowoccnet') at the native Sintel size 1 x 6 x 436 x 1024 (FPN nets run there), eval mode
    x = torch.from_numpy(smooth_field((1, 6, 436, 1024), seed=71, amp=1.0)).float().clamp(-1, 1)
    m = FlowOccModel({"learning_rate": 1e-3, "model": "flowoccnet"})
    rngfree_fill(m)
    m.eval()
    with torch.no_grad():
        fl, oc = m(x)
    arrs.update(flowoccnet_436x1024_flow_sub4=_sub(fl), flowoccnet_436x1024_occ_sub4=_sub(oc),
                flowoccnet_436x1024_flow_absmean=fl.abs().mean(), flowoccnet_436x1024_occ_mean=oc.mean())
    # FlowNetCV ('pwc') at 1 x 6 x 448 x 1024 (sides divisible by 64), train mode like the Cfg1 fixture
    x2 = torch.from_numpy(smooth_field((1, 6, 448, 1024), seed=72, amp=1.0)).float().clamp(-1, 1)
    m = FlowModel({"learning_rate": 1e-3, "model": "pwc", "displacement": 4})
    rngfree_fill(m)
    with torch.no_grad():
        full, quarter = m.flow_pred(x2)
    arrs.update(flownetcv_448x1024_full_sub4=_sub(full), flownetcv_448x1024_quarter=quarter, flownetcv_448x1024_full_absmean=full.abs().mean())
    save("fullsize", **arrs)


# ------------------------------------------- 12. inpainting head (BASELINE configs[3]: KITTI-sized pairs, occlusion mask)
def gold_inpainting():
    from models.inpainting_model import InpaintingModel
    from models.networks.image_inpainting_net import InpaintingNet
    arrs = {}
    manifest = None

    def chain(net, imgs, flow, occ):
        """models/inpainting_model.py:62-73 with the ONE fix the reference needs to run at all: `forward` must hand the
        mask to InpaintingNet -- `self.model(img_occluded, occ)` instead of `self.model(img_occluded)`.  Everything else is the
        reference's own code: its warp method, its network class, its loss expression."""
        img1, img2 = imgs[:, 0:3], imgs[:, 3:6]
        img_warped = InpaintingModel.warp(None, img2, flow)
        img_occluded = img_warped * (1 - occ)
        img_completed = net(img_occluded, occ)
        loss = (torch.abs(img_completed - img1) * occ).sum() / (3 * occ.sum() + 1e-16)
        return img_warped, img_completed, loss

    # (a) small: everything stored, gradients of the network parameters and of the flow-free inputs
    net = InpaintingNet()
    rngfree_fill(net)
    net.train()
    imgs = torch.from_numpy(smooth_field((2, 6, 64, 96), seed=81, amp=1.0)).float().clamp(-1, 1)
    flow = torch.randn((2, 2, 64, 96), generator=g(82)) * 2.0
    occ = (torch.rand((2, 1, 64, 96), generator=g(83)) < 0.3).float()
    warped, completed, loss = chain(net, imgs, flow, occ)
    loss.backward()
    arrs.update(s_imgs=imgs, s_flow=flow, s_occ=occ, s_warped=warped, s_completed=completed, s_loss=loss,
                s_grad_down1_conv1=net.down1.conv1.weight.grad, s_grad_up6_conv3=net.up6.conv3.weight.grad)
    manifest = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    # (b) KITTI render size 4 x 320 x 1216 (375 x 1242 through the reference's crop rule, datasets.py:146-150), eval mode
    net = InpaintingNet()
    rngfree_fill(net)
    net.eval()
    imgs = torch.from_numpy(smooth_field((4, 6, 320, 1216), seed=84, amp=1.0)).float().clamp(-1, 1)
    flow = torch.from_numpy(smooth_field((4, 2, 320, 1216), seed=85, amp=6.0)).float()
    occ = (torch.from_numpy(smooth_field((4, 1, 320, 1216), seed=86, amp=1.0)) > 0.35).float()
    with torch.no_grad():
        warped, completed, loss = chain(net, imgs, flow, occ)
    arrs.update(k_loss=loss, k_completed_sub8=_sub(completed, 8), k_warped_sub8=_sub(warped, 8), k_occ_fraction=occ.mean())
    # (c) op level at the raw KITTI size 375 x 1242 (no network: warp -> mask -> masked L1 against a given prediction)
    img2 = torch.from_numpy(smooth_field((4, 3, 375, 1242), seed=87, amp=1.0)).float().clamp(-1, 1)
    img1 = torch.from_numpy(smooth_field((4, 3, 375, 1242), seed=88, amp=1.0)).float().clamp(-1, 1)
    flow = torch.from_numpy(smooth_field((4, 2, 375, 1242), seed=89, amp=8.0)).float()
    occ = (torch.from_numpy(smooth_field((4, 1, 375, 1242), seed=90, amp=1.0)) > 0.3).float()
    pred = InpaintingModel.warp(None, img2, flow) * (1 - occ)
    pred.requires_grad_()
    loss = (torch.abs(pred - img1) * occ).sum() / (3 * occ.sum() + 1e-16)
    loss2 = (torch.abs(pred * 0.5 - img1) * occ).sum() / (3 * occ.sum() + 1e-16)
    (gp,) = torch.autograd.grad(loss2, pred)
    arrs.update(o_masked_sub8=_sub(pred.detach(), 8), o_loss=loss, o_loss_half=loss2, o_grad_half_sub8=_sub(gp, 8))
    save("inpainting", **arrs)
    import json
    with open(os.path.join(HERE, "state_dict_manifest_inpainting.json"), "w") as f:
        json.dump({"inpaintingnet": manifest}, f)

# ------------------------------------------- 13. FPN nets in TRAIN mode (batch-statistics BatchNorm): what the benchmark trains
def _grad_pick(g):
    """a gradient as stored in the fixture: whole when small, otherwise the leading [<=8, <=16, ...] block"""
    if g.dim() >= 2:
        return g[:8, :16]
    return g


# Gradients kept by the train-step fixtures: encoder (first convolution, BatchNorm weight / bias at the finest and coarsest level), the
# coarsest and the finest estimators, the context network.  Every entry carries `__noise` = how far the reference's own fp32 run is from its
# fp64 run (relative to the gradient's scale); with `seeded_fill` weights that is 1e-7 .. 4e-3 (largest for the encoder, whose gradient
# passes through all five warped levels).  With the sinusoidal `rngfree_fill` weights the same figure was 25-50 %: nearly rank-deficient
# feature maps make batch statistics ill-conditioned, which is why these fixtures do not use them.
TRAIN_FPN_GRADS = {
    "flowoccnet": ("model.feature_pyramid_network.layer1.double_conv.0.weight", "model.feature_pyramid_network.layer1.double_conv.1.weight",
                   "model.feature_pyramid_network.layer1.double_conv.1.bias", "model.feature_pyramid_network.layer6.double_conv.4.weight",
                   "model.occlusion_estimators.0.conv1.weight", "model.opticalflow_estimators.0.conv2.weight",
                   "model.context_network.conv4.weight", "model.context_network.conv7.weight", "model.context_network.conv7.bias",
                   "model.context_network.conv1.weight", "model.occlusion_estimators.4.mask_layer.weight",
                   "model.occlusion_estimators.4.conv4.weight", "model.opticalflow_estimators.4.conv6.weight",
                   "model.opticalflow_estimators.4.conv5.bias"),
    "flownet": ("flow_pred.feature_pyramid_network.layer1.double_conv.0.weight", "flow_pred.feature_pyramid_network.layer1.double_conv.1.weight",
                "flow_pred.feature_pyramid_network.layer6.double_conv.4.bias", "flow_pred.opticalflow_estimators.0.conv2.weight",
                "flow_pred.context_network.conv7.weight", "flow_pred.context_network.conv6.weight", "flow_pred.context_network.conv7.bias",
                "flow_pred.context_network.conv3.weight", "flow_pred.opticalflow_estimators.4.conv6.weight",
                "flow_pred.opticalflow_estimators.4.conv5.weight"),
}
TRAIN_FPN_BUFFERS = ("feature_pyramid_network.layer1.double_conv.1.running_mean", "feature_pyramid_network.layer1.double_conv.1.running_var",
                     "feature_pyramid_network.layer6.double_conv.4.running_mean", "feature_pyramid_network.layer6.double_conv.4.running_var",
                     "feature_pyramid_network.layer6.double_conv.4.num_batches_tracked")


def gold_train_fpn():
    """One TRAINING step of the two feature-pyramid models the benchmark runs (BASELINE configs[1] / the unsupervised leg), B = 2 at
    128 x 192 so that the coarsest level keeps 2 x 3 pixels per sample: outputs, losses, parameter gradients (a BatchNorm weight and bias
    among them) and the BatchNorm running statistics AFTER the step (the reference runs the encoder once per image, i.e. two updates
    per forward: feature_pyramid_net.py:76-93 called at flow_occ_net.py:133-134 / flow_net.py:100-101).

    Weights: golden_util.seeded_fill (see the note above TRAIN_FPN_GRADS).  The fixture stores the reference evaluated in fp64 (rounded
    to fp32 for storage) as the value and, for every quantity, `<name>__noise` = the deviation of the reference's own fp32 run from it;
    the test accepts max(1e-4, 3 x noise)."""
    arrs = {}
    x = torch.from_numpy(smooth_field((2, 6, 128, 192), seed=101, amp=1.0)).float().clamp(-1, 1)
    gt_flow = torch.randn((2, 2, 128, 192), generator=g(102)) * 2
    gt_occ = (torch.rand((2, 1, 128, 192), generator=g(103)) < 0.3).float()
    # inputs are regenerated by the test from the same seeds (tests/test_train_goldens_gpu.py); their checksums pin the regeneration
    arrs.update(x_absmean=x.abs().mean(), gt_flow_absmean=gt_flow.abs().mean(), gt_occ_mean=gt_occ.mean())

    def epe(a, b):
        return ((a.double() - b.double()) ** 2).sum(1).sqrt().mean()

    def run_occ(dt):
        # (a) supervised step of FlowOccModel('flowoccnet'): flow_occ_model.py:43-55
        m = FlowOccModel({"learning_rate": 1e-3, "model": "flowoccnet"})
        seeded_fill(m, 11)
        m = m.to(dt).train()
        seen = []
        h = m.model.register_forward_hook(lambda mod, inp, out: seen.append(out))
        fl_loss, oc_loss = m.general_step((x.to(dt), gt_flow.to(dt), gt_occ.to(dt)), 0, "train")
        h.remove()
        (fl_loss + oc_loss).backward()
        params = dict(m.named_parameters())
        sd = m.state_dict()
        out = dict(occ_flow=seen[0][0], occ_occ=seen[0][1], occ_flow_loss=fl_loss, occ_occ_loss=oc_loss)
        out.update({"occ_grad__" + k: _grad_pick(params[k].grad) for k in TRAIN_FPN_GRADS["flowoccnet"]})
        out.update({"occ_buf__" + k: sd["model." + k] for k in TRAIN_FPN_BUFFERS})
        return {k: v.detach() for k, v in out.items()}

    hp = dict(learning_rate=1e-5, model="flownet", displacement=4, photo_weight=4.0, smooth1_weight=0.5, smooth2_weight=0.0,
              with_occ=False, occ_aware=True)

    def run_stage(dt):
        # (b) unsupervised occlusion-aware step of FlowStageModel('flownet'): model.py:366-409, config weights of unsupervised_config.yml
        m = ref_model.FlowStageModel(dict(hp))
        seeded_fill(m, 11)
        m = m.to(dt).train()
        seen = []
        h = m.flow_pred.register_forward_hook(lambda mod, inp, out: seen.append(out))
        res = m.general_step_occ_aware((x.to(dt), gt_flow.to(dt), gt_occ.to(dt)), 0, "train")
        h.remove()
        loss = hp["photo_weight"] * res[0] + hp["smooth1_weight"] * res[1] + hp["smooth2_weight"] * res[2]
        loss.backward()
        params = dict(m.named_parameters())
        sd = m.state_dict()
        out = dict(stage_flow=seen[0], stage_back_flow=seen[1], stage_scalars=torch.stack([r.detach() for r in res]), stage_loss=loss)
        out.update({"stage_grad__" + k: _grad_pick(params[k].grad) for k in TRAIN_FPN_GRADS["flownet"]})
        out.update({"stage_buf__" + k: sd["flow_pred." + k] for k in TRAIN_FPN_BUFFERS})
        return {k: v.detach() for k, v in out.items()}

    for run in (run_occ, run_stage):
        r32, r64 = run(torch.float32), run(torch.float64)
        for k, v64 in r64.items():
            v32 = r32[k]
            if not v64.is_floating_point():
                arrs[k] = v64
                continue
            arrs[k] = v64.float()
            if v64.dim() == 4 and v64.shape[1] == 2 and "grad" not in k:  # a flow field: EPE
                noise = epe(v32, v64)
            elif "grad" in k or "buf" in k:  # relative to the quantity's own scale (what the test measures)
                noise = (v32.double() - v64).abs().max() / (v64.abs().max() + 1e-300)
            elif v64.dim() == 0:
                noise = (v32.double() - v64).abs() / v64.abs().clamp_min(1.0)
            elif v64.dim() == 1:  # the scalar vector of the stage step
                noise = ((v32.double() - v64).abs() / v64.abs().clamp_min(1.0)).max()
            else:
                noise = (v32.double() - v64).abs().mean()
            arrs[k + "__noise"] = noise.double()
    def run_fpn(dt):
        # (c) the encoder alone, TRAIN mode, both images through the same module one after the other as the nets call it
        # (flow_occ_net.py:133-134): pyramids, a loss that couples the two, every BatchNorm gradient + conv blocks, running statistics
        from models.networks.feature_pyramid_net import FeaturePyramidNet
        fpn = FeaturePyramidNet()
        seeded_fill(fpn, 11)
        fpn = fpn.to(dt).train()
        p1 = fpn(x[:, :3].to(dt))
        p2 = fpn(x[:, 3:].to(dt))
        loss = sum((a * a).mean() + (a * b).mean() for a, b in zip(p1, p2))
        loss.backward()
        out = dict(fpn_loss=loss)
        for lvl, (a, b) in zip((6, 5, 4, 3, 2), zip(p1, p2)):
            out[f"fpn_p1_l{lvl}"], out[f"fpn_p2_l{lvl}"] = a[:, :16], b[:, :16]
        for k, p in fpn.named_parameters():
            if k.endswith("deconv.bias"):
                continue  # a bias in front of a batch-statistics BatchNorm: its gradient is analytically zero (rounding noise only)
            if p.dim() == 1 or k in ("layer1.double_conv.0.weight", "layer4.double_conv.3.weight", "pyr_top.0.weight", "upsample3.deconv.weight"):
                out["fpn_grad__" + k] = _grad_pick(p.grad)
        for k, v in fpn.state_dict().items():
            if "running" in k or "num_batches" in k:
                out["fpn_buf__" + k] = v
        return {k: v.detach() for k, v in out.items()}

    r32, r64 = run_fpn(torch.float32), run_fpn(torch.float64)
    worst = 0.0
    for k, v64 in r64.items():
        if not v64.is_floating_point():
            arrs[k] = v64
            continue
        arrs[k] = v64.float()
        n = float((r32[k].double() - v64).abs().max() / (v64.abs().max() + 1e-300))
        if os.environ.get("GOLD_VERBOSE"):
            print(f"    {k:60s} noise {n:.3e}  scale {float(v64.abs().max()):.3e}")
        worst = max(worst, n)
    arrs["fpn__noise"] = torch.tensor(worst, dtype=torch.float64)  # one figure: the worst relative deviation fp32 vs fp64 over all entries
    for k in sorted(arrs):
        if k.endswith("__noise"):
            print(f"  {k:90s} {float(arrs[k]):.3e}")
    save("train_fpn", **arrs)

# ------------------------------------------- 14. two-stage trainer with ground-truth flow: TwoStageModelGC, 'simple' + 'pixel-wise'
TWO_STAGE_GRADS = ("occ_pred.down1.conv1.weight", "occ_pred.down5.bn3.weight", "occ_pred.up5.conv2.weight", "occ_pred.predict_occ0.2.0.weight",
                   "occ_pred.predict_occ0.2.0.bias", "occ_pred.predict_occ5.0.0.weight", "inpainting.down1.conv1.weight",
                   "inpainting.up6.conv3.weight")


def gold_two_stage():
    """One TRAINING step of TwoStageModelGC (models/model.py:1199-1256) with inpainting_stage='simple', loss_type='pixel-wise', trained from
    scratch: B = 2 at 64 x 96, both networks in train mode (batch-statistics BatchNorm), golden_util.seeded_fill weights.  Stored: the six
    values general_step returns, the total of training_step's expression (:1256; that method itself cannot run, it unpacks six values
    after testing `len(losses) == 3`), the soft occlusion map, the completed image, parameter gradients of both networks.  Values are
    the reference in fp64; `<name>__noise` = deviation of its own fp32 run."""
    arrs = {}
    x = torch.from_numpy(smooth_field((2, 6, 64, 96), seed=111, amp=1.0)).float().clamp(-1, 1)
    gt_flow = torch.randn((2, 2, 64, 96), generator=g(112)) * 2
    gt_occ = (torch.rand((2, 1, 64, 96), generator=g(113)) < 0.3).float()
    arrs.update(x_absmean=x.abs().mean(), gt_flow_absmean=gt_flow.abs().mean(), gt_occ_mean=gt_occ.mean())
    hp = dict(learning_rate=1e-4, inpainting_stage="simple", loss_type="pixel-wise", using_pretrained_inpainting=False, img_size=(64, 96),
              photo_weight=1.0, reconst_weight=1.0, smooth1_weight=0.5, smooth2_weight=0.0, pixelwise_weight=2.0, batch_size=2)
    manifest = {}

    def run(dt):
        m = ref_model.TwoStageModelGC(dict(hp))
        seeded_fill(m, 13)
        manifest["simpleocclusionnet"] = [(k, tuple(v.shape)) for k, v in m.occ_pred.state_dict().items()]
        m = m.to(dt).train()
        seen = {}
        h1 = m.occ_pred.register_forward_hook(lambda mod, inp, out: seen.__setitem__("soft", out))
        h2 = m.inpainting.register_forward_hook(lambda mod, inp, out: seen.__setitem__("completed", out))
        res = m.general_step((x.to(dt), gt_flow.to(dt), gt_occ.to(dt)), 0, "train")
        h1.remove(); h2.remove()
        photo, reconst, smooth, pixelwise, bce, photo_occ = res
        loss = hp["photo_weight"] * photo + hp["reconst_weight"] * reconst + hp["smooth1_weight"] * smooth + hp["pixelwise_weight"] * pixelwise
        loss.backward()
        params = dict(m.named_parameters())
        out = dict(ts_scalars=torch.stack([r.detach() for r in res]), ts_loss=loss, ts_soft=seen["soft"], ts_completed=seen["completed"])
        out.update({"ts_grad__" + k: _grad_pick(params[k].grad) for k in TWO_STAGE_GRADS})
        sd = m.state_dict()
        out["ts_buf__occ_pred.down1.bn1.running_mean"] = sd["occ_pred.down1.bn1.running_mean"]
        out["ts_buf__occ_pred.down1.bn1.running_var"] = sd["occ_pred.down1.bn1.running_var"]
        return {k: v.detach() for k, v in out.items()}

    r32, r64 = run(torch.float32), run(torch.float64)
    for k, v64 in r64.items():
        v32 = r32[k]
        arrs[k] = v64.float()
        if "grad" in k or "buf" in k:
            noise = (v32.double() - v64).abs().max() / (v64.abs().max() + 1e-300)
        elif v64.dim() == 0:
            noise = (v32.double() - v64).abs() / v64.abs().clamp_min(1.0)
        elif v64.dim() == 1:
            noise = ((v32.double() - v64).abs() / v64.abs().clamp_min(1.0)).max()
        else:
            noise = (v32.double() - v64).abs().max()
        arrs[k + "__noise"] = noise.double()
    for k in sorted(arrs):
        if k.endswith("__noise"):
            print(f"  {k:70s} {float(arrs[k]):.3e}")
    print("  scalars (photo, reconst, smooth, pixelwise, bce, photo_occ):", [round(float(v), 6) for v in arrs["ts_scalars"]], float(arrs["ts_loss"]))
    save("two_stage", **arrs)
    import json
    with open(os.path.join(HERE, "state_dict_manifest_two_stage.json"), "w") as f:
        json.dump(manifest, f)

# ------------------------------------------- 15. gated-convolution generator (second half): InpaintSANet + the 'gated' two-stage step
GATED_GRADS = ("occ_pred.down1.conv1.weight", "occ_pred.predict_occ0.2.0.weight", "inpainting.coarse_net.0.conv2d.conv2.weight",
               "inpainting.coarse_net.6.mask_conv2d.conv2.weight", "inpainting.coarse_net.12.conv2d.mask_conv2d.conv3.bias",
               "inpainting.refine_attn.gamma", "inpainting.refine_attn.query_conv.weight", "inpainting.refine_upsample_net.6.batch_norm.weight")


def gold_gated():
    """(a) InpaintSANet(img_size=(64, 96)) alone, golden_util.seeded_fill weights (gamma of the attention block != 0), evaluation and training
    mode: coarse and refined completions of a 2 x 64 x 96 batch; state_dict manifest.  (b) one TRAINING step of TwoStageModelGC with
    inpainting_stage='gated', loss_type='pixel-wise', from scratch (models/model.py:1199-1256): the six step values, the total, eight
    gradients.  fp64 values + `__noise` (the reference's own fp32 deviation) as in gold_two_stage."""
    from models.networks.gated_conv_inpainting_net import InpaintSANet
    import json
    arrs = {}
    imgs = torch.from_numpy(smooth_field((2, 3, 64, 96), seed=121, amp=1.0)).float().clamp(-1, 1)
    masks = (torch.from_numpy(smooth_field((2, 1, 64, 96), seed=122, amp=1.0)) > 0.3).float()
    arrs.update(imgs_absmean=imgs.abs().mean(), masks_mean=masks.mean())

    def run_net(dt, train):
        net = InpaintSANet(img_size=(64, 96))
        seeded_fill(net, 17)
        net = net.to(dt)
        net.train(train)
        with torch.no_grad():
            c, r = net(imgs.to(dt), masks.to(dt))
        tag = "train" if train else "eval"
        return {f"net_{tag}_coarse": c, f"net_{tag}_refined": r}, [(k, tuple(v.shape)) for k, v in net.state_dict().items()]

    manifest = None
    for train in (False, True):
        (r32, manifest), (r64, _) = run_net(torch.float32, train), run_net(torch.float64, train)
        for k in r64:
            arrs[k] = r64[k].float()
            arrs[k + "__noise"] = (r32[k].double() - r64[k]).abs().max().double()

    x = torch.from_numpy(smooth_field((2, 6, 64, 96), seed=123, amp=1.0)).float().clamp(-1, 1)
    gt_flow = torch.randn((2, 2, 64, 96), generator=g(124)) * 2
    gt_occ = (torch.rand((2, 1, 64, 96), generator=g(125)) < 0.3).float()
    arrs.update(x_absmean=x.abs().mean(), gt_occ_mean=gt_occ.mean())
    hp = dict(learning_rate=1e-4, inpainting_stage="gated", loss_type="pixel-wise", using_pretrained_inpainting=False, img_size=(64, 96),
              photo_weight=4.0, reconst_weight=1.0, smooth1_weight=0.5, smooth2_weight=0.0, pixelwise_weight=2.0, batch_size=2)

    def run_step(dt):
        m = ref_model.TwoStageModelGC(dict(hp))
        seeded_fill(m, 19)
        m = m.to(dt).train()
        res = m.general_step((x.to(dt), gt_flow.to(dt), gt_occ.to(dt)), 0, "train")
        photo, reconst, smooth, pixelwise, bce, photo_occ = res
        loss = hp["photo_weight"] * photo + hp["reconst_weight"] * reconst + hp["smooth1_weight"] * smooth + hp["pixelwise_weight"] * pixelwise
        loss.backward()
        params = dict(m.named_parameters())
        out = dict(gs_scalars=torch.stack([r.detach() for r in res]), gs_loss=loss)
        out.update({"gs_grad__" + k: _grad_pick(params[k].grad) for k in GATED_GRADS})
        return {k: v.detach() for k, v in out.items()}

    r32, r64 = run_step(torch.float32), run_step(torch.float64)
    for k, v64 in r64.items():
        v32 = r32[k]
        arrs[k] = v64.float()
        if "grad" in k:
            noise = (v32.double() - v64).abs().max() / (v64.abs().max() + 1e-300)
        elif v64.dim() == 0:
            noise = (v32.double() - v64).abs() / v64.abs().clamp_min(1.0)
        else:
            noise = ((v32.double() - v64).abs() / v64.abs().clamp_min(1.0)).max()
        arrs[k + "__noise"] = noise.double()
    for k in sorted(arrs):
        if k.endswith("__noise"):
            print(f"  {k:70s} {float(arrs[k]):.3e}")
    print("  scalars:", [round(float(v), 6) for v in arrs["gs_scalars"]], float(arrs["gs_loss"]))
    save("gated", **arrs)
    with open(os.path.join(HERE, "state_dict_manifest_gated.json"), "w") as f:
        json.dump({"inpaintsanet": manifest}, f)

# ------------------------------------------- 16. the third step variant of FlowStageModel: photometric term masked by the GROUND-TRUTH occlusion
def gold_stage_gtocc():
    """FlowStageModel('pwc', with_occ=True, occ_aware=False): general_step_occ (models/model.py:343-364) on the inputs of the other two
    stage fixtures (nets.npz: stage_x / stage_gt_flow / stage_gt_occ, regenerated here from the same seeds)"""
    hp = dict(learning_rate=1e-5, model="pwc", displacement=4, photo_weight=4.0, smooth1_weight=0.5, smooth2_weight=0.0, with_occ=True,
              occ_aware=False)
    xs = torch.from_numpy(smooth_field((2, 6, 64, 128), seed=66, amp=1.0)).float().clamp(-1, 1)
    gt_flow2 = torch.randn((2, 2, 64, 128), generator=g(67)) * 2
    gt_occ2 = (torch.rand((2, 1, 64, 128), generator=g(68)) < 0.3).float()
    m = ref_model.FlowStageModel(dict(hp))
    rngfree_fill(m)
    m.train()
    m.logger = None
    m.global_step = 1
    res = m.general_step_occ((xs, gt_flow2, gt_occ2), 0, "train")
    loss = hp["photo_weight"] * res[0] + hp["smooth1_weight"] * res[1] + hp["smooth2_weight"] * res[2]
    loss.backward()
    save("stage_gtocc", scalars=torch.stack([r.detach() for r in res]), loss=loss,
         grad_predict_flow2=m.flow_pred.predict_flow2.weight.grad[:, :16], grad_conv1a=m.flow_pred.conv1a[0].weight.grad,
         x_absmean=xs.abs().mean(), occ_mean=gt_occ2.mean())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "stage_gtocc":
        gold_stage_gtocc()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "norm":
        gold_norm()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "gated":
        gold_gated()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "two_stage":
        gold_two_stage()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "train_fpn":
        gold_train_fpn()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize":
        gold_fullsize()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "inpainting":
        gold_inpainting()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "psnr":
        gold_psnr()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ssim":
        gold_ssim()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "epe":
        gold_epe()
        sys.exit(0)
    gold_corr()
    gold_norm()
    gold_warp()
    gold_losses()
    gold_range()
    gold_epe()
    gold_ssim()
    gold_psnr()
    gold_nets()
    gold_fullsize()
    gold_inpainting()
    gold_train_fpn()
    gold_two_stage()
    gold_gated()
    gold_stage_gtocc()
