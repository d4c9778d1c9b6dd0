"""GPU tests of the BASELINE configs at their OWN sizes ("configs not exercised"):
  configs[1]  FlowOccModel('flowoccnet') 1 x 6 x 436 x 1024 and FlowNetCV 1 x 6 x 448 x 1024, fp32, against goldens produced by the
              reference at those sizes (tests/golden/fullsize.npz: every 4th pixel of the outputs; inputs from smooth_field, weights
              from rngfree_fill) -- north_star bar: flow EPE and occlusion L1 < 1e-4;
  configs[2]  FlowStageModel('pwc', occ_aware) training steps at 8 x 384 x 512 (bf16, channels_last): finite decreasing loss, range-map
              mass conservation on the model's own backward flow;
  configs[3]  the inpainting head at the KITTI render size 4 x 320 x 1216 and the op chain at 375 x 1242 against the reference."""
import os

import numpy as np
import pytest
import torch

from golden_util import rngfree_fill, smooth_field
from oracle import ocflow_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy


@pytest.fixture(autouse=True)
def _no_tf32():
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    yield


def epe(a, b):
    return float(torch.sqrt(((a.float().cpu() - T(b)) ** 2).sum(1)).mean())


def test_config1_flowoccnet_at_436x1024(golden):
    from ocflow_amd.models.flow_occ_model import FlowOccModel
    g = golden("fullsize")
    x = T(smooth_field((1, 6, 436, 1024), seed=71, amp=1.0)).float().clamp(-1, 1).to(DEV)
    for fmt in (torch.contiguous_format, torch.channels_last):
        m = rngfree_fill(FlowOccModel({"learning_rate": 1e-3, "model": "flowoccnet"})).to(DEV).eval().to(memory_format=fmt)
        with torch.no_grad():
            fl, oc = m(x.contiguous(memory_format=fmt))
        assert fl.shape == (1, 2, 436, 1024) and oc.shape == (1, 1, 436, 1024)
        assert epe(fl[..., ::4, ::4], g["flowoccnet_436x1024_flow_sub4"]) < 1e-4
        assert float((oc[..., ::4, ::4].cpu() - T(g["flowoccnet_436x1024_occ_sub4"])).abs().mean()) < 1e-4
        assert abs(float(fl.abs().mean()) - float(g["flowoccnet_436x1024_flow_absmean"])) < 1e-4
        assert abs(float(oc.mean()) - float(g["flowoccnet_436x1024_occ_mean"])) < 1e-4


# what the benchmark's configuration (bf16 autocast, channels_last, this repo's convolution kernels) costs against the reference's fp32
# outputs at the benchmark size; measured on MI355X (round 5): see the numbers printed by the test and README.  Bars = 2 x measured.
BF16_EPE_BAR, BF16_OCC_BAR = 3.1e-4, 2.6e-3  # measured 1.53e-4 px / 1.28e-3 (gpurun_out/r5d_new_tests.txt)


def test_config1_flowoccnet_bf16_native_convolutions_gap_to_the_reference(golden):
    """every <= 1e-4 golden test runs fp32, where each layer goes to the library.  This one runs the configuration
    the benchmark times -- FlowOccModel('flowoccnet'), eval, bf16 autocast, channels_last, native convolution / concatenation / BatchNorm
    kernels -- at 1 x 436 x 1024 against the reference's fp32 golden (tests/golden/fullsize.npz) and holds the gap (bf16 cannot meet 1e-4:
    SURVEY 9 measured 9e-3 px from rounding the correlation's I/O alone; parity is claimed in fp32, this is the throughput mode's error)."""
    from ocflow_amd import ops
    from ocflow_amd.models.flow_occ_model import FlowOccModel
    g = golden("fullsize")
    x = T(smooth_field((1, 6, 436, 1024), seed=71, amp=1.0)).float().clamp(-1, 1).to(DEV).contiguous(memory_format=torch.channels_last)
    m = rngfree_fill(FlowOccModel({"learning_rate": 1e-3, "model": "flowoccnet"})).to(DEV).eval().to(memory_format=torch.channels_last)
    calls = []

    class Spy:
        def launch(self, name, meta, fn):
            calls.append(name)
            return fn()

    ops.set_kernel_timer(Spy())
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            fl, oc = m(x)
    finally:
        ops.set_kernel_timer(None)
    assert calls.count("conv3x3_igemm") >= 40 and "corr_fwd" in calls and "warp_fwd" in calls, sorted(set(calls))
    e = epe(fl[..., ::4, ::4], g["flowoccnet_436x1024_flow_sub4"])
    o = float((oc[..., ::4, ::4].float().cpu() - T(g["flowoccnet_436x1024_occ_sub4"])).abs().mean())
    print(f"bf16 + native convolutions vs the reference's fp32 golden at 1x436x1024: flow EPE {e:.3e} px, occlusion L1 {o:.3e} "
          f"({calls.count('conv3x3_igemm')} implicit-GEMM launches)")
    assert e < BF16_EPE_BAR and o < BF16_OCC_BAR, (e, o)


def test_config1_flownetcv_at_448x1024(golden):
    from ocflow_amd.models.flow_model import FlowModel
    g = golden("fullsize")
    x = T(smooth_field((1, 6, 448, 1024), seed=72, amp=1.0)).float().clamp(-1, 1).to(DEV)
    m = rngfree_fill(FlowModel({"learning_rate": 1e-3, "model": "pwc", "displacement": 4})).to(DEV)
    with torch.no_grad():
        full, quarter = m.flow_pred(x)
    assert epe(full[..., ::4, ::4], g["flownetcv_448x1024_full_sub4"]) < 1e-4
    np.testing.assert_allclose(quarter.cpu().numpy(), g["flownetcv_448x1024_quarter"], rtol=1e-4, atol=1e-4)


def test_config2_pwc_unsupervised_steps_at_8x384x512():
    """BASELINE configs[2]: unsupervised_config.yml weights with network_type 'flow', model 'pwc', batch 8 at 384 x 512"""
    from ocflow_amd import ops
    from ocflow_amd.models.model import FlowStageModel
    from ocflow_amd.runtime import DistEnv, TrainStepper, synthetic_batch
    torch.manual_seed(3)
    hp = dict(learning_rate=1e-4, model="pwc", displacement=4, photo_weight=4.0, smooth1_weight=0.5, smooth2_weight=0.0, with_occ=False,
              occ_aware=True)
    env = DistEnv(device=torch.device(DEV))
    st = TrainStepper(FlowStageModel(hp), env, amp_dtype=torch.bfloat16, channels_last=True, fused_optimizer=True)
    batch = st.prepare(synthetic_batch(8, 384, 512, DEV, seed=5))
    losses = [float(st.step(batch, i)) for i in range(12)]
    assert all(np.isfinite(losses)), losses
    assert min(losses[6:]) < losses[0], losses  # the photometric + smoothness objective goes down on a fixed batch
    gmax = max(float(p.grad.abs().max()) for p in st.raw.parameters() if p.grad is not None)
    assert np.isfinite(gmax) and gmax < 1e4
    # range-map mass of the model's own backward flow: sum(range) == number of in-bounds bilinear weight (oracle), occ in [0, 1]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        back = st.raw(torch.cat((batch[0][:, 3:6], batch[0][:, 0:3]), 1))[0].float()
    rm = ops.compute_range_map(back)
    ref = O.compute_range_map(back.cpu())
    assert abs(float(rm.double().sum()) - float(ref.double().sum())) <= 1e-5 * float(ref.double().sum())
    occ = ops.occlusion_from_flow(back)
    assert float(occ.min()) >= 0.0 and float(occ.max()) <= 1.0


def test_config3_inpainting_head_at_kitti_size(golden):
    """4 x 320 x 1216 (375 x 1242 through the reference's crop rule): warp -> mask -> InpaintingNet -> masked L1, eval mode, fp32"""
    from ocflow_amd.data.flow_io import render_size
    from ocflow_amd.models.inpainting_model import InpaintingModel
    g = golden("inpainting")
    assert render_size(375, 1242) == (320, 1216)
    imgs = T(smooth_field((4, 6, 320, 1216), seed=84, amp=1.0)).float().clamp(-1, 1).to(DEV)
    flow = T(smooth_field((4, 2, 320, 1216), seed=85, amp=6.0)).float().to(DEV)
    occ = (T(smooth_field((4, 1, 320, 1216), seed=86, amp=1.0)) > 0.35).float().to(DEV)
    m = InpaintingModel({"learning_rate": 1e-4}).to(DEV)
    rngfree_fill(m.model)
    m.eval()
    with torch.no_grad():
        warped = m.warp(imgs[:, 3:6], flow)
        completed = m(warped * (1 - occ), occ)
        loss = m.general_step((imgs, flow, occ), 0, "val")
    # (at W = 1216 an fp32 sample coordinate resolves ~1e-4 px: the reference's normalised grid and the kernel's closed form differ there)
    np.testing.assert_allclose(warped[..., ::8, ::8].cpu().numpy(), g["k_warped_sub8"], rtol=1e-4, atol=1e-4)
    assert float((completed[..., ::8, ::8].cpu() - T(g["k_completed_sub8"])).abs().mean()) < 1e-4
    assert abs(float(loss) - float(g["k_loss"])) < 1e-4 * max(1.0, abs(float(g["k_loss"])))
    # one bf16 training step of the head runs and produces finite gradients
    m.train()
    opt = m.configure_optimizers()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        l2 = m.training_step((imgs, flow, occ), 0)
    l2.backward()
    opt.step()
    assert torch.isfinite(l2) and all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_config3_op_chain_at_375x1242(golden):
    """op level at the raw KITTI size: AC-False warp of img2, (1 - occ) mask, masked L1 and its gradient vs the reference"""
    from ocflow_amd import ops
    g = golden("inpainting")
    img2 = T(smooth_field((4, 3, 375, 1242), seed=87, amp=1.0)).float().clamp(-1, 1).to(DEV)
    img1 = T(smooth_field((4, 3, 375, 1242), seed=88, amp=1.0)).float().clamp(-1, 1).to(DEV)
    flow = T(smooth_field((4, 2, 375, 1242), seed=89, amp=8.0)).float().to(DEV)
    occ = (T(smooth_field((4, 1, 375, 1242), seed=90, amp=1.0)) > 0.3).float().to(DEV)
    pred = (ops.warp(img2, flow, align_corners=False) * (1 - occ)).requires_grad_()
    np.testing.assert_allclose(pred.detach()[..., ::8, ::8].cpu().numpy(), g["o_masked_sub8"], rtol=1e-4, atol=1e-4)
    assert abs(float(ops.masked_l1(pred, img1, occ)) - float(g["o_loss"])) < 2e-5 * abs(float(g["o_loss"]))
    loss_half = ops.masked_l1(pred * 0.5, img1, occ)
    assert abs(float(loss_half) - float(g["o_loss_half"])) < 2e-5 * abs(float(g["o_loss_half"]))
    (gp,) = torch.autograd.grad(loss_half, pred)
    np.testing.assert_allclose(gp[..., ::8, ::8].cpu().numpy(), g["o_grad_half_sub8"], rtol=1e-5, atol=1e-10)
    # strided views (the [B,6,H,W] batch) and a channels_last prediction go through the stride ABI unchanged
    both = torch.cat((img1, img2), 1)
    assert abs(float(ops.masked_l1(pred.detach().contiguous(memory_format=torch.channels_last), both[:, 0:3], occ)) - float(g["o_loss"])) < 2e-5
    with pytest.raises(RuntimeError, match="no CPU fallback|HIP"):
        ops.masked_l1(pred.detach().cpu(), img1.cpu(), occ.cpu())


def test_launcher_trains_from_a_sintel_tree(tmp_path):
    """`python -m ocflow_amd.train` with `root` set: DatasetModule in decode-only mode (PNG / .flo / occlusion PNG on the host), crop +
    resize + normalise + flow rescale + binarise on the GPU, the occlusion-aware unsupervised step replayed as a hipGraph over changing
    batches (the callers and data formats either side of the hot path)"""
    import yaml
    from PIL import Image
    from ocflow_amd import train
    from ocflow_amd.data import flow_io
    rs = np.random.RandomState(3)
    for scene, n in (("alley_1", 6), ("cave_2", 5)):
        for sub in ("clean", "flow", "occlusions"):
            (tmp_path / sub / scene).mkdir(parents=True)
        base = smooth_field((1, 3, 140, 212), seed=len(scene), amp=1.0)[0].transpose(1, 2, 0)
        for i in range(1, n + 1):
            img = np.clip((np.roll(base, 2 * i, axis=1) * 0.5 + 0.5) * 255, 0, 255).astype(np.uint8)
            Image.fromarray(img).save(tmp_path / "clean" / scene / f"frame_{i:04d}.png")
            if i < n:
                fl = np.zeros((140, 212, 2), np.float32)
                fl[..., 0] = 2.0
                flow_io.save_flow(str(tmp_path / "flow" / scene / f"frame_{i:04d}.flo"), fl)
                Image.fromarray(((rs.rand(140, 212) < 0.2) * 255).astype(np.uint8)).save(tmp_path / "occlusions" / scene / f"frame_{i:04d}.png")
    cfg = dict(batch_size=2, dataset_name="MpiSintelCleanFlowOcc", epochs=1, image_size=(64, 128), learning_rate=1e-4, log_every_n_steps=4,
               model="flownet", network_type="flow", org=False, photo_weight=4.0, smooth1_weight=0.5, smooth2_weight=0.0, displacement=4,
               with_occ=False, occ_aware=True, root=str(tmp_path), num_workers=0)
    path = tmp_path / "cfg.yml"
    path.write_text(yaml.dump(cfg))
    hist = train.main([str(path), "--steps", "13", "--out", str(tmp_path / "out")])
    assert len(hist) >= 3 and all(np.isfinite(h["train_loss"]) for h in hist)
    sd = torch.load(tmp_path / "out" / "state_dict.pt")
    assert any(k.startswith("flow_pred.feature_pyramid_network.") for k in sd)


def test_launcher_trains_the_two_stage_model(tmp_path):
    """`python -m ocflow_amd.train` with `network_type: twostage`, `model: with_gt_flow` (test_unsupervised.py:72-98 -> TwoStageModelGC,
    inpainting_stage 'simple', loss_type 'pixel-wise'): the straight-through two-stage step as a replayed hipGraph on synthetic
    Sintel-shaped batches; the loss goes down over 40 steps of one batch"""
    import yaml
    from ocflow_amd import train
    cfg = dict(batch_size=2, dataset_name="MpiSintelCleanFlowOcc", epochs=1, image_size=(64, 128), learning_rate=1e-3, log_every_n_steps=13,
               model="with_gt_flow", network_type="twostage", org=False, photo_weight=1.0, reconst_weight=1.0, smooth1_weight=0.5,
               smooth2_weight=0.0, pixelwise_weight=1.0, inpainting_root=None, using_pretrained_inpainting=False, loss_type="pixel-wise",
               inpainting_stage="simple", n_display_images=1, log_image_every_epoch=10, root="/nonexistent")
    path = tmp_path / "cfg.yml"
    path.write_text(yaml.dump(cfg))
    hist = train.main([str(path), "--steps", "40", "--fp32", "--out", str(tmp_path / "out")])
    assert len(hist) >= 3 and all(np.isfinite(h["train_loss"]) for h in hist)
    assert hist[-1]["train_loss"] < hist[0]["train_loss"], hist
    sd = torch.load(tmp_path / "out" / "state_dict.pt")
    assert any(k.startswith("occ_pred.predict_occ0.") for k in sd) and any(k.startswith("inpainting.up6.") for k in sd)


@pytest.mark.parametrize("native", ["0", "1"])
def test_fpn_net_step_at_a_shape_outside_the_find_db(monkeypatch, native):
    """the decoder-input alignment at 4 x 320 x 1216 -- a shape the shipped MIOpen find-db has no entry for -- in both configurations:
    native = "0" (the LIBRARY's convolutions consume the concatenations: DESIGN 11.3): they stay unaligned, `_padded_shape_is_known` says no
    for every decoder input of the step; native = "1" (the repo's own kernels consume them, round 4): always aligned, by
    ops.concat_channels, and the library's solver choice is never asked.  Either way the occlusion-aware step of FlowStageModel('flownet')
    trains: finite, decreasing loss on a fixed batch, finite gradients"""
    monkeypatch.setenv("OCFLOW_NATIVE_CONV", native)
    from ocflow_amd.models.model import FlowStageModel
    from ocflow_amd.models.networks import _epilogue
    from ocflow_amd.runtime import DistEnv, TrainStepper, synthetic_batch, use_repo_miopen_db
    use_repo_miopen_db()
    torch.manual_seed(4)
    asked = []
    real = _epilogue._padded_shape_is_known

    def spy(n, c, h, w):
        r = real(n, c, h, w)
        asked.append(((n, c, h, w), r))
        return r

    _epilogue._padded_shape_is_known = spy
    try:
        hp = dict(learning_rate=1e-4, model="flownet", displacement=4, photo_weight=4.0, smooth1_weight=0.5, smooth2_weight=0.0, with_occ=False,
                  occ_aware=True)
        st = TrainStepper(FlowStageModel(hp), DistEnv(device=torch.device(DEV)), amp_dtype=torch.bfloat16, channels_last=True, fused_optimizer=True)
        batch = st.prepare(synthetic_batch(4, 320, 1216, DEV, seed=6))
        losses = [float(st.step(batch, i)) for i in range(8)]
    finally:
        _epilogue._padded_shape_is_known = real
    assert all(np.isfinite(losses)) and min(losses[4:]) < losses[0], losses
    if native == "0":
        assert asked and not any(r for _, r in asked), [a for a in asked if a[1]][:3]
    else:
        assert not asked, asked[:3]
    gmax = max(float(p.grad.abs().max()) for p in st.raw.parameters() if p.grad is not None)
    assert np.isfinite(gmax) and gmax < 1e4


def test_launcher_runs_the_shipped_two_stage_configuration(tmp_path):
    """config/unsupervised_twostage.yml = the reference's default run (twostage / with_gt_flow / gated / pixel-wise / FlyingChairs2 at
    96 x 128) with the completion network trained from scratch: a synthetic FlyingChairs2 directory, decode on the host, sample rules on
    the GPU, the step (occlusion U-Net + gated generator with attention + HIP loss kernels) replayed as a hipGraph"""
    from PIL import Image
    from ocflow_amd import train
    from ocflow_amd.data import flow_io
    rs = np.random.RandomState(7)
    root = tmp_path / "chairs2"
    root.mkdir()
    base = smooth_field((1, 3, 140, 200), seed=9, amp=1.0)[0].transpose(1, 2, 0)
    for i in range(12):
        for j in (0, 1):
            img = np.clip((np.roll(base, 3 * i + 2 * j, axis=1) * 0.5 + 0.5) * 255, 0, 255).astype(np.uint8)
            Image.fromarray(img).save(root / f"{i:07d}-img_{j}.png")
        fl = np.zeros((140, 200, 2), np.float32)
        fl[..., 0] = 2.0
        flow_io.save_flow(str(root / f"{i:07d}-flow_01.flo"), fl)
        Image.fromarray(((rs.rand(140, 200) < 0.2) * 255).astype(np.uint8)).save(root / f"{i:07d}-occ_01.png")
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config", "unsupervised_twostage.yml")
    # In the pytest process (round 4 ran it as `python -m ocflow_amd.train` in a process of its own after an unexplained abort in one test
    # order; round 5: the order passes with and without output capture, and the runtime's process-global state now has owners --
    # DESIGN.md 13.2).  OCFLOW_TEST_LAUNCHER_SUBPROCESS=1 runs the launcher as a user starts it.
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    argv = [cfg, "--steps", "13", "--override", f"root={root}", "batch_size=4", "num_workers=0", "log_every_n_steps=4", "--out", str(tmp_path / "out")]
    if os.environ.get("OCFLOW_TEST_LAUNCHER_SUBPROCESS") != "1":
        train.main(argv)
    else:
        run = subprocess.run([sys.executable, "-m", "ocflow_amd.train"] + argv, cwd=repo, capture_output=True, text=True, timeout=600)
        assert run.returncode == 0, run.stderr[-3000:]
    hist = json.load(open(tmp_path / "out" / "metrics.json"))["history"]
    assert len(hist) >= 3 and all(np.isfinite(h["train_loss"]) for h in hist)
    sd = torch.load(tmp_path / "out" / "state_dict.pt")
    assert any(k.startswith("inpainting.refine_attn.") for k in sd) and any(k.startswith("occ_pred.") for k in sd)
