"""Host logic on CPU: the model mirror (ocflow_amd.models.*) reproduces the reference's outputs, state_dict keys and
step scalars when its hot ops are computed by the oracle (monkeypatched in -- the product has no CPU path).
Goldens: tests/golden/nets.npz + state_dict_manifest.json, produced by the reference with RNG-free weights."""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import rngfree_fill
from oracle_backend import patch_ops

T = torch.from_numpy
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture
def cpu_ops(monkeypatch):
    patch_ops(monkeypatch)


def close(a, b, tol):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol)


def epe(a, b):
    return float(torch.sqrt(((a - T(b)) ** 2).sum(1)).mean())


def test_config0_flow_model_cpu_no_patch(golden):
    """BASELINE configs[0]: models.flow_model forward on 2 random 64x64x3 frames, CPU fp32 -- default model 'simple'
    has no hot-path op, so it runs on CPU without any HIP kernel."""
    from ocflow_amd.models.flow_model import FlowModel
    g = golden("nets")
    m = rngfree_fill(FlowModel({"learning_rate": 1e-3}))
    m.train()
    with torch.no_grad():
        out = m(T(g["x64b2"]))
    assert epe(out, g["flowmodel_simple_train"]) < 1e-4
    rngfree_fill(m).eval()  # the train-mode pass above moved the BatchNorm running statistics
    with torch.no_grad():
        assert epe(m(T(g["x64"])), g["flowmodel_simple_eval"]) < 1e-4


def test_ops_refuse_cpu():
    from ocflow_amd import ops
    f = torch.zeros(1, 4, 8, 8)
    for call in (lambda: ops.compute_cost_volume(f, f), lambda: ops.warp(f, f[:, :2]), lambda: ops.compute_range_map(f[:, :2]),
                 lambda: ops.photometric_error(f, f), lambda: ops.smoothness_losses(f, f[:, :2])):
        with pytest.raises(RuntimeError, match="no CPU fallback|HIP"):
            call()


def test_unknown_model_raises():
    from ocflow_amd.models.flow_model import FlowModel
    from ocflow_amd.models.flow_occ_model import FlowOccModel
    with pytest.raises(ValueError, match="Unsupported model"):
        FlowModel({"learning_rate": 1e-3, "model": "nope"})
    with pytest.raises(ValueError, match="Unsupported model"):
        FlowOccModel({"learning_rate": 1e-3, "model": "nope"})
    with pytest.raises(KeyError):
        FlowModel({})  # learning_rate is required (flow_model.py:21)


@pytest.mark.parametrize("key,cls,name", [("flowmodel_simple", "FlowModel", "simple"), ("flowmodel_pwc", "FlowModel", "pwc"),
                                           ("flowmodel_flownet", "FlowModel", "flownet"),
                                           ("flowoccmodel_flowoccnet", "FlowOccModel", "flowoccnet"),
                                           ("flowoccmodel_pwoc", "FlowOccModel", "pwoc"),
                                           ("flowoccmodel_simple", "FlowOccModel", "simple"),
                                           ("flowstage_pwc", "FlowStageModel", "pwc")])
def test_state_dict_manifest(key, cls, name):
    """checkpoint compatibility: same keys, shapes AND order as the reference's state_dict"""
    from ocflow_amd.models.flow_model import FlowModel
    from ocflow_amd.models.flow_occ_model import FlowOccModel
    from ocflow_amd.models.model import FlowStageModel
    with open(os.path.join(GOLDEN, "state_dict_manifest.json")) as f:
        manifest = json.load(f)[key]
    m = {"FlowModel": FlowModel, "FlowOccModel": FlowOccModel, "FlowStageModel": FlowStageModel}[cls](
        {"learning_rate": 1e-3, "model": name})
    mine = [(k, list(v.shape)) for k, v in m.state_dict().items()]
    assert mine == [(k, list(s)) for k, s in manifest]


def test_flow_model_pwc_and_flownet(cpu_ops, golden):
    from ocflow_amd.models.flow_model import FlowModel
    g = golden("nets")
    m = rngfree_fill(FlowModel({"learning_rate": 1e-3, "model": "pwc", "displacement": 4}))
    with torch.no_grad():
        assert epe(m(T(g["x64"])), g["flowmodel_pwc_train"]) < 1e-4
        full, quarter = m.flow_pred(T(g["x64"]))
    close(full, g["flownetcv_full"], 1e-4)
    close(quarter, g["flownetcv_quarter"], 1e-4)
    m = rngfree_fill(FlowModel({"learning_rate": 1e-3, "model": "flownet"})).eval()
    with torch.no_grad():
        assert epe(m(T(g["x96"])), g["flowmodel_flownet_eval"]) < 1e-4


def test_flow_occ_models(cpu_ops, golden):
    from ocflow_amd.models.flow_occ_model import FlowOccModel
    g = golden("nets")
    m = rngfree_fill(FlowOccModel({"learning_rate": 1e-3, "model": "flowoccnet"})).eval()
    with torch.no_grad():
        fl, oc = m(T(g["x96"]))
    assert epe(fl, g["flowoccmodel_flowoccnet_eval_flow"]) < 1e-4
    assert float((oc - T(g["flowoccmodel_flowoccnet_eval_occ"])).abs().mean()) < 1e-4
    m = rngfree_fill(FlowOccModel({"learning_rate": 1e-3, "model": "pwoc"}))
    with torch.no_grad():
        fl, oc = m(T(g["x64"]))
    assert epe(fl, g["flowoccmodel_pwoc_train_flow"]) < 1e-4
    assert float((oc - T(g["flowoccmodel_pwoc_train_occ"])).abs().mean()) < 1e-4
    m = rngfree_fill(FlowOccModel({"learning_rate": 1e-3}))
    with torch.no_grad():
        fl, oc = m(T(g["x64b2"]))
    assert epe(fl, g["flowoccmodel_simple_train_flow"]) < 1e-4
    close(oc, g["flowoccmodel_simple_train_occ"], 1e-4)


def test_supervised_step_with_grads(cpu_ops, golden):
    from ocflow_amd.models.flow_occ_model import FlowOccModel
    g = golden("nets")
    m = rngfree_fill(FlowOccModel({"learning_rate": 1e-3, "model": "pwoc"}))
    fl_loss, oc_loss = m.general_step((T(g["x64"]), T(g["pwoc_step_gt_flow"]), T(g["pwoc_step_gt_occ"])), 0, "train")
    close(fl_loss, g["pwoc_step_flow_loss"], 1e-5)
    close(oc_loss, g["pwoc_step_occ_loss"], 1e-5)
    (fl_loss + oc_loss).backward()
    close(m.model.conv2_0[0].weight.grad[:4, :8], g["pwoc_step_grad_conv2_0"], 2e-5)
    close(m.model.conv1a[0].weight.grad, g["pwoc_step_grad_conv1a"], 2e-4)
    opt = m.configure_optimizers()
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["lr"] == 1e-3


@pytest.mark.parametrize("occ_aware,tag", [(True, "occaware"), (False, "plain")])
def test_unsupervised_stage_step(cpu_ops, golden, occ_aware, tag):
    from ocflow_amd.models.model import FlowStageModel
    g = golden("nets")
    hp = dict(learning_rate=1e-5, model="pwc", displacement=4, photo_weight=4.0, smooth1_weight=0.5, smooth2_weight=0.0,
              with_occ=False, occ_aware=occ_aware)
    m = rngfree_fill(FlowStageModel(hp))
    batch = (T(g["stage_x"]), T(g["stage_gt_flow"]), T(g["stage_gt_occ"]))
    res = m.general_step_occ_aware(batch, 0, "train") if occ_aware else m.general_step(batch, 0, "train")
    close(torch.stack([r.detach() for r in res]), g[f"stage_{tag}_scalars"], 2e-5)
    loss = m.training_step(batch, 0)
    close(loss, g[f"stage_{tag}_loss"], 2e-5)
    loss.backward()
    close(m.flow_pred.predict_flow2.weight.grad[:, :16], g[f"stage_{tag}_grad_predict_flow2"], 5e-5)
    close(m.flow_pred.conv1a[0].weight.grad, g[f"stage_{tag}_grad_conv1a"], 5e-4)


def test_inpainting_head_cpu(cpu_ops, golden):
    """InpaintingNet / InpaintingModel mirror: state_dict keys + shapes + order of the reference's network, and the whole
    configs[3] chain (AC-False warp -> mask -> InpaintingNet -> masked L1) against the reference's golden at 2 x 64 x 96,
    including two weight gradients"""
    from ocflow_amd.models.inpainting_model import InpaintingModel
    from ocflow_amd.models.networks.image_inpainting_net import InpaintingNet
    g = golden("inpainting")
    with open(os.path.join(GOLDEN, "state_dict_manifest_inpainting.json")) as f:
        want = [(k, tuple(s)) for k, s in json.load(f)["inpaintingnet"]]
    net = InpaintingNet()
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == want
    m = InpaintingModel({"learning_rate": 1e-4})
    rngfree_fill(m.model)
    m.train()
    batch = (T(g["s_imgs"]), T(g["s_flow"]), T(g["s_occ"]))
    close(m.warp(batch[0][:, 3:6], batch[1]), g["s_warped"], 1e-5)
    loss = m.training_step(batch, 0)
    close(loss, g["s_loss"], 1e-5)
    loss.backward()
    close(m.model.down1.conv1.weight.grad, g["s_grad_down1_conv1"], 2e-5)
    close(m.model.up6.conv3.weight.grad, g["s_grad_up6_conv3"], 2e-5)
    with torch.no_grad():
        m2 = InpaintingModel({"learning_rate": 1e-4})
        rngfree_fill(m2.model)
        m2.train()
        img_occluded = m2.warp(batch[0][:, 3:6], batch[1]) * (1 - batch[2])
        close(m2(img_occluded, batch[2]), g["s_completed"], 2e-5)
    assert isinstance(m.configure_optimizers(), torch.optim.Adam)


def test_launcher_builds_the_reference_hparams(tmp_path):
    """`python -m ocflow_amd.train`: YAML -> hparams -> module as test_unsupervised.py:35-58 (network_type 'flow'), the shipped config,
    the !!python/tuple image_size, and the out-of-scope stages"""
    from ocflow_amd import train
    from ocflow_amd.models.model import FlowStageModel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = train.load_config(os.path.join(root, "config", "unsupervised_flow.yml"))
    assert args["image_size"] == (384, 512) and args["network_type"] == "flow"
    m, hp = train.build_module(args)
    assert isinstance(m, FlowStageModel) and m.model == "pwc" and m.occ_aware and m.photo_weight == 4.0 and m.smooth2_weight == 0.0
    assert hp["batch_size"] == 8 and hp["img_size"] == (384, 512) and "using_pretrained_inpainting" not in hp
    with pytest.raises(NotImplementedError):
        train.build_module({**args, "network_type": "twostage"})
    sup, _ = train.build_module({"network_type": "flowocc", "model": "flowoccnet", "learning_rate": 1e-3})
    assert type(sup).__name__ == "FlowOccModel"


def test_graph_env_guard_detects_late_import(monkeypatch):
    """ocflow_amd/__init__.py: the hipGraph environment switch is read when HIP initialises.  The guard reports 'late' exactly when the
    variable was absent and torch.cuda had already been initialised at import; a value chosen by the launcher is left alone."""
    import ocflow_amd
    import torch

    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", raising=False)
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    assert ocflow_amd._graph_env_guard() is True
    assert os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] == "0"
    assert ocflow_amd._graph_env_guard() is False  # now present: nothing to decide
    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", raising=False)
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: False)
    assert ocflow_amd._graph_env_guard() is False
    monkeypatch.setenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "1")
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    assert ocflow_amd._graph_env_guard() is False and os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] == "1"


def test_torch_library_ops_are_registered_with_fake_kernels():
    """ocflow_amd/torch_ops.py: dispatcher identities of the two core operators -- shape propagation on meta tensors needs no GPU"""
    import torch
    import ocflow_amd.torch_ops  # noqa: F401
    f = torch.empty((2, 32, 12, 20), device="meta")
    out = torch.ops.ocflow.cost_volume(f, f, 4)
    assert out.shape == (2, 81, 12, 20) and out.device.type == "meta"
    fl = torch.empty((2, 2, 12, 20), device="meta")
    assert torch.ops.ocflow.warp(f, fl, 1.25, False).shape == f.shape
    g1, g2 = torch.ops.ocflow.cost_volume_backward(f, f, out, 4)
    assert g1.shape == f.shape and g2.shape == f.shape
    assert "max_displacement" in str(torch.ops.ocflow.cost_volume.default._schema)
