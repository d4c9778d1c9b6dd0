"""`FlowModel` -- supervised optical-flow surface (reference: models/flow_model.py:17-217).

Same constructor (`FlowModel(hparams: dict)`), hparams keys (`learning_rate` required, `model`, `displacement`),
forward contract (`x[B,6,H,W]` image pair in [-1,1] -> full-resolution flow), step / optimizer methods and helper
methods (`warp`, `flow_to_warp`, `compute_range_map`, `save_state_dict`, `is_cuda`).  The helpers that the reference
spells out with ATen op chains are the HIP operators here."""
import torch
import torch.nn.functional as F
from torch.optim import Adam

from ocflow_amd import ops
from .lightning_shim import LightningModule, set_hparams
from .networks.cost_volume_flow_net import FlowNetCV
from .networks.flow_net import FlowNet
from .networks.simple_flow_net import SimpleFlowNet

# reference model names whose networks are outside the hot path (dense-conv-only, broken or dead code: SURVEY 2.1, F7)
_OUT_OF_SCOPE = ("flownets", "flownetc", "eflownet", "eflownet2")


def build_flow_network(model: str, displacement: int = 4):
    if model == "simple":
        return SimpleFlowNet()
    if model == "pwc":
        return FlowNetCV(displacement=displacement)
    if model == "flownet":
        return FlowNet()
    if model in _OUT_OF_SCOPE:
        raise NotImplementedError(f"model '{model}' has no correlation/warp hot path and is not part of the MI355X build "
                                  "(SURVEY.md section 2.1)")
    raise ValueError(f"Unsupported model: {model}")


class HotPathHelpers:
    """warp / flow_to_warp / compute_range_map / is_cuda / save_state_dict shared by FlowModel and FlowStageModel
    (the reference duplicates them: flow_model.py:49-171, model.py:191-313)."""

    def warp(self, img, flow):
        """loss-side image warp, align_corners=True (flow_model.py:49-79)"""
        return ops.warp(img, flow, align_corners=True)

    def flow_to_warp(self, flow):
        """[B,H,W,2] flow -> [B,H,W,2] end points (pixel grid + flow); flow_model.py:81-99"""
        B, H, W, _ = flow.size()
        ys, xs = torch.meshgrid(torch.arange(H, device=flow.device, dtype=flow.dtype),
                                torch.arange(W, device=flow.device, dtype=flow.dtype), indexing="ij")
        return torch.stack((xs, ys), dim=-1).unsqueeze(0) + flow

    def compute_range_map(self, flow):
        """[B,2,H,W] flow -> [B,1,H,W] range map (flow_model.py:101-163)"""
        return ops.compute_range_map(flow)

    @property
    def is_cuda(self):
        return next(self.parameters()).is_cuda

    def save_state_dict(self, path):
        torch.save(self.state_dict(), path)


class FlowModel(HotPathHelpers, LightningModule):
    def __init__(self, hparams):
        super().__init__()
        set_hparams(self, hparams)
        self.lr = hparams["learning_rate"]
        self.model = hparams.get("model", "simple")
        self.flow_pred = build_flow_network(self.model, hparams.get("displacement", 4))

    def forward(self, x):
        out = self.flow_pred(x)
        return out[0] if self.model == "pwc" else out

    def general_step(self, batch, batch_idx, mode):
        if not isinstance(batch, (list, tuple)) or len(batch) not in (2, 3):
            raise ValueError("Not supported dataset")
        imgs, flow = batch[0], batch[1]
        return F.mse_loss(self(imgs).float(), flow)

    def training_step(self, batch, batch_idx):
        return self.general_step(batch, batch_idx, "train")

    def validation_step(self, batch, batch_idx):
        return self.general_step(batch, batch_idx, "val")

    def test_step(self, batch, batch_idx):
        return self.general_step(batch, batch_idx, "test")

    def _epoch_end(self, outputs, tag, monitor=False):
        avg = torch.stack([o["loss"] if isinstance(o, dict) else o for o in outputs]).mean()
        if self.logger is not None:
            self.logger.experiment.add_scalars("losses", {f"{tag}_loss": avg}, global_step=self.current_epoch)
        if monitor:
            self.log("monitored_loss", avg, prog_bar=True, logger=True)
        return avg

    def training_epoch_end(self, outputs):
        self._epoch_end(outputs, "train")

    def validation_epoch_end(self, outputs):
        self._epoch_end(outputs, "val", monitor=True)

    def test_epoch_end(self, outputs):
        self._epoch_end(outputs, "test")

    def configure_optimizers(self):
        return Adam(self.parameters(), self.lr)
