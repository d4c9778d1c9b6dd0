"""`FlowOccModel` -- supervised flow + occlusion surface (reference: models/flow_occ_model.py:17-92).

`FlowOccModel(hparams)`; `forward(x[B,6,H,W]) -> (flow[B,2,H,W], occ[B,1,H,W])`; loss = L1(flow) + BCE(occ);
Adam over `self.model.parameters()` with `hparams['learning_rate']`."""
import torch
import torch.nn.functional as F

from ocflow_amd import ops
from torch.optim import Adam

from .lightning_shim import LightningModule, set_hparams
from .networks.cost_volume_flow_occ_net import FlowOccNetCV
from .networks.flow_occ_net import FlowOccNet
from .networks.simple_flow_net import SimpleFlowOccNet

_OUT_OF_SCOPE = ("flowoccnets", "flowoccnetc")


def build_flow_occ_network(model: str):
    if model == "simple":
        return SimpleFlowOccNet()
    if model == "pwoc":
        return FlowOccNetCV()
    if model == "flowoccnet":
        return FlowOccNet()
    if model in _OUT_OF_SCOPE:
        raise NotImplementedError(f"model '{model}' is outside the hot path of the MI355X build (SURVEY.md section 2.1)")
    raise ValueError(f"Unsupported model: {model}")


class FlowOccModel(LightningModule):
    def __init__(self, hparams):
        super().__init__()
        set_hparams(self, hparams)
        self.model = build_flow_occ_network(hparams.get("model", "simple"))

    def forward(self, x):
        return self.model(x)

    @property
    def is_cuda(self):
        return next(self.parameters()).is_cuda

    def save_state_dict(self, path):
        torch.save(self.state_dict(), path)

    def general_step(self, batch, batch_idx, mode):
        imgs, flow, occ = batch
        if getattr(self.model, "lowres_head", False) and imgs.is_cuda:
            # the two x4 up-samplings and both losses on the quarter-resolution outputs in one forward and one backward launch
            flow_lr, occ_lr = self.model(imgs, upsample=False)
            if ops.flow_occ_head_eligible(flow_lr, occ_lr, flow, occ):
                return ops.flow_occ_head_loss(flow_lr, occ_lr, flow, occ)
            flow_pred, occ_pred = self.model.upsample1(flow_lr), self.model.upsample2(occ_lr)
        else:
            flow_pred, occ_pred = self.forward(imgs)
        flow_loss = F.l1_loss(flow_pred.float(), flow)
        # BCE is evaluated in fp32 (autocast refuses it in reduced precision); identical in fp32 mode
        with torch.autocast(device_type=imgs.device.type, enabled=False):
            occ_loss = F.binary_cross_entropy(occ_pred.float(), occ)
        return flow_loss, occ_loss

    def general_epoch_end(self, outputs, mode):
        return torch.stack([o[mode + "_loss"] for o in outputs]).cpu().mean()

    def training_step(self, batch, batch_idx):
        flow_loss, occ_loss = self.general_step(batch, batch_idx, "train")
        loss = flow_loss + occ_loss
        self.log("train_loss", loss, prog_bar=True, on_step=True, on_epoch=True, logger=True)
        return loss

    def _eval_step(self, batch, batch_idx, mode):
        flow_loss, occ_loss = self.general_step(batch, batch_idx, mode)
        loss = flow_loss + occ_loss
        self.log(f"{mode}_flow_loss", flow_loss, logger=True)
        self.log(f"{mode}_occ_loss", occ_loss, logger=True)
        self.log(f"{mode}_loss", loss, prog_bar=True, logger=True)
        return loss

    def validation_step(self, batch, batch_idx):
        return self._eval_step(batch, batch_idx, "val")

    def test_step(self, batch, batch_idx):
        return self._eval_step(batch, batch_idx, "test")

    def configure_optimizers(self):
        return Adam(self.model.parameters(), self.hparams["learning_rate"])
