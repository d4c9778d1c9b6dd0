"""`InpaintingModel` -- the completion head trained with ground-truth flow and occlusion
(reference: models/inpainting_model.py:10-96; BASELINE configs[3]: KITTI-sized pairs, occlusion mask, inpainting head).

One step (inpainting_model.py:62-75):  img_warped = warp(img2, flow)  [AC-False, the reference's own copy of the feature warp,
:22-52]  ->  img_occluded = img_warped * (1 - occ)  ->  completed = InpaintingNet(img_occluded, occ)  ->
loss = sum(|completed - img1| * occ) / (3 * sum(occ) + 1e-16).

The reference's `forward` calls `InpaintingNet(x)` without the `masks` argument the network requires (it cannot run
as shipped); here `forward(x, masks)` passes the mask, which is the only reading under which `general_step` is well defined.
Hot-path pieces on HIP: the warp (ops.warp) and the masked-L1 reduction
(ops.masked_l1, one pass forward, one backward)."""
import torch
from torch.optim import Adam

from ocflow_amd import ops
from .lightning_shim import LightningModule, set_hparams
from .networks.image_inpainting_net import InpaintingNet


class InpaintingModel(LightningModule):
    def __init__(self, hparams):
        super().__init__()
        set_hparams(self, hparams)
        self.model = InpaintingNet()

    def forward(self, x, masks):
        return self.model(x, masks)

    def warp(self, img, flow):
        """image warp with the align_corners=False sampling of inpainting_model.py:22-52"""
        return ops.warp(img, flow, align_corners=False)

    @property
    def is_cuda(self):
        return next(self.parameters()).is_cuda

    def save_state_dict(self, path):
        torch.save(self.state_dict(), path)

    def general_step(self, batch, batch_idx, mode):
        imgs, flow, occ = batch
        img1, img2 = imgs[:, 0:3], imgs[:, 3:6]
        img_occluded = self.warp(img2, flow) * (1 - occ)  # 1: occluded, 0: non-occluded
        img_completed = self.forward(img_occluded, occ)
        return ops.masked_l1(img_completed, img1, occ)

    def training_step(self, batch, batch_idx):
        loss = self.general_step(batch, batch_idx, "train")
        self.log("train_loss", loss, prog_bar=True, on_step=True, on_epoch=True, logger=True)
        return loss

    def validation_step(self, batch, batch_idx):
        loss = self.general_step(batch, batch_idx, "val")
        self.log("val_loss", loss, prog_bar=True, logger=True)
        return loss

    def test_step(self, batch, batch_idx):
        loss = self.general_step(batch, batch_idx, "test")
        self.log("test_loss", loss, prog_bar=True, logger=True)
        return loss

    def configure_optimizers(self):
        return Adam(self.model.parameters(), self.hparams["learning_rate"])
