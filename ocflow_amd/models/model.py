"""Unsupervised flow training stage: loss functions + `FlowStageModel` (reference: models/model.py:27-114, 155-509).

The free functions keep the reference's names and signatures and are the HIP operators; `FlowStageModel` keeps the
reference's hparams, step methods, logged scalar names and optimizer.  The training step's loss side -- image warp +
Charbonnier photometric error (+ occlusion mask), first/second-order edge-aware smoothness, range-map occlusion of
the backward flow -- runs as fused HIP kernels (ops.warp_photometric_error, ops.smoothness_losses,
ops.occlusion_from_flow).  The two-stage trainers (model.py:928-1394) are in two_stage_model.py and re-exported here; the
inpainting-stage GAN trainers (model.py:511-926) are out of scope."""
import os

import torch
import torch.nn.functional as F
from torch.optim import Adam

from ocflow_amd import ops
from ocflow_amd.ops import (first_order_smoothness_loss, gradient, photometric_error, robust_l1,  # noqa: F401  (re-exported)
                            second_order_smoothness_loss)
from .flow_model import HotPathHelpers, build_flow_network
from .lightning_shim import LightningModule, set_hparams
from .two_stage_model import ReconLoss, TwoStageModel, TwoStageModelGC  # noqa: F401  (models/model.py:928-1394 live in two_stage_model.py)

__all__ = ["robust_l1", "photometric_error", "gradient", "first_order_smoothness_loss", "second_order_smoothness_loss",
           "PhotometricLoss", "FlowStageModel", "TwoStageModel", "TwoStageModelGC", "ReconLoss"]


class PhotometricLoss(torch.nn.Module):
    """models/model.py:47-51"""

    def forward(self, img_pred, img):
        return photometric_error(img_pred, img)


class FlowStageModel(HotPathHelpers, LightningModule):
    """Training with one stage: optical flow prediction (models/model.py:155-509)."""

    def __init__(self, hparams):
        super().__init__()
        set_hparams(self, hparams)
        self.lr = hparams["learning_rate"]
        self.photo_weight = hparams.get("photo_weight", 1.0)
        self.smooth1_weight = hparams.get("smooth1_weight", 0.0)
        self.smooth2_weight = hparams.get("smooth2_weight", 1.0)
        self.with_occ = hparams.get("with_occ", False)
        self.log_every_n_steps = hparams.get("log_every_n_steps", 20)
        self.occ_aware = hparams.get("occ_aware", False)
        self.displacement = hparams.get("displacement", 4)
        # photometric term: 'charbonnier' = the reference's robust-L1 (models/model.py:27-46, the default and the only mode the
        # reference trains with); 'ssim' = 1 - SSIM(warp(img2, flow), img1) with the reference's own SSIM definition
        # (inpainting_metrics/ssim/ssim.py:17-37, Gaussian window `ssim_window`), the SURVEY 8f-2 option.  SSIM has no per-pixel
        # mask: the occlusion mask is ignored in that mode.
        self.photo_mode = hparams.get("photo_mode", "charbonnier")
        if self.photo_mode not in ("charbonnier", "ssim"):
            raise ValueError(f"Unsupported photo_mode: {self.photo_mode}")
        self.ssim_window = hparams.get("ssim_window", 11)
        self.model = hparams.get("model", "simple")
        self.flow_pred = build_flow_network(self.model, self.displacement)

    def forward(self, x):
        return self.flow_pred(x)

    # ------------------------------------------------------------------------------------------ loss side
    def _predict(self, imgs, net_out=None):
        """-> (full-res flow, flow used by the smoothness terms, image used by the smoothness terms)"""
        img1 = imgs[:, 0:3]
        if self.model == "pwc":
            flow_pred, flow_l2 = self(imgs) if net_out is None else net_out
            if img1.is_cuda and img1.dtype == torch.float32 and not img1.requires_grad and not torch.compiler.is_compiling():
                # floor(size * 0.25): the output size F.interpolate derives from the scale factor (one launch instead of the framework's
                # per-pixel loop over the strided channel slice: 75 us at 8 x 384 x 512)
                img1_l2 = ops.resize_bilinear(img1, (int(img1.shape[2] * 0.25), int(img1.shape[3] * 0.25)), align_corners=True)
            else:
                img1_l2 = F.interpolate(img1, scale_factor=0.25, mode="bilinear", align_corners=True)
            return flow_pred, flow_l2, img1_l2
        flow_pred = self(imgs) if net_out is None else net_out
        return flow_pred, flow_pred, img1

    @staticmethod
    def _unpack(batch):
        if len(batch) == 2:
            return batch[0], batch[1], None
        if len(batch) == 3:
            return batch
        raise ValueError("Not supported dataset")

    def _losses(self, imgs, flow_gt, occ_mask, pred=None):
        img1, img2 = imgs[:, 0:3], imgs[:, 3:6]
        flow_pred, flow_s, img_s = pred if pred is not None else self._predict(imgs)
        if self.photo_mode == "ssim":
            photo = 1.0 - ops.ssim(ops.warp(img2.float(), flow_pred.float(), align_corners=True), img1, self.ssim_window)
        else:
            photo = ops.warp_photometric_error(img1, img2, flow_pred, occ_mask)
        smooth1, smooth2 = ops.smoothness_losses(img_s, flow_s)
        flow_error = F.mse_loss(flow_pred.float(), flow_gt)
        return flow_pred, photo, smooth1, smooth2, flow_error

    def general_step(self, batch, batch_idx, mode):
        """photometric (no mask) + smoothness  (models/model.py:315-341)"""
        imgs, flow, _ = self._unpack(batch)
        return self._losses(imgs, flow, None)[1:]

    def general_step_occ(self, batch, batch_idx, mode):
        """photometric masked by the ground-truth occlusion  (models/model.py:343-364)"""
        imgs, flow, occ = batch
        return self._losses(imgs, flow, occ)[1:]

    def general_step_occ_aware(self, batch, batch_idx, mode):
        """photometric masked by the range-map occlusion of the backward flow  (models/model.py:366-409)"""
        imgs, flow, occ = self._unpack(batch)
        img1, img2 = imgs[:, 0:3], imgs[:, 3:6]
        two_way = getattr(self.flow_pred, "forward_two_way", None)
        if two_way is not None and os.environ.get("OCFLOW_SHARED_ENCODER", "1") != "0":
            # both directions from ONE encoder pass (FlowNetCV: no batch-coupled layer, the swapped call's pyramids are these, swapped)
            out, back = two_way(imgs)
            pred = self._predict(imgs, out)
        else:
            pred = self._predict(imgs)  # forward flow first, then the no-grad backward flow, as the reference orders them
            with torch.no_grad():
                back = self(torch.cat((img2, img1), dim=1))
        with torch.no_grad():
            back_flow = back[0] if self.model == "pwc" else back
            occ_pred = ops.occlusion_from_flow(back_flow)  # 0: non-occluded, 1: occluded
        flow_pred, photo, smooth1, smooth2, flow_error = self._losses(imgs, flow, occ_pred, pred)
        with torch.no_grad():  # logging-only terms
            photo_occ = ops.warp_photometric_error(img1, img2, flow_pred.detach(), 1.0 - occ_pred)
            if occ is not None:
                # argument order as in the reference (input=occ, target=occ_pred): model.py:407, logged only
                with torch.autocast(device_type=imgs.device.type, enabled=False):  # BCE refuses reduced-precision autocast
                    occ_error = F.binary_cross_entropy(occ.float(), occ_pred.float())
                return photo, smooth1, smooth2, flow_error, photo_occ, occ_error
        return photo, smooth1, smooth2, flow_error, photo_occ

    # ------------------------------------------------------------------------------------------ steps
    def _step(self, batch, batch_idx, mode):
        photo_occ = occ_error = None
        if not self.occ_aware:
            fn = self.general_step_occ if self.with_occ else self.general_step
            photo, smooth1, smooth2, flow_error = fn(batch, batch_idx, mode)
        else:
            res = self.general_step_occ_aware(batch, batch_idx, mode)
            photo, smooth1, smooth2, flow_error, photo_occ = res[:5]
            occ_error = res[5] if len(res) == 6 else None
        loss = self.photo_weight * photo + self.smooth1_weight * smooth1 + self.smooth2_weight * smooth2
        if self.logger is not None and self.global_step % self.log_every_n_steps == 0:
            tb = self.logger.experiment
            for name, val in (("photometric", photo), ("smooth1", smooth1), ("smooth2", smooth2), ("flow_error", flow_error),
                              ("occ_error", occ_error), ("photometric_occ", photo_occ)):
                if val is not None:
                    tb.add_scalar(f"{mode}_{name}", val, global_step=self.global_step)
        return loss

    def training_step(self, batch, batch_idx):
        return self._step(batch, batch_idx, "train")

    def validation_step(self, batch, batch_idx):
        return self._step(batch, batch_idx, "val")

    def test_step(self, batch, batch_idx):
        return self._step(batch, batch_idx, "test")

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
