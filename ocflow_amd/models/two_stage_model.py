"""Two-stage trainers: occlusion prediction, then completion of the occluded pixels (reference: models/model.py:928-1103
`TwoStageModel`, :1105-1394 `TwoStageModelGC`).

`TwoStageModelGC` (ground-truth flow) is the one the reference's launcher trains (test_unsupervised.py:93-98).  One step
(model.py:1199-1243):

    img_warped = warp(img2, flow_gt)                         AC-True image warp                      -> ops.warp          (HIP, K2)
    soft       = SimpleOcclusionNet(imgs)                    soft occlusion map in (0, 1)            (MIOpen)
    hard       = (where(soft > 0.5, 1, 0) - soft).detach() + soft     straight-through hard mask (value 0/1, gradient of soft)
    smooth     = first_order_smoothness_loss(img_warped, soft)        edge-aware, on the MAP          -> ops.smoothness_losses (HIP, K5)
    completed  = inpainting(img_warped, soft)                                                         (MIOpen)
    photo      = photometric_error(img_warped * (1 - soft), img1 * (1 - soft))                        -> ops.photometric_error (HIP, K4)
    photo_occ  = photometric_error(img_warped * soft, img1 * soft)                                    -> ops.photometric_error
    reconst    = loss_func(soft * completed, soft * img1)             'pixel-wise' = Charbonnier mean -> ops.photometric_error
    pixelwise  = ReconLoss(1, 1, 1, 1)(completed, img1, soft)         hole / non-hole L1, each normalised by its mask mean per sample
    loss       = photo_weight * photo + reconst_weight * reconst + smooth1_weight * smooth + pixelwise_weight * pixelwise

The smoothness kernel takes a two-channel field: the one-channel map is passed as two views of itself (`expand`, no copy) -- the mean
over two identical channels is the mean over one, and autograd adds the two gradient planes back together.

Scope: `inpainting_stage='simple'` (InpaintingNet) or `'gated'` (InpaintSANet, networks/gated_conv_inpainting_net.py), trained from
scratch, with `loss_type='pixel-wise'`.  The VGG perceptual loss (pretrained torchvision weights), a pretrained completion network
(`using_pretrained_inpainting`: a Lightning checkpoint of the GAN trainer) and the non-projected generator `'gated_org'` raise
NotImplementedError naming the reference file.  Reference defects
not reproduced: `training_step` / `test_step` test `len(losses) == 3` although `general_step` returns 5 or 6 values, so a batch without
an occlusion label cannot train there (ValueError at model.py:1253); `TwoStageModel.general_step` reads `occ` before assignment for such
batches (:1030) and calls the inpainting network without its mask argument (:1023).  Here both batch forms work."""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.optim import Adam

from ocflow_amd import ops
from .lightning_shim import LightningModule, set_hparams
from .networks.gated_conv_inpainting_net import InpaintSANet
from .networks.image_inpainting_net import InpaintingNet
from .networks.simple_flow_net import SimpleFlowNet
from .networks.simple_occlusion_net import SimpleOcclusionNet


class PhotometricLoss(nn.Module):
    """models/model.py:47-51"""

    def forward(self, img_pred, img):
        return ops.photometric_error(img_pred, img)


class ReconLoss(nn.Module):
    """L1 inside and outside the hole, each divided by its mask's per-sample mean
    (reference: models/networks/gated_conv_inpainting_net.py:563-589).  -> (total, hole term, non-hole term)"""

    def __init__(self, chole_alpha=1, cunhole_alpha=1, rhole_alpha=1, runhole_alpha=1):
        super().__init__()
        self.chole_alpha, self.cunhole_alpha, self.rhole_alpha, self.runhole_alpha = chole_alpha, cunhole_alpha, rhole_alpha, runhole_alpha

    def forward(self, imgs, recon_imgs, masks, coarse_imgs=None):
        m = masks.reshape(masks.size(0), -1).mean(1).view(-1, 1, 1, 1)
        err = torch.abs(imgs - recon_imgs)
        rhole = torch.mean(err * masks / m)
        runhole = torch.mean(err * (1.0 - masks) / (1.0 - m))
        total = self.rhole_alpha * rhole + self.runhole_alpha * runhole
        if coarse_imgs is not None:
            cerr = torch.abs(imgs - coarse_imgs)
            total = total + self.chole_alpha * torch.mean(cerr * masks / m) + self.cunhole_alpha * torch.mean(cerr * (1.0 - masks) / (1.0 - m))
        return total, rhole, runhole


def straight_through_mask(soft):
    """value of the hard 0/1 mask, gradient of the soft one (model.py:1214)"""
    return (torch.where(soft > 0.5, 1.0, 0.0) - soft).detach() + soft


def _unpack(batch):
    if not isinstance(batch, (list, tuple)):
        return batch, None, None
    if len(batch) == 2:
        return batch[0], batch[1], None
    if len(batch) == 3:
        return batch
    raise ValueError("Not supported dataset")


def _bce(pred, target):
    """F.binary_cross_entropy in fp32 (autocast refuses it in reduced precision); identical in fp32 mode"""
    with torch.autocast(device_type=pred.device.type, enabled=False):
        return F.binary_cross_entropy(pred.float(), target.float())


def _map_smoothness(img, soft):
    """first_order_smoothness_loss(img, soft) for a ONE-channel map on the two-channel HIP kernel"""
    return ops.first_order_smoothness_loss(img, soft.expand(-1, 2, -1, -1))


class _TwoStageBase(LightningModule):
    def warp(self, img, flow):
        """image warp with align_corners=True (model.py:962-992 / :1159-1189)"""
        return ops.warp(img, flow, align_corners=True)

    @property
    def is_cuda(self):
        return next(self.parameters()).is_cuda

    def save_state_dict(self, path):
        torch.save(self.state_dict(), path)

    def _log_all(self, mode, names, values, on_step):
        for n, v in zip(names, values):
            if v is not None:
                self.log(f"{mode}_{n}", v, on_step=on_step, on_epoch=True, logger=True)


class TwoStageModelGC(_TwoStageBase):
    """Two stages with ground-truth optical flow (models/model.py:1105-1394)."""

    def __init__(self, hparams):
        super().__init__()
        set_hparams(self, hparams)
        self.lr = hparams["learning_rate"]
        self.reconst_weight = hparams.get("reconst_weight", 1.0)
        self.log_every_n_steps = hparams.get("log_every_n_steps", 1)
        self.occ_pred = SimpleOcclusionNet()
        self.inpainting_stage = hparams.get("inpainting_stage", "gated")
        self.img_size = hparams.get("img_size")
        self.loss_type = hparams.get("loss_type", "vgg")
        self.smooth1_weight = hparams.get("smooth1_weight", 1.0)
        self.smooth2_weight = hparams.get("smooth2_weight", 0.0)
        self.photo_weight = hparams.get("photo_weight", 0.0)
        self.pixelwise_weight = hparams.get("pixelwise_weight", 1.0)
        self.batch_size = hparams.get("batch_size", 16)
        self.using_pretrained_inpainting = hparams.get("using_pretrained_inpainting", True)
        if self.using_pretrained_inpainting:
            raise NotImplementedError("using_pretrained_inpainting=True loads an InpaintingStageModel checkpoint (models/model.py:1134): "
                                      "outside the hot path (SURVEY 8f-4); train the completion network from scratch")
        if self.inpainting_stage == "simple":
            self.inpainting = InpaintingNet()
        elif self.inpainting_stage == "gated":
            self.inpainting = InpaintSANet(img_size=self.img_size)
        else:
            raise NotImplementedError(f"inpainting_stage '{self.inpainting_stage}': InpaintSANetOrg (models/networks/gated_conv_inpainting_net.py:"
                                      "430-514, the non-projected generator) is not built; use 'gated' or 'simple'")
        if self.loss_type != "pixel-wise":
            raise NotImplementedError(f"loss_type '{self.loss_type}': VGGPerceptualLoss (models/model.py:116-153) needs pretrained torchvision "
                                      "weights; use 'pixel-wise'")
        self.loss_func = PhotometricLoss()
        self.pixelwise_loss_func = ReconLoss(1.0, 1.0, 1.0, 1.0)

    def forward(self, imgs):
        """soft occlusion map of an image pair"""
        return self.occ_pred(imgs)

    def general_step(self, batch, batch_idx, mode):
        imgs, flow, occ = _unpack(batch)
        if flow is None:
            raise ValueError("TwoStageModelGC needs the ground-truth flow in the batch")
        img1, img2 = imgs[:, 0:3], imgs[:, 3:6]
        img_warped = self.warp(img2, flow)
        soft = self.occ_pred(imgs)
        smoothness_loss = _map_smoothness(img_warped, soft)
        img_completed = self.inpainting(img_warped, soft)
        if self.inpainting_stage == "gated":
            img_completed = img_completed[1]  # (coarse, refined): the step uses the refined completion (model.py:1222)
        photo_error = ops.photometric_error(img_warped * (1 - soft), img1 * (1 - soft))
        photo_error_occluded = ops.photometric_error(img_warped * soft, img1 * soft)
        reconst_error = self.loss_func(soft * img_completed, soft * img1)
        pixelwise_loss, _, _ = self.pixelwise_loss_func(img_completed, img1, soft)
        bce_loss = _bce(soft, occ) if occ is not None else None
        return photo_error, reconst_error, smoothness_loss, pixelwise_loss, bce_loss, photo_error_occluded

    _NAMES = ("photometric", "reconst", "smoothness", "pixelwise", "bce_loss", "photometric_occluded")

    def _total(self, losses, pixelwise_weight):
        photo_error, reconst_error, smoothness_loss, pixelwise_loss = losses[:4]
        return (self.photo_weight * photo_error + self.reconst_weight * reconst_error + self.smooth1_weight * smoothness_loss
                + pixelwise_weight * pixelwise_loss)

    def training_step(self, batch, batch_idx):
        losses = self.general_step(batch, batch_idx, "train")
        self._log_all("train", self._NAMES, losses, True)
        return self._total(losses, self.pixelwise_weight)

    def validation_step(self, batch, batch_idx):
        """model.py:1277-1358 without the image dumps: the same loss; 'monitored_loss' is what the early-stopping callback watches"""
        losses = self.general_step(batch, batch_idx, "val")
        loss = self._total(losses, self.pixelwise_weight)
        self._log_all("val", self._NAMES, losses, False)
        self.log("monitored_loss", loss, prog_bar=True, logger=True)
        return loss

    def test_step(self, batch, batch_idx):
        losses = self.general_step(batch, batch_idx, "test")
        self._log_all("test", self._NAMES, losses, False)
        return self._total(losses, 1.0)  # (model.py:1374 adds the pixel-wise term unweighted)

    def hard_mask(self, imgs):
        """the straight-through 0/1 mask the validation images are rendered with (model.py:1292-1295)"""
        return straight_through_mask(self.occ_pred(imgs))

    def configure_optimizers(self):
        return Adam(self.parameters(), self.lr)


class TwoStageModel(_TwoStageBase):
    """Two stages with a frozen flow network and a frozen completion network: only the occlusion network trains
    (models/model.py:928-1103).  Checkpoint loading of the two frozen networks (`flow_root`, `inpainting_root`: Lightning checkpoints of
    other trainers) is replaced by `load_state_dict` on `self.flow_pred` / `self.inpainting` by the caller."""

    def __init__(self, hparams):
        super().__init__()
        set_hparams(self, hparams)
        self.lr = hparams["learning_rate"]
        self.smoothness_weight = hparams.get("smoothness_weight", 0.0)
        self.reconst_weight = hparams.get("reconst_weight", 1.0)
        self.log_every_n_steps = hparams.get("log_every_n_steps", 1)
        if hparams.get("flow_root") or hparams.get("inpainting_root"):
            raise NotImplementedError("flow_root / inpainting_root load Lightning checkpoints of FlowModel / FlowStageModel / "
                                      "InpaintingStageModel (models/model.py:947-953); load state_dicts into .flow_pred / .inpainting instead")
        self.flow_pred = SimpleFlowNet()
        self.occ_pred = SimpleOcclusionNet()
        self.inpainting = InpaintingNet()
        for net in (self.flow_pred, self.inpainting):
            for p in net.parameters():
                p.requires_grad = False

    def forward(self, imgs):
        return self.occ_pred(imgs)

    def general_step(self, batch, batch_idx, mode):
        imgs, _, occ = _unpack(batch)
        img1, img2 = imgs[:, 0:3], imgs[:, 3:6]
        flow_pred = self.flow_pred(imgs)
        occ_pred = self.occ_pred(imgs)
        img_warped = self.warp(img2, flow_pred)
        img_completed = self.inpainting(img_warped, occ_pred)  # (the reference's call lacks the mask argument: SURVEY F7)
        smoothness_term = ops.first_order_smoothness_loss(img1, flow_pred)
        photo_error = ops.photometric_error(img_warped * (1 - occ_pred), img1 * (1 - occ_pred))
        reconst_error = ops.photometric_error(img_warped * occ_pred, img1 * occ_pred)
        bce_loss = _bce(occ_pred, occ) if occ is not None else None
        self.last_completed = img_completed.detach()
        return photo_error, reconst_error, smoothness_term, bce_loss

    _NAMES = ("photometric", "reconst", "smoothness", "bce_loss")

    def _step(self, batch, batch_idx, mode):
        losses = self.general_step(batch, batch_idx, mode)
        self._log_all(mode, self._NAMES, losses, mode == "train")
        return losses[0] + self.reconst_weight * losses[1] + self.smoothness_weight * losses[2]

    def training_step(self, batch, batch_idx):
        return self._step(batch, batch_idx, "train")

    def validation_step(self, batch, batch_idx):
        loss = self._step(batch, batch_idx, "val")
        self.log("monitored_loss", loss, prog_bar=True, logger=True)
        return loss

    def test_step(self, batch, batch_idx):
        return self._step(batch, batch_idx, "test")

    def configure_optimizers(self):
        return Adam(self.parameters(), self.lr)
