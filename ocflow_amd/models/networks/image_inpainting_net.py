"""`InpaintingNet` -- the small encoder/decoder completion network of the inpainting head
(reference: models/networks/image_inpainting_net.py:5-103; 0.22 M parameters, BASELINE configs[3]).

Dense convolutions only: it stays in PyTorch-ROCm (MIOpen), same parameter names / shapes / order and the same initialisation, so
reference checkpoints load unchanged.  What the hot path contributes to this head sits around it (models/inpainting_model.py): the
AC-False image warp, the occlusion masking and the masked-L1 reduction."""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init


class _ConvBNAct(nn.Module):
    """three conv -> BN -> LeakyReLU(0.1) stages with the reference's attribute names (conv1/bn1/lrelu1 ...)"""

    def _stage(self, i, cin, cout, k, stride=1, act=True):
        setattr(self, f"conv{i}", nn.Conv2d(cin, cout, k, stride=stride, padding=0 if stride == 2 else k // 2, bias=False))
        setattr(self, f"bn{i}", nn.BatchNorm2d(cout) if act else nn.Identity())
        setattr(self, f"lrelu{i}", nn.LeakyReLU(0.1) if act else nn.Identity())

    def _run(self, x):
        for i in (1, 2, 3):
            x = getattr(self, f"lrelu{i}")(getattr(self, f"bn{i}")(getattr(self, f"conv{i}")(x)))
        return x


class Downsample(_ConvBNAct):
    """2x2 stride-2 projection, k x k conv, 1x1 expansion (image_inpainting_net.py:5-24)"""

    def __init__(self, in_channels, out_channels, kernel_size=3, proj_ratio=4):
        super().__init__()
        inter = in_channels // proj_ratio
        self._stage(1, in_channels, inter, 2, stride=2)
        self._stage(2, inter, inter, kernel_size)
        self._stage(3, inter, out_channels, 1)

    def forward(self, x):
        return self._run(x)


class Upsample(_ConvBNAct):
    """bilinear x2, pad to the skip tensor, concat [skip, up], 1x1 -> 3x3 -> 1x1 (image_inpainting_net.py:26-56)"""

    def __init__(self, in_channels, out_channels, proj_ratio=4, activation=True):
        super().__init__()
        inter = in_channels // proj_ratio
        self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False)
        self._stage(1, in_channels, inter, 1)
        self._stage(2, inter, inter, 3)
        self._stage(3, inter, out_channels, 1, act=activation)

    def forward(self, x1, x2):
        x1 = self.up(x1)
        dy, dx = x2.size(2) - x1.size(2), x2.size(3) - x1.size(3)
        x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return self._run(torch.cat([x2, x1], dim=1))


class InpaintingNet(nn.Module):
    def __init__(self, in_channels=4):
        super().__init__()
        self.down1 = Downsample(in_channels, 32, kernel_size=7, proj_ratio=1)
        self.down2 = Downsample(32, 64, kernel_size=5)
        self.down3 = Downsample(64, 128, kernel_size=5)
        self.down4 = Downsample(128, 128)
        self.down5 = Downsample(128, 128)
        self.down6 = Downsample(128, 128)
        self.up1 = Upsample(128 + 128, 128, proj_ratio=8)
        self.up2 = Upsample(128 + 128, 128, proj_ratio=8)
        self.up3 = Upsample(128 + 128, 128, proj_ratio=8)
        self.up4 = Upsample(128 + 64, 64)
        self.up5 = Upsample(64 + 32, 32)
        self.up6 = Upsample(32 + 3, 3, activation=False)
        self.tanh = nn.Tanh()
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                if m.bias is not None:
                    init.uniform_(m.bias)
                init.xavier_uniform_(m.weight)

    def forward(self, imgs, masks):
        """imgs [B,3,H,W], masks [B,1,H,W] (1 = pixel to complete) -> completed image in (-1, 1)"""
        masked = imgs * (1 - masks)
        x1 = self.down1(torch.cat([masked, masks], dim=1))
        x2 = self.down2(x1)
        x3 = self.down3(x2)
        x4 = self.down4(x3)
        x5 = self.down5(x4)
        x6 = self.down6(x5)
        x = self.up1(x6, x5)
        x = self.up2(x, x4)
        x = self.up3(x, x3)
        x = self.up4(x, x2)
        x = self.up5(x, x1)
        return self.tanh(self.up6(x, masked))
