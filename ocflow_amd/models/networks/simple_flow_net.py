"""`SimpleFlowNet` / `SimpleFlowOccNet` (model 'simple', the default of FlowModel / FlowOccModel): small U-Net
baselines without correlation or warp (reference: models/networks/simple_flow_net.py:6-143,
simple_flow_occ_net.py:6-162).  Pure PyTorch; present because BASELINE configs[0] (64x64 CPU forward of
models.flow_model) runs through it.  Parameter names are the reference's."""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init

_ENC = ((16, 1), (32, 2), (64, 4), (96, 4), (128, 4))  # (out channels, projection ratio) of down1..down5


class _Bottleneck(nn.Module):
    """conv1/bn1 -> conv2(3x3)/bn2 -> conv3(1x1)/bn3, LeakyReLU(0.1) after each"""

    def __init__(self, in_channels, out_channels, proj_ratio, first_kernel, first_stride):
        super().__init__()
        mid = in_channels // proj_ratio
        self.conv1 = nn.Conv2d(in_channels, mid, first_kernel, stride=first_stride, bias=False)
        self.bn1 = nn.BatchNorm2d(mid)
        self.lrelu1 = nn.LeakyReLU(0.1)
        self.conv2 = nn.Conv2d(mid, mid, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(mid)
        self.lrelu2 = nn.LeakyReLU(0.1)
        self.conv3 = nn.Conv2d(mid, out_channels, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(out_channels)
        self.lrelu3 = nn.LeakyReLU(0.1)

    def _body(self, x):
        x = self.lrelu1(self.bn1(self.conv1(x)))
        x = self.lrelu2(self.bn2(self.conv2(x)))
        return self.lrelu3(self.bn3(self.conv3(x)))


class Downsample(_Bottleneck):
    def __init__(self, in_channels, out_channels, proj_ratio=4):
        super().__init__(in_channels, out_channels, proj_ratio, 2, 2)

    def forward(self, x):
        return self._body(x)


class Upsample(_Bottleneck):
    def __init__(self, in_channels, out_channels, proj_ratio=4):
        super().__init__(in_channels, out_channels, proj_ratio, 1, 1)
        self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False)

    def forward(self, x1, x2):
        x1 = self.up(x1)
        dy, dx = x2.size(2) - x1.size(2), x2.size(3) - x1.size(3)
        x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return self._body(torch.cat([x2, x1], dim=1))


def conv(in_channels, out_channels, kernel_size=3, stride=1, activation=True):
    return nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=kernel_size // 2),
                         nn.LeakyReLU(0.1, inplace=True) if activation else nn.Identity())


def predict_flow(in_channels):
    return nn.Sequential(conv(in_channels, 32), conv(32, 16), conv(16, 2, activation=False))


def predict_occ(in_channels, is_last=False):
    return nn.Sequential(conv(in_channels, 32), conv(32, 16), conv(16, 1, activation=False),
                         nn.Sigmoid() if not is_last else nn.Identity())


def _xavier_init_(module):
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            if m.bias is not None:
                init.uniform_(m.bias)
            init.xavier_uniform_(m.weight)


def _build_encoder(net, in_channels):
    cin = in_channels
    for i, (cout, ratio) in enumerate(_ENC, start=1):
        setattr(net, f"down{i}", Downsample(cin, cout, proj_ratio=ratio))
        cin = cout


def _encode(net, img):
    xs = [img]
    for i in range(1, 6):
        xs.append(getattr(net, f"down{i}")(xs[-1]))
    return xs  # [img, x1..x5]


class SimpleFlowNet(nn.Module):
    def __init__(self, in_channels=6):
        super().__init__()
        _build_encoder(self, in_channels)
        skip = (96, 64, 32, 16, in_channels)
        cin = 128
        for i, s in enumerate(skip, start=1):
            cout = s if i < 5 else 16
            setattr(self, f"up{i}", Upsample(cin + s + 2, cout))
            cin = cout
        for lvl, c in zip((5, 4, 3, 2, 1, 0), (128, 96, 64, 32, 16, 16)):
            setattr(self, f"predict_flow{lvl}", predict_flow(c))
        _xavier_init_(self)

    def forward(self, img):
        xs = _encode(self, img)
        x = xs[5]
        for i in range(1, 6):
            flow = getattr(self, f"predict_flow{6 - i}")(x)
            x = getattr(self, f"up{i}")(torch.cat((x, flow), dim=1), xs[5 - i])
        return self.predict_flow0(x)


class SimpleFlowOccNet(nn.Module):
    def __init__(self, in_channels=6):
        super().__init__()
        _build_encoder(self, in_channels)
        cin = 128
        for i, s in enumerate((96, 64, 32, 16), start=1):
            setattr(self, f"up{i}", Upsample(cin + s + 3, s))
            cin = s
        widths = {5: 128, 4: 96, 3: 64, 2: 32, 1: 16}
        for lvl in (5, 4, 3, 2, 1):
            setattr(self, f"predict_flow{lvl}", predict_flow(widths[lvl]))
        self.upsample_flow = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False)
        for lvl in (5, 4, 3, 2, 1):
            setattr(self, f"predict_occ{lvl}", predict_occ(widths[lvl], is_last=(lvl == 1)))
        self.upsample_occ = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False)
        _xavier_init_(self)

    def forward(self, img):
        xs = _encode(self, img)
        x = xs[5]
        for i in range(1, 5):
            lvl = 6 - i
            flow, occ = getattr(self, f"predict_flow{lvl}")(x), getattr(self, f"predict_occ{lvl}")(x)
            x = getattr(self, f"up{i}")(torch.cat((x, flow, occ), dim=1), xs[5 - i])
        flow = self.upsample_flow(self.predict_flow1(x))
        occ_soft = torch.sigmoid(10 * self.upsample_occ(self.predict_occ1(x)))
        occ_hard = torch.where(occ_soft > 0.5, 1.0, 0.0)
        return flow, (occ_hard - occ_soft).detach() + occ_soft  # straight-through hard mask
