"""`SimpleOcclusionNet` -- the U-Net that predicts a soft occlusion map from an image pair, the first stage of the two-stage trainers
(reference: models/networks/simple_occlusion_net.py:6-143; used by models/model.py:946, :1117).

Same encoder / bottleneck blocks as `SimpleFlowNet` (simple_flow_net.py here) with a one-channel sigmoid head per level instead of the
flow head; parameter names, shapes, registration order and the Xavier initialisation are the reference's, so its checkpoints load
unchanged.  Dense convolutions only (MIOpen); what the hot path contributes to the two-stage step sits around it
(models/two_stage_model.py)."""
import torch
import torch.nn as nn

from .simple_flow_net import Upsample, _build_encoder, _encode, _xavier_init_, predict_occ


class SimpleOcclusionNet(nn.Module):
    def __init__(self, in_channels=6):
        super().__init__()
        _build_encoder(self, in_channels)
        cin = 128
        for i, s in enumerate((96, 64, 32, 16, in_channels), start=1):  # skip tensors: down4 .. down1, then the image pair itself
            cout = s if i < 5 else 16
            setattr(self, f"up{i}", Upsample(cin + s + 1, cout))
            cin = cout
        for lvl, c in zip((5, 4, 3, 2, 1, 0), (128, 96, 64, 32, 16, 16)):
            setattr(self, f"predict_occ{lvl}", predict_occ(c))
        _xavier_init_(self)

    def forward(self, img):
        """[B, 6, H, W] -> soft occlusion map [B, 1, H, W] in (0, 1) at full resolution (1 = occluded)"""
        xs = _encode(self, img)
        x = xs[5]
        for i in range(1, 6):
            occ = getattr(self, f"predict_occ{6 - i}")(x)
            x = getattr(self, f"up{i}")(torch.cat((x, occ), dim=1), xs[5 - i])
        return self.predict_occ0(x)
