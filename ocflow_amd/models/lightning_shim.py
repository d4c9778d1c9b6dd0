"""`pl.LightningModule` when pytorch_lightning is installed, else a minimal stand-in with the hooks the reference's
modules use (hparams, log, logger, global_step, current_epoch), so the same classes drop into a Lightning Trainer or
into the plain loop of `python -m ocflow_amd.train` (ocflow_amd/train.py; this image ships no pytorch_lightning)."""
from __future__ import annotations

import torch.nn as nn

try:  # pragma: no cover - not installed in the build image
    import pytorch_lightning as pl

    LightningModule = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:
    HAVE_LIGHTNING = False

    class LightningModule(nn.Module):
        """Attribute surface used by the reference: models/flow_model.py:17-217, models/model.py:155-509."""

        def __init__(self):
            super().__init__()
            self.global_step = 0
            self.current_epoch = 0
            self.logger = None
            self.logged = {}

        def log(self, name, value, **_kwargs):
            self.logged[name] = value.detach() if hasattr(value, "detach") else value


def set_hparams(module, hparams: dict):
    """`self.hparams = hparams` (reference, Lightning 1.1) -- newer Lightning makes `hparams` read-only."""
    try:
        module.hparams = hparams
    except AttributeError:  # pragma: no cover
        module.save_hyperparameters(hparams)
