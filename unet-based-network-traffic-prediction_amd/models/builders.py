"""Model factory with the reference's interface (models/builders.py:83-132).

Only the U-Net is built here: it is the one model whose hot path this package
implements.  The reference's other ``model_type`` values (cnn_gru, cnn_lstm, tcn,
coregan -- models/builders.py:117-130) are different networks outside this package;
asking for them raises the same ``ValueError`` the reference raises for an unknown type,
with a message that says why."""
from __future__ import annotations

import torch.nn as nn

from .unet2d import UNet, build_unet_from_cfg

_OTHER_REFERENCE_MODELS = ("cnn_gru", "cnn_lstm", "tcn", "coregan")


def build_model_from_cfg(cfg) -> nn.Module:
    """``cfg.model_type`` (default "unet", case-insensitive as in models/builders.py:113)
    selects the model; ``"unet"`` returns the HIP-backed drop-in module."""
    model_type = getattr(cfg, "model_type", "unet").lower()
    if model_type == "unet":
        return build_unet_from_cfg(cfg)
    if model_type in _OTHER_REFERENCE_MODELS:
        raise ValueError(f"cfg.model_type={cfg.model_type!r} is a different network of the reference; "
                         "this package accelerates the 'unet' path only")
    raise ValueError(f"Unknown cfg.model_type: {cfg.model_type}")


__all__ = ["UNet", "build_unet_from_cfg", "build_model_from_cfg"]
