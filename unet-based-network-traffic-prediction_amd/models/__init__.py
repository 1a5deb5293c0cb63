"""Same import surface as the reference's ``models`` package for the U-Net path
(models/__init__.py:3-20): ``from models import UNet, build_unet_from_cfg, build_model_from_cfg``."""
from .unet2d import UNet
from .builders import build_unet_from_cfg, build_model_from_cfg

__all__ = ["UNet", "build_unet_from_cfg", "build_model_from_cfg"]
