"""MI355X-native U-Net hot path for traffic-matrix prediction.

Drop-in for the reference's U-Net route: ``build_model_from_cfg(cfg)`` returns an
``nn.Module`` with the reference's parameter names and call signature whose forward and
backward run as hand-written gfx950 HIP kernels behind a C ABI (``include/unet_tm_hip.h``).
The directory name contains '-', so import it through the ``unet_amd`` module at the
repository root (``import unet_amd``)."""
from ._lib import LIB_PATH, HipLibraryError, lib
from .models import UNet, build_model_from_cfg, build_unet_from_cfg
from .loss import MSELoss, mse_loss, eval_sums
from .optim import FusedAdam
from .distributed import GradientReducer, broadcast_parameters, init_process_group_from_env, shard_batch
from .data import SeriesWindows, reconstruct_series, reference_shuffle, reference_open_loader
from . import harness
from . import ops

__all__ = ["UNet", "build_model_from_cfg", "build_unet_from_cfg", "MSELoss", "mse_loss", "eval_sums", "FusedAdam",
           "GradientReducer", "broadcast_parameters", "init_process_group_from_env", "shard_batch", "SeriesWindows", "reconstruct_series", "reference_shuffle",
           "reference_open_loader", "harness", "ops", "lib",
           "LIB_PATH", "HipLibraryError"]
