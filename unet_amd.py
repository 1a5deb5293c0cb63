"""Import alias: loads the package in ``unet-based-network-traffic-prediction_amd/`` (a name
Python cannot import directly) as the module ``unet_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "unet-based-network-traffic-prediction_amd")
_spec = importlib.util.spec_from_file_location("unet_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["unet_amd"] = _mod
_spec.loader.exec_module(_mod)
