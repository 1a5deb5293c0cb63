"""Development aid: run a script against another build of the library (A/B timing of kernel variants).

    python scripts/with_lib.py path/to/libunet_tm_hip.so scripts/bench_conv.py --only fwd ...
"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd import _lib

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
unet_amd.LIB_PATH = _lib.LIB_PATH
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
