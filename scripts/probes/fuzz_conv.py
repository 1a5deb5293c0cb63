"""Randomised parity sweep of the 3x3 conv and transposed-conv entry points (forward, input gradient, weight
gradient) against torch on the CPU, both dtypes: shapes drawn around the borders between the kernels
(few-pixel / pixel-tiled / position-major, halo-tile / generic weight gradient).  Prints the failures.
    python scripts/probes/fuzz_conv.py [cases] [seed]"""
import os, random, sys, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import unet_amd
import test_gpu_ops as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for i in range(n):
    H = rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 32, 64]); W = rng.choice([2, 4, 6, 8, 16, 32, 64, 128])
    while H * W > 4096: H = max(1, H // 2)
    cin = rng.choice([3, 12, 16, 32, 48, 64, 128, 256]); cout = rng.choice([8, 16, 32, 64, 96, 128, 256])
    maxb = max(1, min(64, (1 << 21) // (H * W * max(cin, cout))))
    B = rng.choice([1, 2, 3, 8, maxb, max(1, maxb // 2)])
    split = rng.choice([0, 0, 0, cin // 2]) if cin >= 32 and cin % 32 == 0 else 0
    dtype = rng.choice(T.DTYPES)
    shape = (B, cin, cout, H, W, split)
    try:
        T._check_conv3x3(unet_amd, dtype, shape)
    except Exception as e:                       # noqa: BLE001 - report every kind of failure
        bad += 1
        print("FAIL", dtype, shape, type(e).__name__, str(e)[:200], flush=True)
        if not isinstance(e, AssertionError):
            traceback.print_exc()
    # transposed conv 2x2 stride 2 on a related shape (the test function builds its own tensors from the shape)
    tshape = (B, rng.choice([32, 64, 128, 256, 512]), rng.choice([16, 24, 32, 64, 128, 256]), max(1, H // 2), max(1, min(W, 32) // 2))
    try:
        T.test_convt2x2.__wrapped__(unet_amd, dtype, tshape) if hasattr(T.test_convt2x2, "__wrapped__") else T.test_convt2x2(unet_amd, dtype, tshape)
    except Exception as e:                       # noqa: BLE001
        bad += 1
        print("FAIL convt", dtype, tshape, type(e).__name__, str(e)[:200], flush=True)
        if not isinstance(e, AssertionError):
            traceback.print_exc()
    if i % 20 == 19:
        print("..", i + 1, "cases,", bad, "failures", flush=True)
print("done:", n, "cases,", bad, "failures")
sys.exit(1 if bad else 0)
