"""Micro-benchmark of single conv layers through the C ABI (development aid).

    python scripts/bench_conv.py [--dtype bf16] [--iters 50] [--only fwd,dgrad,wgrad] [--shapes name,...]

Shapes are the layers of the default workload (batch 512, 32x32, 12 channels in)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import unet_amd
from unet_amd import ops

SHAPES = {  # name: (B, H, W, Cin, split, Cout)
    "L0a": (512, 32, 32, 12, 0, 32), "L0b": (512, 32, 32, 32, 0, 32), "L0c": (512, 32, 32, 64, 32, 32),
    "L1a": (512, 16, 16, 32, 0, 64), "L1b": (512, 16, 16, 64, 0, 64), "L1c": (512, 16, 16, 128, 64, 64),
    "L2a": (512, 8, 8, 64, 0, 128), "L2b": (512, 8, 8, 128, 0, 128), "L2c": (512, 8, 8, 256, 128, 128),
    "L3a": (512, 4, 4, 128, 0, 256), "L3b": (512, 4, 4, 256, 0, 256), "L3c": (512, 4, 4, 512, 256, 256),
    "BNa": (512, 2, 2, 256, 0, 512), "BNb": (512, 2, 2, 512, 0, 512),
    # probes: same layers with input-channel counts that are not powers of two (weight-row stride)
    "X3c": (512, 4, 4, 544, 0, 256), "X2c": (512, 8, 8, 288, 0, 128), "X3n": (512, 4, 4, 512, 0, 256), "X2n": (512, 8, 8, 256, 0, 128),
}


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--only", default="fwd,dgrad,wgrad")
    ap.add_argument("--shapes", default=",".join(k for k in SHAPES if not k.startswith("X")))
    ap.add_argument("--batch", type=int, default=0)
    args = ap.parse_args()
    dt = ops.torch_dtype(args.dtype)
    only = args.only.split(",")
    print(f"{'layer':5s} {'shape':28s} " + " ".join(f"{k:>18s}" for k in only))
    for name in args.shapes.split(","):
        B, H, W, Cin, split, Cout = SHAPES[name]
        if args.batch:
            B = args.batch
        cp = ops.pad_ch(split) + ops.pad_ch(Cin - split) if split else ops.pad_ch(Cin)
        x = (torch.rand((B, H, W, cp), device="cuda") - 0.3).to(dt)
        w = (torch.rand((Cout, Cin, 3, 3), device="cuda") - 0.5) * 0.1
        b = torch.rand(Cout, device="cuda") - 0.5
        wf, wb = ops.pack_conv3x3(w, dt, split=split)
        y = ops.conv3x3_relu_fwd(x, wf, b, Cin, Cout, split=split)
        dy = (torch.rand_like(y.float()) - 0.5).to(dt)
        fl = 2.0 * B * H * W * Cout * Cin * 9
        res = []
        for k in only:
            if k == "fwd":
                out = torch.empty_like(y)
                us = timeit(lambda: ops.conv3x3_relu_fwd(x, wf, b, Cin, Cout, split=split, out=out), args.iters)
            elif k == "dgrad":
                us = timeit(lambda: ops.conv3x3_dgrad(dy, wb, Cin, Cout, split=split, mask=x), args.iters)
            else:
                dW = torch.empty((Cout, Cin, 3, 3), device="cuda"); db = torch.empty(Cout, device="cuda")
                us = timeit(lambda: ops.conv3x3_wgrad(x, dy, Cin, Cout, split=split, dW=dW, db=db), args.iters)
            res.append(f"{us:8.1f}us {fl / us / 1e6:6.0f}TF")
        print(f"{name:5s} {str((B, H, W, Cin, Cout)):28s} " + " ".join(f"{r:>18s}" for r in res))


if __name__ == "__main__":
    main()
