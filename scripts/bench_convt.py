"""Micro-benchmark of the transposed-conv layers through the C ABI (development aid).

    python scripts/bench_convt.py [--iters 50] [--batch 512]

Shapes are the four ConvTranspose2d layers of the default workload (batch 512, 32x32): the output is
written into the upper channel half of a concat buffer, as the network does."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import unet_amd
from unet_amd import ops
from bench_conv import timeit

SHAPES = {"ups.0": (2, 512, 256), "ups.2": (4, 256, 128), "ups.4": (8, 128, 64), "ups.6": (16, 64, 32)}  # h=w, Cin, Cout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--batch", type=int, default=512)
    args = ap.parse_args()
    dt = torch.bfloat16
    B = args.batch
    print(f"{'layer':6s} {'fwd':>10s} {'dgrad':>10s} {'wgrad':>10s}")
    for name, (h, Cin, Cout) in SHAPES.items():
        x = (torch.rand((B, h, h, Cin), device="cuda") - 0.3).to(dt)
        w = (torch.rand((Cin, Cout, 2, 2), device="cuda") - 0.5) * 0.1
        b = torch.rand(Cout, device="cuda") - 0.5
        wf, wb = ops.pack_convt2x2(w, dt)
        cat = torch.zeros((B, 2 * h, 2 * h, 2 * ops.pad_ch(Cout)), device="cuda", dtype=dt)
        dcat = (torch.rand_like(cat.float()) - 0.5).to(dt)
        off = ops.pad_ch(Cout)
        t_f = timeit(lambda: ops.convt2x2_fwd(x, wf, b, Cin, Cout, out=cat, out_coff=off), args.iters)
        t_d = timeit(lambda: ops.convt2x2_dgrad(dcat, wb, Cin, Cout, mask=x, dy_coff=off), args.iters)
        t_w = timeit(lambda: ops.convt2x2_wgrad(x, dcat, Cin, Cout, dy_coff=off), args.iters)
        print(f"{name:6s} {t_f:8.1f}us {t_d:8.1f}us {t_w:8.1f}us")


if __name__ == "__main__":
    main()
