#!/usr/bin/env python3
"""Micro-benchmark of the device input pipeline (development aid): time per batch of mi_window_batch
and its rate against the bytes it must write (B*(K*H*W + H*W) floats)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import unet_amd


def main():
    rng = np.random.RandomState(0)
    for (hw, K, B, stride) in [(32, 12, 512, 1), (32, 12, 512, 64), (128, 24, 64, 16), (24, 1, 128, 1)]:
        n = 2 * hw * hw + stride * 20000
        sw = unet_amd.SeriesWindows(rng.rand(n) * 1e4, hw, hw, hw, hw, 0, stride, context=K)
        ids = sw.to_device_ids(rng.randint(K - 1, sw.num_windows, size=B))
        out = sw.batch(ids)
        for _ in range(5):
            sw.batch(ids, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 200
        e0.record()
        for _ in range(iters):
            sw.batch(ids, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        nbytes = 4.0 * B * (K + 1) * hw * hw
        print(f"{hw}x{hw} K={K} B={B} stride={stride}: {us:7.1f} us/batch  {nbytes / us / 1e3:7.1f} GB/s written "
              f"({nbytes / 1e6:.1f} MB), host copy of the same batch over PCIe ~{nbytes / 50e3:.0f} us")


if __name__ == "__main__":
    main()
