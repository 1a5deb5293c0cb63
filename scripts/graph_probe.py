"""Development probe: eager launches vs one captured HIP graph for forward+loss+backward.

    python scripts/graph_probe.py [--batch 512] [--iters 50]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import unet_amd
from oracle import unet_oracle as O


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = unet_amd.UNet(12, 1, (32, 64, 128, 256), compute_dtype="bf16").to(dev)
    X, Y = O.synthetic_batch(args.batch, 12, 1, 32, 32)
    X, Y = X.to(dev), Y.to(dev)
    crit = unet_amd.MSELoss()
    params = list(model.parameters())

    def fwd_bwd():
        for p in params:
            p.grad = None
        loss = crit(model(X), Y)
        loss.backward()
        return loss

    def timeit(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.iters * 1e3

    def fwd_only():
        with torch.no_grad():
            model(X)

    print(f"eager fwd only     {timeit(fwd_only):7.3f} ms")
    print(f"eager fwd+bwd      {timeit(fwd_bwd):7.3f} ms")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fwd_bwd()
    print(f"graph fwd+bwd      {timeit(g.replay):7.3f} ms")


if __name__ == "__main__":
    main()
