"""Context measurement (not part of the product or its tests): the same U-Net built from stock
torch.nn layers, i.e. what PyTorch-ROCm + MIOpen/hipBLASLt does for this training step on the box.

    python scripts/stock_torch_gpu.py [--batch 512] [--steps 20] [--dtype bf16|fp32]
"""
import argparse
import time

import torch
import torch.nn as nn


class StockUNet(nn.Module):
    def __init__(self, cin=12, cout=1, feats=(32, 64, 128, 256)):
        super().__init__()
        def block(a, b):
            return nn.Sequential(nn.Conv2d(a, b, 3, padding=1), nn.ReLU(inplace=True),
                                 nn.Conv2d(b, b, 3, padding=1), nn.ReLU(inplace=True))
        self.downs, self.ups = nn.ModuleList(), nn.ModuleList()
        ch = cin
        for f in feats:
            self.downs.append(block(ch, f)); ch = f
        self.bottleneck = block(feats[-1], 2 * feats[-1])
        for f in reversed(feats):
            self.ups.append(nn.ConvTranspose2d(2 * f, f, 2, 2)); self.ups.append(block(2 * f, f))
        self.final = nn.Conv2d(feats[0], cout, 1)
        self.pool = nn.MaxPool2d(2, 2)

    def forward(self, x):
        skips = []
        for d in self.downs:
            x = d(x); skips.append(x); x = self.pool(x)
        x = self.bottleneck(x)
        for i in range(0, len(self.ups), 2):
            x = self.ups[i](x)
            x = self.ups[i + 1](torch.cat([skips[-1 - i // 2], x], dim=1))
        return self.final(x)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--channels-last", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(0)
    m = StockUNet().cuda()
    X, Y = torch.rand(a.batch, 12, 32, 32, device="cuda"), torch.rand(a.batch, 1, 32, 32, device="cuda")
    if a.channels_last:
        m = m.to(memory_format=torch.channels_last); X = X.contiguous(memory_format=torch.channels_last)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)
    crit = nn.MSELoss()

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.dtype == "bf16"):
            loss = crit(m(X).float(), Y)
        opt.zero_grad(); loss.backward(); opt.step()

    for _ in range(5):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    print(f"stock torch {a.dtype} channels_last={a.channels_last} batch {a.batch}: {dt * 1e3:.2f} ms/step, {a.batch / dt:.0f} TMs/s", flush=True)


if __name__ == "__main__":
    main()
