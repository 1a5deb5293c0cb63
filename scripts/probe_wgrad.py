"""Times the 3x3 weight gradient of the deep layers alone (kernel + combine), for kernel experiments."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd import ops
shapes = [("ups.1.0", 512, 512, 256, 4, 4, 256), ("ups.1.2", 512, 256, 256, 4, 4, 0), ("bott.0", 512, 256, 512, 2, 2, 0),
          ("bott.2", 512, 512, 512, 2, 2, 0), ("downs.3.0", 512, 128, 256, 4, 4, 0), ("downs.3.2", 512, 256, 256, 4, 4, 0)]
dt = torch.bfloat16
for name, B, Cin, Cout, H, W, split in shapes:
    cp = ops.pad_ch(Cin) if not split else ops.pad_ch(split) + ops.pad_ch(Cin - split)
    x = torch.randn(B, H, W, cp, device="cuda").to(dt)
    dy = torch.randn(B, H, W, ops.pad_ch(Cout), device="cuda").to(dt)
    dW = torch.empty(Cout, Cin, 3, 3, device="cuda"); db = torch.empty(Cout, device="cuda")
    for _ in range(5):
        ops.conv3x3_wgrad(x, dy, Cin, Cout, split, dW, db)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.conv3x3_wgrad(x, dy, Cin, Cout, split, dW, db)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:10s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us (kernel + combine)", flush=True)
