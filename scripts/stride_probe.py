"""Development probe: cost of reading a 32-channel slice of 64-channel NHWC rows (half cache lines)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_amd
from unet_amd import ops

def timeit(fn, iters=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

B = 512
for cs in (32, 64):
    x = torch.rand((B, 32, 32, cs), device="cuda").to(torch.bfloat16)
    dyp = torch.rand((B, 16, 16, 32), device="cuda").to(torch.bfloat16)
    sk = torch.rand((B, 32, 32, cs), device="cuda").to(torch.bfloat16)
    print(f"cs={cs}: pool fwd {timeit(lambda: ops.maxpool2_fwd(x, 32)):6.1f} us   "
          f"pool bwd+skip {timeit(lambda: ops.maxpool2_bwd(x, dyp, 32, dskip=sk, relu_mask=True)):6.1f} us")
