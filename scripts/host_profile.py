"""Development aid: host time of the pieces of a training step (tiny batch, so the GPU never limits)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_amd
from oracle import unet_oracle as O

torch.manual_seed(0)
m = unet_amd.UNet(12, 1, (32, 64, 128, 256), compute_dtype="bf16").cuda()
X, Y = (t.cuda() for t in O.synthetic_batch(int(os.environ.get("B", "8")), 12, 1, 32, 32))
opt, crit = unet_amd.FusedAdam(m.parameters(), lr=1e-4, model=m), unet_amd.MSELoss()

def timeit(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt

def fwd_nograd():
    with torch.no_grad(): m(X)
def fwd_grad(): return m(X)
def fwd_loss(): return crit(m(X), Y)
def fwd_bwd():
    loss = crit(m(X), Y); opt.zero_grad(); loss.backward()
def full():
    loss = crit(m(X), Y); opt.zero_grad(); loss.backward(); opt.step()

for name, fn in [("forward no_grad", fwd_nograd), ("forward grad", fwd_grad), ("forward+loss", fwd_loss),
                 ("forward+loss+zero_grad+backward", fwd_bwd), ("full step", full)]:
    print(f"{name:34s} {timeit(fn):8.1f} us host", flush=True)

# ---- inside backward
plan = m._forward_impl(X, train=True)
dpred = torch.ones_like(m._last_pred)
def bwd_impl(): m._backward_impl(plan, dpred, None, 0.0)
def bwd_impl_views():
    g, _ = m._backward_impl(plan, dpred, None, 0.0)
    return tuple(g[o:o + n].view(shape) for (o, n, shape) in m._flat_layout)
def zero(): opt.zero_grad()
def mse_fb():
    p = m._last_pred.detach().requires_grad_(True)
    crit(p, Y).backward()
def optstep(): opt.step()
for name, fn in [("_backward_impl (C call + clone)", bwd_impl), ("_backward_impl + 46 views", bwd_impl_views), ("zero_grad", zero),
                 ("mse fwd+bwd via autograd", mse_fb), ("opt.step", optstep)]:
    print(f"{name:34s} {timeit(fn):8.1f} us host", flush=True)
