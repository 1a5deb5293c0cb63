"""Development aid: host (enqueue) time of each piece of a training step at the benched size.
The device is drained every few steps so that a full launch queue never blocks the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_amd
from oracle import unet_oracle as O

torch.manual_seed(0)
m = unet_amd.UNet(12, 1, (32, 64, 128, 256), compute_dtype="bf16").cuda()
X, Y = (t.cuda() for t in O.synthetic_batch(512, 12, 1, 32, 32))
opt, crit = unet_amd.FusedAdam(m.parameters(), lr=1e-4, model=m), unet_amd.MSELoss()
dp = os.environ.get("UNET_TM_FORCE_DP") == "1"
if dp:
    unet_amd.init_process_group_from_env()
    unet_amd.GradientReducer(force=True).attach(m)
acc = {}
def tick(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
for it in range(64):
    if it == 14: acc.clear()
    if it % 4 == 0: torch.cuda.synchronize()
    t = time.perf_counter()
    pred = m(X); t = tick("forward", t)
    loss = crit(pred, Y); t = tick("loss", t)
    opt.zero_grad(); t = tick("zero_grad", t)
    loss.backward(); t = tick("backward", t)
    opt.step(); t = tick("step", t)
torch.cuda.synchronize()
n = 50
print("dp" if dp else "single", {k: round(v / n * 1e6, 1) for k, v in acc.items()}, "total us", round(sum(acc.values()) / n * 1e6, 1))
