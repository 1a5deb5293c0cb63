"""Development aid: host time of the pieces of backward() on the data-parallel path (one-rank RCCL
group): the library's launch calls, the collective calls and the waits, timed by wrapping them."""
import os, sys, time
os.environ["UNET_TM_FORCE_DP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_amd
from oracle import unet_oracle as O

torch.manual_seed(0)
m = unet_amd.UNet(12, 1, (32, 64, 128, 256), compute_dtype="bf16").cuda()
X, Y = (t.cuda() for t in O.synthetic_batch(512, 12, 1, 32, 32))
opt, crit = unet_amd.FusedAdam(m.parameters(), lr=1e-4, model=m), unet_amd.MSELoss()
unet_amd.init_process_group_from_env()
red = unet_amd.GradientReducer(force=True).attach(m)
acc, cnt = {}, {}
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        cnt[name] = cnt.get(name, 0) + 1
        return r
    return w
L = unet_amd.lib()
for name in ("mi_unet_backward", "mi_unet_plan_side_stream", "mi_unet_plan_defer_join"):
    setattr(L, name, timed(name, getattr(L, name)))
red.reduce_async = timed("reduce_async", red.reduce_async)
red.wait_all = timed("wait_all", red.wait_all)
m._backward_impl = timed("_backward_impl", m._backward_impl)
tb = 0.0
for it in range(64):
    if it == 14: acc.clear(); cnt.clear(); tb = 0.0
    if it % 4 == 0: torch.cuda.synchronize()
    pred = m(X); loss = crit(pred, Y); opt.zero_grad()
    t0 = time.perf_counter(); loss.backward(); tb += time.perf_counter() - t0
    opt.step()
torch.cuda.synchronize()
print("backward host us/step:", round(tb / 50 * 1e6, 1))
for k in acc: print(f"  {k:28s} {acc[k] / 50 * 1e6:8.1f} us/step  ({cnt[k] / 50:.0f} calls)")
