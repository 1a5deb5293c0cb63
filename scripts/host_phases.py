"""Host time per phase of the reference loop body (no device sync inside the loop): where the ~0.9 ms of
enqueue time per training step goes.   python scripts/host_phases.py [steps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_amd

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
model = unet_amd.UNet(12, 1, (32, 64, 128, 256), compute_dtype="bf16").to(dev)
opt, crit = unet_amd.FusedAdam(model.parameters(), lr=1e-4, model=model), unet_amd.MSELoss()
X = torch.randn(512, 12, 32, 32, device=dev)
Y = torch.randn(512, 1, 32, 32, device=dev)
acc = {k: 0.0 for k in ("forward", "loss", "zero_grad", "backward", "step", "add")}
loss_acc = torch.zeros((), device=dev)
def one(record):
    t = [time.perf_counter()]
    pred = model(X); t.append(time.perf_counter())
    loss = crit(pred, Y); t.append(time.perf_counter())
    opt.zero_grad(); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    loss_acc.add_(loss.detach()); t.append(time.perf_counter())
    if record:
        for k, a, b in zip(acc, t, t[1:]):
            acc[k] += b - a
for _ in range(30): one(False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    one(True)
    if i % 8 == 7: torch.cuda.synchronize()      # keep the queue short: the host never blocks on a full queue
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("host us per step:", {k: round(v / steps * 1e6, 1) for k, v in acc.items()}, "sum", round(sum(acc.values()) / steps * 1e6, 1),
      "| wall per step (with a sync every 8 steps)", round(wall / steps * 1e6, 1))

# ---- the library calls alone (host time inside the C entry points)
import ctypes
L = unet_amd.lib()
calls = {}
def wrap(name):
    f = getattr(L, name)
    def g(*a):
        t0 = time.perf_counter()
        r = f(*a)
        d = calls.setdefault(name, [0.0, 0])
        d[0] += time.perf_counter() - t0
        d[1] += 1
        return r
    setattr(L, name, g)
for n in ("mi_unet_forward", "mi_unet_backward", "mi_mse_fwd_bwd", "mi_unet_adam_pack", "mi_adam_step"):
    if hasattr(L, n):
        wrap(n)
for i in range(steps):
    one(False)
    if i % 8 == 7: torch.cuda.synchronize()
torch.cuda.synchronize()
print("host us per step inside library calls:", {k: (round(v[0] / steps * 1e6, 1), v[1] // steps) for k, v in calls.items()})
