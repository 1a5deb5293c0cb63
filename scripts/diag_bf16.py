"""Diagnostic: per-layer error of the bf16 HIP path vs the fp32 fixture, next to the error of
torch CPU autocast(bf16) of the oracle on the same fixture (a yardstick for bf16 noise)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from collections import OrderedDict
import unet_amd as ua
from oracle import unet_oracle as O

name = sys.argv[1] if len(sys.argv) > 1 else "small5"
g = np.load(f"tests/golden/{name}.npz")
feats = tuple(int(v) for v in g["meta.features"]); cin = int(g["meta.cin"]); cout = int(g["meta.cout"])
osz = tuple(int(v) for v in g["meta.out_size"]); osz = None if osz[0] < 0 else osz
state = OrderedDict((k[6:], torch.from_numpy(g[k])) for k in g.files if k.startswith("param."))
X, Y = torch.from_numpy(g["X"]), torch.from_numpy(g["Y"])
rel = lambda a, b: ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()

# yardstick: CPU autocast
params = {k: v.clone().requires_grad_(True) for k, v in state.items()}
taps = {}
with torch.autocast("cpu", dtype=torch.bfloat16):
    pred = O.unet_forward(params, X, osz, taps)
loss = O.mse_loss(pred.float(), Y)
names = [n for n in taps if n not in ("out",)]
gr = torch.autograd.grad(loss, [taps[n] for n in names] + list(params.values()), allow_unused=True)
ac_actgrad = {n: gg for n, gg in zip(names, gr[:len(names)])}
ac_pgrad = {k: gg for k, gg in zip(params.keys(), gr[len(names):])}

model = ua.UNet(cin, cout, feats, out_size=osz, compute_dtype="bf16")
model.load_state_dict(state); model = model.cuda()
Xg = X.cuda().requires_grad_(True)
p = model(Xg); l = ua.MSELoss()(p, Y.cuda()); l.backward(); torch.cuda.synchronize()
print(f"{'tensor':22s} {'hip act':>9s} {'ac act':>9s} {'hip grad':>9s} {'ac grad':>9s}")
for k in g.files:
    if not k.startswith("actgrad."): continue
    n = k[8:]
    if n == "final_conv": continue
    ref_a = torch.from_numpy(g["act." + n]); ref_g = torch.from_numpy(g[k])
    post = not n.startswith("pool.") and (n.endswith(".1") or n.endswith(".3"))
    hip_a = model.workspace_tensor("act." + n).cpu()
    hip_g = model.workspace_tensor("grad." + n).cpu()
    rg = ref_g * (ref_a > 0) if post else ref_g
    acg = ac_actgrad[n].float()
    acg = acg * (taps[n].float() > 0) if post else acg
    print(f"{n:22s} {rel(hip_a, ref_a):9.2e} {rel(taps[n].float(), ref_a):9.2e} {rel(hip_g, rg):9.2e} {rel(acg, rg):9.2e}")
print("param grads: hip vs autocast (relL2 to fp32 fixture)")
eh, ea = [], []
for k, pp in model.named_parameters():
    ref = torch.from_numpy(g["grad." + k])
    eh.append(rel(pp.grad.cpu(), ref)); ea.append(rel(ac_pgrad[k].float(), ref))
    print(f"{k:26s} {eh[-1]:9.2e} {ea[-1]:9.2e}")
print("median hip %.3e autocast %.3e ; worst hip %.3e autocast %.3e" % (np.median(eh), np.median(ea), max(eh), max(ea)))
