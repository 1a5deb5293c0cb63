"""Autograd-semantics probe: several forward passes before backward, accumulation, retained graphs."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
torch.manual_seed(0)
m = unet_amd.UNet(3, 2, features=(8, 16), compute_dtype="fp32").cuda()
x1 = torch.randn(4, 3, 16, 16, device="cuda"); x2 = torch.randn(4, 3, 16, 16, device="cuda")
def grads(xs, mode):
    for p in m.parameters(): p.grad = None
    if mode == "sum_loss":
        l = sum((m(x) ** 2).mean() for x in xs); l.backward()
    elif mode == "accumulate":
        for x in xs: (m(x) ** 2).mean().backward()
    elif mode == "zero_inplace":
        (m(xs[0]) ** 2).mean().backward()
        for p in m.parameters(): p.grad.zero_()
        for x in xs: (m(x) ** 2).mean().backward()
    elif mode == "eval_between":
        l = (m(xs[0]) ** 2).mean()
        with torch.no_grad(): m(xs[1])
        l.backward()
        (m(xs[1]) ** 2).mean().backward()
    elif mode == "retain":
        l = (m(xs[0]) ** 2).mean() + (m(xs[1]) ** 2).mean()
        l.backward(retain_graph=True)
        for p in m.parameters(): p.grad = None
        l.backward()
    return [p.grad.clone() for p in m.parameters()]
g1 = grads([x1], "accumulate"); g2 = grads([x2], "accumulate")
ref = [a + b for a, b in zip(g1, g2)]
for mode in ("accumulate", "zero_inplace", "sum_loss", "eval_between", "retain"):
    g = grads([x1, x2], mode)
    err = max(((a - b).norm() / (b.norm() + 1e-30)).item() for a, b in zip(g, ref))
    print(mode, "max rel err vs g1+g2:", err, "plans:", sum(len(v) for v in m._plans.values()))
# stale graph: backward after the workspace was reused must raise
l = (m(x1) ** 2).mean(); l.backward(retain_graph=True)
(m(x2) ** 2).mean().backward()
try:
    l.backward(); print("stale backward: NO ERROR")
except RuntimeError as e:
    print("stale backward raises:", str(e)[:60])
