"""One case of the whole-network sweep in detail: every activation gradient against the oracle's, with the number of
elements that differ by more than 1e-3 of the tensor's largest magnitude (a flipped ReLU shows as a few isolated
elements; a kernel error as many).   python scripts/probes/fuzz_net_debug.py dtype cin cout f0,f1,.. B H W state_seed data_seed"""
import os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root)
import torch
import unet_amd as ua
from oracle import unet_oracle as O
from test_gpu_unet import rel

dtype, cin, cout = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
feats = tuple(int(v) for v in sys.argv[4].split(","))
B, H, W, s1, s2 = (int(v) for v in sys.argv[5:10])
state = O.init_state_dict(cin, cout, feats, seed=s1)
model = ua.UNet(cin, cout, feats, compute_dtype=dtype); model.load_state_dict(state); model = model.to("cuda")
X, Y = O.synthetic_batch(B, cin, cout, H, W, seed=s2)
loss_ref, taps, agr, pgrads, dX = O.forward_backward(state, X, Y, None, want_input_grad=True)
Xg = X.cuda().requires_grad_(True)
pred = model(Xg); ua.MSELoss()(pred, Y.cuda()).backward()
names = model.last_plan.tensors()
print("kernels:", sorted(set(v.get("kernel", "?") for v in names.values() if isinstance(v, dict) and "kernel" in v)))
for k, ref in taps.items():
    if "act." + k in names:
        got = model.workspace_tensor("act." + k).cpu()
        d = (got - ref).abs()
        print("act  %-16s rel %.2e  max|d| %.2e  n(|d|>1e-3 max) %d of %d" % (k, rel(got, ref), d.max().item(), int((d > 1e-3 * ref.abs().max()).sum()), ref.numel()))
for k, ref in agr.items():
    if "grad." + k in names:
        post_relu = not k.startswith("pool.") and (k.endswith(".1") or k.endswith(".3"))
        r = ref * (taps[k] > 0) if post_relu else ref
        got = model.workspace_tensor("grad." + k).cpu()
        d = (got - r).abs()
        print("grad %-16s rel %.2e  max|d| %.2e  n(|d|>1e-3 max) %d of %d" % (k, rel(got, r), d.max().item(), int((d > 1e-3 * r.abs().max()).sum()), r.numel()))
for k, p in model.named_parameters():
    print("param %-22s rel %.2e" % (k, rel(p.grad.cpu(), pgrads[k])))
