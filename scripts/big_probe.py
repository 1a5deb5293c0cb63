"""Large-size consistency probe: a 16384-image batch made of 32 copies of one 512-image batch must
predict the same rows (to rounding: kernel choice depends on the batch) and give the same mean-loss gradients as the 512-image batch alone; 256-wide
images (beyond the halo-tile kernels) against the CPU oracle."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from oracle import unet_oracle as O
rel = lambda a, b: ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()
for dtype in ("bf16", "fp32"):
    torch.manual_seed(0)
    m = unet_amd.UNet(12, 1, (32, 64, 128, 256), compute_dtype=dtype).cuda()
    X, Y = (t.cuda() for t in O.synthetic_batch(512, 12, 1, 32, 32))
    crit = unet_amd.MSELoss()
    p1 = m(X); crit(p1, Y).backward()
    g1 = [p.grad.clone() for p in m.parameters()]
    for p in m.parameters(): p.grad = None
    reps = 32 if dtype == "bf16" else 8
    Xb, Yb = X.repeat(reps, 1, 1, 1), Y.repeat(reps, 1, 1, 1)
    pb = m(Xb); crit(pb, Yb).backward()
    gb = [p.grad for p in m.parameters()]
    torch.cuda.synchronize()
    print(dtype, "batch", Xb.shape[0], "pred rows rel:", rel(pb[:512], p1.detach()), rel(pb[-512:], p1.detach()),
          "worst grad rel:", max(rel(a, b) for a, b in zip(gb, g1)), flush=True)
    del m, Xb, Yb, pb, gb
    torch.cuda.empty_cache()
for (B, H, W) in ((2, 256, 256), (1, 96, 200)):
    state = O.init_state_dict(3, 2, (8, 16, 32), seed=1)
    m = unet_amd.UNet(3, 2, (8, 16, 32), compute_dtype="fp32"); m.load_state_dict(state); m = m.cuda()
    X, Y = O.synthetic_batch(B, 3, 2, H, W, seed=2)
    loss_ref, taps, _, pgrads = O.forward_backward(state, X, Y)
    pred = m(X.cuda()); unet_amd.MSELoss()(pred, Y.cuda()).backward()
    print("fp32", (B, H, W), "out rel:", rel(pred.detach().cpu(), taps["out"]),
          "worst grad rel:", max(rel(p.grad.cpu(), pgrads[k]) for k, p in m.named_parameters()), flush=True)
