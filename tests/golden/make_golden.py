#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run in the build container only (the reference tree does not exist on the GPU box):

    cd /tmp && python /root/repo/tests/golden/make_golden.py

It imports ``/root/reference/models`` (the reference's own ``build_model_from_cfg`` ->
``models.cells.UNet``, byte-identical to ``models/unet2d.py:8-102``), runs it on seeded
synthetic inputs on the CPU, and stores only DATA: inputs, parameters, activations,
gradients and loss curves.  No reference source text is stored.

Files written next to this script:
  small4.npz     4-level, reduced widths, 16x16, full tensors
  small5.npz     5-level, 24->12 channels, 32x32, full tensors
  odd.npz        27x24 -> out_size 15x24 (both bilinear branches fire), full tensors
  cfg1_full.npz  default widths, 12ch 16x16 B=32: norms, samples and 20-step loss curve
  cfg2_b32.npz   default widths, 12ch 32x32 B=32: same
  cfg4_b2.npz    5-level default widths 24->12ch 128x128 B=2: same (8 steps)
  pool_ties.npz  2x2 max-pool known answers incl. tied windows
  curve240.npz   reduced widths, 12ch 16x16, four batches of 16 in rotation, 240 Adam steps at lr 1e-3:
                 the loss of every step (a curve that actually descends) and the final test sums

Name cases on the command line to regenerate only those (default: all).
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
from models import build_model_from_cfg  # noqa: E402  (reference factory, models/builders.py:107)

torch.set_num_threads(8)


def make_cfg(cin, cout, features, out_size=None):
    cfg = types.SimpleNamespace(model_type="unet", in_channels=cin, out_channels=cout,
                                unet_features=tuple(features))
    if out_size is not None:
        cfg.H_out, cfg.W_out = out_size
    return cfg


def synthetic(batch, cin, cout, h, w, oh=None, ow=None, seed=1):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(batch, cin, h, w, generator=g)
    Y = torch.rand(batch, cout, oh or h, ow or w, generator=g)
    return X, Y


def instrument(model):
    """Hooks that record activations and the gradients reaching them."""
    acts, grads, handles = {}, {}, []
    counters = {"pool": 0}

    def rec(name):
        def hook(_m, _inp, out):
            acts[name] = out.detach().clone()
            if out.requires_grad:
                out.register_hook(lambda g, n=name: grads.__setitem__(n, g.detach().clone()))
        return hook

    def pool_hook(_m, _inp, out):
        name = f"pool.{counters['pool']}"
        counters["pool"] += 1
        acts[name] = out.detach().clone()
        if out.requires_grad:
            out.register_hook(lambda g, n=name: grads.__setitem__(n, g.detach().clone()))

    for i, blk in enumerate(model.downs):
        handles.append(blk[1].register_forward_hook(rec(f"downs.{i}.1")))
        handles.append(blk[3].register_forward_hook(rec(f"downs.{i}.3")))
    handles.append(model.pool.register_forward_hook(pool_hook))
    handles.append(model.bottleneck[1].register_forward_hook(rec("bottleneck.1")))
    handles.append(model.bottleneck[3].register_forward_hook(rec("bottleneck.3")))
    for idx in range(0, len(model.ups), 2):
        j = idx // 2
        handles.append(model.ups[idx].register_forward_hook(rec(f"ups.{idx}")))

        def pre(_m, inp, j=j, idx=idx):
            cat = inp[0]
            up = acts[f"ups.{idx}"]
            if cat.shape[2:] != up.shape[2:]:
                f = up.shape[1]
                acts[f"align.{j}"] = cat[:, f:].detach().clone()
        handles.append(model.ups[idx + 1].register_forward_pre_hook(pre))
        handles.append(model.ups[idx + 1][1].register_forward_hook(rec(f"ups.{idx + 1}.1")))
        handles.append(model.ups[idx + 1][3].register_forward_hook(rec(f"ups.{idx + 1}.3")))
    handles.append(model.final_conv.register_forward_hook(rec("final_conv")))

    def reset():
        acts.clear(); grads.clear(); counters["pool"] = 0

    def remove():
        for h in handles:
            h.remove()
    return acts, grads, reset, remove


def run_case(name, cin, cout, features, h, w, batch, out_size=None, steps=20, full=True, seed_model=0):
    torch.manual_seed(seed_model)
    model = build_model_from_cfg(make_cfg(cin, cout, features, out_size))
    oh, ow = out_size if out_size is not None else (h, w)
    X, Y = synthetic(batch, cin, cout, h, w, oh, ow)
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}

    acts, grads, reset, remove = instrument(model)
    Xg = X.clone().requires_grad_(True)
    pred = model(Xg)
    loss = torch.nn.MSELoss()(pred, Y)
    model.zero_grad()
    loss.backward()
    pgrads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    out = {"meta.cin": cin, "meta.cout": cout, "meta.features": np.array(features),
           "meta.out_size": np.array(out_size if out_size else (-1, -1)),
           "meta.batch": batch, "meta.hw": np.array((h, w)),
           "loss0": np.float32(loss.item()),
           "Xsum": np.float64(X.double().sum().item()), "Ysum": np.float64(Y.double().sum().item())}
    if full:
        out["X"] = X.numpy()
        out["Y"] = Y.numpy()
        out["pred"] = pred.detach().numpy()
        for k, v in init.items():
            out["param." + k] = v.numpy()
        for k, v in acts.items():
            out["act." + k] = v.numpy()
        for k, v in grads.items():
            out["actgrad." + k] = v.numpy()
        for k, v in pgrads.items():
            out["grad." + k] = v.numpy()
        out["dX"] = Xg.grad.detach().numpy()
    else:
        # scalars only: the full parameter set is tens of MB.  Parameters are re-created
        # from the seed by the oracle (bit-identical init is itself a tested property).
        idx = torch.Generator().manual_seed(7)
        flat = pred.detach().flatten()
        pick = torch.randint(0, flat.numel(), (64,), generator=idx)
        out["predidx"] = pick.numpy()
        out["predval"] = flat[pick].numpy()
        out["prednorm"] = np.float64(pred.detach().double().norm().item())
        for k, v in init.items():
            out["paramnorm." + k] = np.float64(v.double().norm().item())
        for k, v in acts.items():
            out["actnorm." + k] = np.float64(v.double().norm().item())
            flat = v.flatten()
            pick = torch.randint(0, flat.numel(), (16,), generator=idx)
            out["actidx." + k] = pick.numpy()
            out["actval." + k] = flat[pick].numpy()
        for k, v in grads.items():
            out["actgradnorm." + k] = np.float64(v.double().norm().item())
        for k, v in pgrads.items():
            out["gradnorm." + k] = np.float64(v.double().norm().item())
            flat = v.flatten()
            pick = torch.randint(0, flat.numel(), (16,), generator=idx)
            out["gradidx." + k] = pick.numpy()
            out["gradval." + k] = flat[pick].numpy()
        out["dXnorm"] = np.float64(Xg.grad.double().norm().item())
    remove()

    # N-step Adam loss curve: the loop body of train.py:382-393 on one fixed batch.
    torch.manual_seed(seed_model)
    model = build_model_from_cfg(make_cfg(cin, cout, features, out_size))
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)  # train.py:497, cfg.lr (config.py:55)
    crit = torch.nn.MSELoss()                            # train.py:498
    losses = []
    model.train()
    for _ in range(steps):
        p = model(X)
        l = crit(p, Y)
        opt.zero_grad()
        l.backward()
        opt.step()
        losses.append(l.item())
    out["losses"] = np.array(losses, dtype=np.float64)
    model.eval()
    with torch.no_grad():
        d = model(X) - Y
        out["eval.abs_sum"] = np.float64(d.abs().sum().item())   # test.py:41
        out["eval.sq_sum"] = np.float64((d ** 2).sum().item())   # test.py:42
        out["eval.numel"] = np.int64(d.numel())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: loss0={loss.item():.8f} losses[-1]={losses[-1]:.8f} "
          f"acts={len(acts)} grads={len(grads)}")


def pool_ties():
    pool = torch.nn.MaxPool2d(kernel_size=2, stride=2)  # models/unet2d.py:22
    g = torch.Generator().manual_seed(3)
    x = torch.zeros(1, 2, 6, 6)
    x[0, 0, 0:2, 0:2] = 0.0                       # all-zero window
    x[0, 0, 0:2, 2:4] = torch.tensor([[1., 1.], [1., 1.]])     # 4-way positive tie
    x[0, 0, 0:2, 4:6] = torch.tensor([[0., 2.], [2., 0.]])     # 2-way tie, anti-diagonal
    x[0, 0, 2:4, 0:2] = torch.tensor([[0., 0.], [3., 3.]])     # tie on bottom row
    x[0, 0, 2:4, 2:4] = torch.tensor([[0., 5.], [0., 5.]])     # tie in right column
    x[0, 0, 4:6, :] = torch.rand(2, 6, generator=g)
    x[0, 1] = torch.relu(torch.rand(6, 6, generator=g) - 0.5)   # post-ReLU-like: many zeros
    xo = torch.rand(2, 3, 7, 5, generator=g)                    # odd size: floor drops last row/col
    out = {}
    for name, t in (("even", x), ("odd", xo)):
        t = t.clone().requires_grad_(True)
        y = pool(t)
        gy = torch.rand(y.shape, generator=g)
        y.backward(gy)
        out[name + ".x"] = t.detach().numpy()
        out[name + ".y"] = y.detach().numpy()
        out[name + ".gy"] = gy.numpy()
        out[name + ".gx"] = t.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "pool_ties.npz"), **out)
    print("pool_ties: all-zero window grad =", out["even.gx"][0, 0, 0:2, 0:2].tolist())


def curve_case(name, cin, cout, features, h, w, batch, n_batches, steps, lr, seed_model=0):
    """The loop of train.py:382-393 over several batches in rotation, long enough and with a learning
    rate large enough for the loss to fall by a large factor: every step's loss, and the sums of
    test.py:39-43 over all batches at the end."""
    torch.manual_seed(seed_model)
    model = build_model_from_cfg(make_cfg(cin, cout, features))
    batches = [synthetic(batch, cin, cout, h, w, seed=100 + i) for i in range(n_batches)]
    # a learnable target: a fixed smooth function of the input, so the curve has somewhere to go
    batches = [(X, X[:, :cout] * 0.5 + 0.25 * X[:, -cout:].flip(-1)) for X, _ in batches]
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    crit = torch.nn.MSELoss()
    losses = []
    model.train()
    for s in range(steps):
        X, Y = batches[s % n_batches]
        l = crit(model(X), Y)
        opt.zero_grad()
        l.backward()
        opt.step()
        losses.append(l.item())
    model.eval()
    abs_sum = sq_sum = 0.0
    with torch.no_grad():
        for X, Y in batches:
            d = model(X) - Y
            abs_sum += d.abs().sum().item()
            sq_sum += (d ** 2).sum().item()
    out = {"meta.cin": cin, "meta.cout": cout, "meta.features": np.array(features), "meta.out_size": np.array((-1, -1)),
           "meta.batch": batch, "meta.hw": np.array((h, w)), "meta.n_batches": n_batches, "meta.lr": np.float64(lr),
           "losses": np.array(losses, dtype=np.float64), "eval.abs_sum": np.float64(abs_sum),
           "eval.sq_sum": np.float64(sq_sum)}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: loss[0]={losses[0]:.6f} loss[{steps // 2}]={losses[steps // 2]:.6f} loss[-1]={losses[-1]:.6f}")


CASES = {
    "small4": lambda: run_case("small4", 12, 1, (4, 8, 16, 32), 16, 16, batch=2),
    "small5": lambda: run_case("small5", 24, 12, (4, 8, 16, 32, 64), 32, 32, batch=2),
    "odd": lambda: run_case("odd", 3, 1, (4, 8, 16, 32), 27, 24, batch=2, out_size=(15, 24)),
    "cfg1_full": lambda: run_case("cfg1_full", 12, 1, (32, 64, 128, 256), 16, 16, batch=32, full=False),
    "cfg2_b32": lambda: run_case("cfg2_b32", 12, 1, (32, 64, 128, 256), 32, 32, batch=32, full=False),
    "cfg4_b2": lambda: run_case("cfg4_b2", 24, 12, (32, 64, 128, 256, 512), 128, 128, batch=2, full=False, steps=8),
    "pool_ties": pool_ties,
    "curve240": lambda: curve_case("curve240", 12, 1, (8, 16, 32, 64), 16, 16, batch=16, n_batches=4, steps=240, lr=1e-3),
}

if __name__ == "__main__":
    for case in (sys.argv[1:] or list(CASES)):
        CASES[case]()
