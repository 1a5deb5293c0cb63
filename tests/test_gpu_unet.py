"""Whole-network parity of the HIP U-Net against the golden fixtures recorded from the
reference and against the CPU oracle.

Tolerances (relative L2 unless stated), grounded in the measured fp32-vs-fp64 and
bf16-autocast-vs-fp64 noise floors of this network:
  fp32 kernels : layer outputs <= 1e-5, gradients <= 5e-5, 20-step loss curve rel <= 1e-5
  bf16 kernels : layer outputs <= 1e-2, gradients <= 5e-2 median / 1.5e-1 worst tensor,
                 20-step loss curve rel <= 2e-3
"""
import os
import types
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu

BAR = {
    "fp32": dict(out=1e-5, grad=5e-5, grad_med=5e-5, loss=1e-5),
    "bf16": dict(out=1e-2, grad=1.5e-1, grad_med=5e-2, loss=2e-3),
}


@pytest.fixture(scope="module")
def ua():
    import unet_amd
    unet_amd.lib()
    return unet_amd


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def meta(g):
    feats = tuple(int(v) for v in g["meta.features"])
    osz = tuple(int(v) for v in g["meta.out_size"])
    return int(g["meta.cin"]), int(g["meta.cout"]), feats, (None if osz[0] < 0 else osz)


def model_from_fixture(ua, g, dtype):
    cin, cout, feats, osz = meta(g)
    model = ua.UNet(cin, cout, feats, out_size=osz, compute_dtype=dtype)
    state = OrderedDict((k[6:], torch.from_numpy(g[k])) for k in g.files if k.startswith("param."))
    model.load_state_dict(state, strict=True)          # test.py:135 path
    return model.to("cuda"), state


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["small4", "small5", "odd"])
def test_per_layer_outputs_and_gradients(ua, golden, name, dtype):
    g = golden(name)
    bar = BAR[dtype]
    model, state = model_from_fixture(ua, g, dtype)
    X = torch.from_numpy(g["X"]).cuda().requires_grad_(True)
    Y = torch.from_numpy(g["Y"]).cuda()
    pred = model(X)
    loss = ua.MSELoss()(pred, Y)
    loss.backward()
    torch.cuda.synchronize()
    assert rel(pred.detach().cpu(), torch.from_numpy(g["pred"])) < bar["out"]
    assert abs(loss.item() - float(g["loss0"])) <= bar["out"] * abs(float(g["loss0"]))
    names = model.last_plan.tensors()
    n_act = n_grad = 0
    worst = {}
    for k in g.files:
        if k.startswith("act."):
            if k in names:
                e = rel(model.workspace_tensor(k).cpu(), torch.from_numpy(g[k]))
                assert e < bar["out"], (k, e)
                n_act += 1
            else:
                assert k == "act.final_conv" and "act.final_conv" not in names   # head output is `pred` itself
        elif k.startswith("actgrad."):
            lname = k[len("actgrad."):]
            ref = torch.from_numpy(g[k])
            if lname == "final_conv":
                continue
            post_relu = not lname.startswith("pool.") and (lname.endswith(".1") or lname.endswith(".3"))
            if post_relu:
                # the workspace holds the gradient w.r.t. the conv's pre-activation:
                # reference gradient of the post-ReLU tensor times the ReLU mask
                ref = ref * (torch.from_numpy(g["act." + lname]) > 0)
            got = model.workspace_tensor("grad." + lname).cpu()
            if dtype == "bf16" and post_relu:
                # an activation that rounds to zero in bf16 flips its mask; compare where both agree it is on
                on = (model.workspace_tensor("act." + lname).cpu() > 0) == (torch.from_numpy(g["act." + lname]) > 0)
                got, ref = got * on, ref * on
            e = rel(got, ref)
            worst[lname] = e
            n_grad += 1
    assert n_act >= 26 and n_grad >= 26
    assert max(worst.values()) < bar["grad"], sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    if dtype == "fp32":
        assert float(np.median(list(worst.values()))) < bar["grad_med"]
    # (bf16: the median bar is stated for PARAMETER gradients.  Activation gradients of these
    # reduced-width nets sit at 4-7 % for this path and at 5-14 % for torch's own CPU
    # autocast(bf16) of the oracle on the same fixture -- scripts/diag_bf16.py prints both --
    # so only the worst-tensor bar applies to them.)
    errs = {k: rel(p.grad.cpu(), torch.from_numpy(g["grad." + k])) for k, p in model.named_parameters()}
    assert max(errs.values()) < bar["grad"], sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    assert float(np.median(list(errs.values()))) < bar["grad_med"]
    assert rel(X.grad.cpu(), torch.from_numpy(g["dX"])) < bar["grad"]
    if name == "odd":
        assert "act.align.0" in names and tuple(pred.shape[2:]) == (15, 24)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(8, 64, 32), (16, 16, 64), (4, 128, 16), (32, 8, 16), (32, 16, 8), (64, 8, 32), (16, 16, 128)])
def test_rectangular_images_against_oracle(ua, shape, dtype):
    """Non-square power-of-two images: tiles that are bands of an image, whole images, or several
    images (the geometry the fused max-pool indexes by), checked layer by layer against the CPU oracle."""
    B, H, W = shape
    bar = BAR[dtype]
    feats = (8, 16, 32)
    state = O.init_state_dict(12, 1, feats, seed=3)
    model = ua.UNet(12, 1, feats, compute_dtype=dtype)
    model.load_state_dict(state)
    model = model.to("cuda")
    X, Y = O.synthetic_batch(B, 12, 1, H, W, seed=4)
    loss_ref, taps, _, pgrads, dX = O.forward_backward(state, X, Y, None, want_input_grad=True)
    Xg = X.cuda().requires_grad_(True)
    pred = model(Xg)
    loss = ua.MSELoss()(pred, Y.cuda())
    loss.backward()
    assert rel(pred.detach().cpu(), taps["out"]) < bar["out"]
    names = model.last_plan.tensors()
    checked = 0
    for k, ref in taps.items():
        if "act." + k in names:
            e = rel(model.workspace_tensor("act." + k).cpu(), ref)
            assert e < bar["out"], (k, e)
            checked += 1
            if k.startswith("pool."):       # max-pool is exact given its input (rounding is monotonic: also where it is fused)
                got_in = model.workspace_tensor("act.downs." + k[5:] + ".3")
                assert torch.equal(model.workspace_tensor("act." + k), torch.nn.functional.max_pool2d(got_in, 2)), k
    assert checked >= 3 * len(feats) + 2
    errs = {k: rel(p.grad.cpu(), pgrads[k]) for k, p in model.named_parameters()}
    assert max(errs.values()) < bar["grad"], sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    assert rel(Xg.grad.cpu(), dX) < bar["grad"]
    # the level-0 concat: two tensors where the bf16 kernels take two-tensor operands (the upsampled half is then a
    # tensor of its own: channel offset 0, stride 32), one shared buffer (offset 32 of 64) otherwise
    up = names["act.ups.%d" % (2 * (len(feats) - 1))]
    if dtype == "bf16" and shape == (8, 64, 32):
        assert (up["coff"], up["cstride"]) == (0, 32)
    if dtype == "fp32":
        assert (up["coff"], up["cstride"]) == (32, 64)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("cin,feats,B,H,W", [(12, (32, 64), 4, 32, 32), (16, (8, 16), 3, 16, 16), (3, (16,), 2, 8, 8),
                                             (17, (32, 64), 2, 16, 16), (12, (32, 64, 128, 256), 1, 32, 32)])
def test_narrow_input_pixels_and_a_poisoned_workspace(ua, cin, feats, B, H, W, dtype):
    """Inputs of up to 16 channels are stored as 16-channel pixels although the first layer reads 32 per pixel: the
    rest of its chunk is the NEXT pixel (and, behind the last one, a zeroed tail), which meets zero weights.  Nothing
    a pass reads may be memory it did not write: a workspace filled with NaN bit patterns before the first pass
    gives bit-identical predictions and gradients (and finite ones) to a zero-filled one."""
    state = O.init_state_dict(cin, 1, feats, seed=11)
    X, Y = O.synthetic_batch(B, cin, 1, H, W, seed=12)
    results = []
    for fill in (0x00, 0xFF):
        model = ua.UNet(cin, 1, feats, compute_dtype=dtype)
        model.load_state_dict(state)
        model = model.to("cuda")
        Xg = X.cuda().requires_grad_(True)
        model._ensure_flat(Xg.device)
        plan = model._get_plan(Xg.detach(), True)
        plan.ws.fill_(fill)
        pred = model(Xg)
        ua.MSELoss()(pred, Y.cuda()).backward()
        info = plan.tensors()["act.x"]
        assert info["cstride"] == (16 if cin <= 16 else 32), info
        assert torch.equal(model.workspace_tensor("act.x"), X.cuda().to(model.compute_dtype).float())
        results.append([pred.detach().clone(), Xg.grad.clone()] + [p.grad.clone() for p in model.parameters()])
    for a, b in zip(*results):
        assert torch.isfinite(b).all()
        assert torch.equal(a, b)
    taps = O.forward_backward(state, X, Y, None)[1]
    assert rel(results[1][0].cpu(), taps["out"]) < BAR[dtype]["out"]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("arch", [
    # (in, out, features, B, H, W): depths 1..6, widths that are not multiples of 16, one-image batches,
    # sizes that are odd at some level (both bilinear resizes), many output channels.  (The one-level
    # net gets 48 images: with 3 its 4x4 bottleneck has 48 pixels and a single bf16-flipped ReLU moves
    # a gradient tensor by a quarter of its norm.)
    (1, 1, (8,), 48, 8, 8), (3, 2, (6, 12), 1, 12, 20), (12, 1, (16, 32, 64, 128, 256), 2, 32, 32),
    (5, 7, (4, 8, 16, 32, 64, 128), 2, 64, 64), (2, 16, (24, 48), 5, 10, 6), (12, 12, (32, 64), 1, 17, 23),
    (7, 3, (20, 40, 80), 2, 36, 28),
    # 64 channels on 8x4 images, 1280 pixels: the few-pixel conv kernel pools 32-pixel tiles of 4-wide images
    (12, 1, (32, 64), 40, 16, 8)])
def test_architecture_variants_against_oracle(ua, arch, dtype):
    """The constructor's whole argument space (models/unet2d.py:9-15), not only the benched widths:
    prediction, every parameter gradient and the input gradient against the CPU oracle."""
    cin, cout, feats, B, H, W = arch
    bar = BAR[dtype]
    state = O.init_state_dict(cin, cout, feats, seed=5)
    model = ua.UNet(cin, cout, feats, compute_dtype=dtype)
    model.load_state_dict(state)
    model = model.to("cuda")
    X, Y = O.synthetic_batch(B, cin, cout, H, W, seed=6)
    loss_ref, taps, _, pgrads, dX = O.forward_backward(state, X, Y, None, want_input_grad=True)
    Xg = X.cuda().requires_grad_(True)
    pred = model(Xg)
    assert tuple(pred.shape) == tuple(taps["out"].shape)
    loss = ua.MSELoss()(pred, Y.cuda())
    loss.backward()
    assert rel(pred.detach().cpu(), taps["out"]) < bar["out"]
    assert abs(loss.item() - loss_ref.item()) <= bar["out"] * abs(loss_ref.item())
    errs = {k: rel(p.grad.cpu(), pgrads[k]) for k, p in model.named_parameters()}
    assert max(errs.values()) < bar["grad"], sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    assert rel(Xg.grad.cpu(), dX) < bar["grad"]
    # the max-pool, wherever it ran (own launch or a conv's epilogue), is exact given the tensor it pools
    names = model.last_plan.tensors()
    for l in range(len(feats)):
        if "act.pool.%d" % l in names and "act.downs.%d.3" % l in names:
            src = model.workspace_tensor("act.downs.%d.3" % l)
            assert torch.equal(model.workspace_tensor("act.pool.%d" % l), torch.nn.functional.max_pool2d(src, 2)), l


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["small4", "odd", "cfg1_full", "cfg2_b32", "cfg4_b2"])
def test_loss_after_n_adam_steps(ua, golden, name, dtype):
    """The loop body of train.py:382-393 (forward, MSE, zero_grad, backward, Adam step) for
    20 steps on one fixed batch; the loss curve must track the reference's."""
    g = golden(name)
    cin, cout, feats, osz = meta(g)
    b = int(g["meta.batch"]); h, w = (int(v) for v in g["meta.hw"])
    torch.manual_seed(0)
    model = ua.UNet(cin, cout, feats, out_size=osz, compute_dtype=dtype).to("cuda")
    oh, ow = osz if osz else (h, w)
    gen = torch.Generator().manual_seed(1)
    X = torch.rand(b, cin, h, w, generator=gen).cuda()
    Y = torch.rand(b, cout, oh, ow, generator=gen).cuda()
    opt = ua.FusedAdam(model.parameters(), lr=1e-4, model=model)
    crit = ua.MSELoss()
    losses = []
    model.train()
    for _ in range(len(g["losses"])):
        pred = model(X)
        loss = crit(pred, Y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    ref = np.asarray(g["losses"])
    dev = np.max(np.abs(np.asarray(losses) - ref) / np.abs(ref))
    assert dev < BAR[dtype]["loss"], (dev, losses[:3], ref[:3])
    model.eval()
    with torch.no_grad():
        sums, n = ua.eval_sums(model(X), Y)          # test.py:39-43
    sums = sums.cpu()
    # observed on MI355X over the five fixtures: fp32 <= 8e-8, bf16 <= 6.1e-5 (|d|) / 1.0e-4 (d^2); bars = 3x that
    tol = 3e-7 if dtype == "fp32" else 3e-4
    print(f"eval sums[{name},{dtype}]: rel dev |d| {abs(sums[0].item() - float(g['eval.abs_sum'])) / float(g['eval.abs_sum']):.3e} "
          f"d^2 {abs(sums[1].item() - float(g['eval.sq_sum'])) / float(g['eval.sq_sum']):.3e} loss curve {dev:.3e}")
    assert n == int(g["eval.numel"])
    assert abs(sums[0].item() - float(g["eval.abs_sum"])) <= tol * float(g["eval.abs_sum"])
    assert abs(sums[1].item() - float(g["eval.sq_sum"])) <= tol * float(g["eval.sq_sum"])


def test_unchanged_reference_loop_with_stock_optimizer_and_loss(ua, golden):
    """torch.optim.Adam + nn.MSELoss exactly as train.py:497-498 builds them, driving the
    HIP module through autograd."""
    g = golden("small4")
    cin, cout, feats, osz = meta(g)
    cfg = types.SimpleNamespace(model_type="UNet", in_channels=cin, out_channels=cout, unet_features=feats,
                                compute_dtype="fp32")
    torch.manual_seed(0)
    model = ua.build_model_from_cfg(cfg).to("cuda:0")        # train.py:495, config.py:64
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-4)
    criterion = torch.nn.MSELoss()
    X, Y = torch.from_numpy(g["X"]).to("cuda:0"), torch.from_numpy(g["Y"]).to("cuda:0")
    losses = []
    for _ in range(len(g["losses"])):
        pred = model(X)
        loss = criterion(pred, Y)
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        losses.append(loss.item())
    ref = np.asarray(g["losses"])
    assert np.max(np.abs(np.asarray(losses) - ref) / np.abs(ref)) < 1e-5


@pytest.mark.parametrize("name", ["cfg1_full", "cfg2_b32", "cfg4_b2"])
def test_full_width_samples_and_norms(ua, golden, name):
    g = golden(name)
    cin, cout, feats, osz = meta(g)
    b = int(g["meta.batch"]); h, w = (int(v) for v in g["meta.hw"])
    torch.manual_seed(0)
    model = ua.UNet(cin, cout, feats, compute_dtype="fp32")
    for k, v in model.state_dict().items():                 # same-seed init == reference init
        assert v.double().norm().item() == float(g["paramnorm." + k])
    model = model.to("cuda")
    X, Y = O.synthetic_batch(b, cin, cout, h, w)
    pred = model(X.cuda())
    loss = ua.MSELoss()(pred, Y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss0"])) <= 1e-5 * float(g["loss0"])
    pv = pred.detach().cpu().flatten()[torch.from_numpy(g["predidx"])]
    assert rel(pv, torch.from_numpy(g["predval"])) < 1e-5
    assert abs(pred.detach().double().norm().item() - float(g["prednorm"])) <= 1e-5 * float(g["prednorm"])
    for k, p in model.named_parameters():
        gn = float(g["gradnorm." + k])
        assert abs(p.grad.double().norm().item() - gn) <= 5e-5 * gn, k
        got = p.grad.cpu().flatten()[torch.from_numpy(g["gradidx." + k])]
        ref = torch.from_numpy(g["gradval." + k])
        assert (got - ref).abs().max().item() <= 5e-5 * gn / np.sqrt(p.numel()) * 30 + 1e-12, k
    for k in g.files:
        if k.startswith("actnorm.") and ("act." + k[8:]) in model.last_plan.tensors():
            t = model.workspace_tensor("act." + k[8:]).cpu()
            assert abs(t.double().norm().item() - float(g[k])) <= 1e-5 * float(g[k]), k
            idx = torch.from_numpy(g["actidx." + k[8:]])
            assert rel(t.flatten()[idx], torch.from_numpy(g["actval." + k[8:]])) < 1e-4, k


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_inference_sweep_64x64_against_oracle(ua, dtype):
    """The inference configuration (test.py:24-47 / :33-38: model.eval(), torch.no_grad(), pred = model(X)):
    default widths, 12 channels, 64x64 matrices.  Batches 1, 8 and 256 against the CPU oracle's forward
    pass; batch 8192 (activation tensors beyond 2^31 bytes, so the 64-bit index paths run) through a
    size-independent property: a batch made of 32 copies of the 256 matrices predicts, row for row,
    what the 256 alone predict (to rounding only -- the library may pick other kernels at that size)."""
    bar = BAR[dtype]["out"]
    state = O.init_state_dict(12, 1, (32, 64, 128, 256), seed=0)
    model = ua.UNet(12, 1, (32, 64, 128, 256), compute_dtype=dtype)
    model.load_state_dict(state)
    model = model.to("cuda").eval()
    X, _ = O.synthetic_batch(256, 12, 1, 64, 64, seed=7)
    ref = O.unet_forward(state, X)
    Xd = X.cuda()
    with torch.no_grad():
        p256 = model(Xd).clone()
        for b in (1, 8):
            pb = model(Xd[:b])
            assert tuple(pb.shape) == (b, 1, 64, 64)
            assert rel(pb.cpu(), ref[:b]) < bar, b
            for i in range(b):                      # every row on its own, not only the batch's norm
                assert rel(pb[i].cpu(), ref[i]) < 2 * bar, (b, i)
        assert rel(p256.cpu(), ref) < bar
        worst = max(rel(p256[i].cpu(), ref[i]) for i in range(256))
        assert worst < 2 * bar, worst
        big = model(Xd.repeat(32, 1, 1, 1))
        assert tuple(big.shape) == (8192, 1, 64, 64)
        tol = 1e-6 if dtype == "fp32" else bar
        for r in (0, 1, 15, 16, 31):
            assert rel(big[r * 256:(r + 1) * 256], p256) < tol, r
        assert torch.isfinite(big).all()
        sums, n = ua.eval_sums(big, torch.zeros_like(big))      # test.py:39-43 at that size
        assert n == 8192 * 64 * 64
        assert abs(sums[1].item() - 32 * p256.double().pow(2).sum().item()) <= 1e-4 * sums[1].item()
    del big, model
    torch.cuda.empty_cache()


def test_state_dict_interchange_and_eval(ua):
    torch.manual_seed(3)
    model = ua.UNet(12, 1, (8, 16, 32, 64), compute_dtype="fp32")
    ref_state = O.init_state_dict(12, 1, (8, 16, 32, 64), seed=3)
    sd = model.state_dict()
    assert list(sd.keys()) == list(ref_state.keys())        # downs, ups, bottleneck, final_conv
    for k in sd:
        assert sd[k].shape == ref_state[k].shape and torch.equal(sd[k], ref_state[k])
    model = model.to("cuda")
    X, _ = O.synthetic_batch(3, 12, 1, 16, 16)
    model.eval()
    with torch.no_grad():
        p_eval = model(X.cuda())
    model.train()
    p_train = model(X.cuda())
    assert torch.equal(p_eval, p_train.detach())            # no BatchNorm/Dropout: modes agree bit for bit
    assert rel(p_eval.cpu(), O.unet_forward(ref_state, X)) < 1e-5
    # round trip through a checkpoint dict as train.py:522-529 / test.py:134-135 do
    ckpt = {"model_state_dict": {k: v.cpu() for k, v in model.state_dict().items()}, "best_val_loss": 0.1, "epoch": 1}
    other = ua.UNet(12, 1, (8, 16, 32, 64), compute_dtype="fp32").to("cuda")
    other.load_state_dict(ckpt["model_state_dict"])
    with torch.no_grad():
        assert torch.equal(other(X.cuda()), p_eval)
    # parameters updated in place by a stock optimiser are picked up (weights are re-packed)
    with torch.no_grad():
        for p in other.parameters():
            p.mul_(0.5)
        assert not torch.equal(other(X.cuda()), p_eval)


def test_errors(ua):
    cfg = types.SimpleNamespace(model_type="transformer", in_channels=1)
    with pytest.raises(ValueError, match="Unknown cfg.model_type"):   # models/builders.py:132
        ua.build_model_from_cfg(cfg)
    model = ua.UNet(12, 1, (8, 16, 32, 64)).to("cuda")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.zeros(1, 12, 16, 16))
    with pytest.raises(ValueError):
        model(torch.zeros(1, 3, 16, 16, device="cuda"))
    with pytest.raises(RuntimeError, match="too small"):             # the reference fails the same way (12x12 input)
        model(torch.zeros(1, 12, 12, 12, device="cuda"))
    with pytest.raises(RuntimeError, match="double"):
        ua.UNet(1, 1, (8, 16, 24)).to("cuda")(torch.zeros(1, 1, 16, 16, device="cuda"))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("stock_optim", [False, True])
def test_train_step_equals_the_loop_body_bit_for_bit(ua, dtype, stock_optim):
    """UNet.train_step (direct library calls) against forward / MSELoss / zero_grad / backward / step
    through autograd: same losses and same parameters after six steps, with either optimizer; mixing
    the two forms in one run works too."""
    feats = (8, 16, 32)
    batches = [tuple(t.cuda() for t in O.synthetic_batch(6, 12, 1, 16, 16, seed=s)) for s in range(6)]

    def run(mode):
        torch.manual_seed(0)
        m = ua.UNet(12, 1, feats, compute_dtype=dtype).cuda()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3) if stock_optim else ua.FusedAdam(m.parameters(), lr=1e-3, model=m)
        crit = ua.MSELoss()
        losses = []
        for i, (X, Y) in enumerate(batches):
            if mode == "fused" or (mode == "mixed" and i % 2 == 0):
                losses.append(m.train_step(X, Y, opt))
            else:
                loss = crit(m(X), Y)
                opt.zero_grad()
                loss.backward()
                opt.step()
                losses.append(loss.detach())
        return torch.stack(losses).cpu(), [p.detach().cpu().clone() for p in m.parameters()]

    l_ref, p_ref = run("loop")
    for mode in ("fused", "mixed"):
        l, p = run(mode)
        assert torch.equal(l, l_ref), mode
        for a, b in zip(p, p_ref):
            assert torch.equal(a, b), mode


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_fused_adam_alone_refreshes_packed_weights_and_checkpoints(ua, dtype):
    """``FusedAdam(model.parameters(), lr=...)`` -- the drop-in for train.py:497, with no reference to the
    module -- writes the parameters through raw pointers; the module must still notice (packed MFMA
    weight images refreshed every step), so the run equals, bit for bit, one whose optimizer was also
    handed the module, and tracks stock torch.optim.Adam.  The optimizer's state_dict has torch.optim.Adam's
    format and a run resumed from it continues bit for bit."""
    feats = (8, 16, 32)
    batches = [tuple(t.cuda() for t in O.synthetic_batch(6, 12, 1, 16, 16, seed=s)) for s in range(8)]
    crit = ua.MSELoss()

    def run(make_opt, resume_at=None):
        torch.manual_seed(0)
        m = ua.UNet(12, 1, feats, compute_dtype=dtype).cuda()
        opt = make_opt(m)
        losses = []
        for i, (X, Y) in enumerate(batches):
            if resume_at is not None and i == resume_at:
                sd = opt.state_dict()
                p0 = next(iter(sd["state"].values()))
                assert set(p0) >= {"step", "exp_avg", "exp_avg_sq"} and float(p0["step"]) == resume_at
                stock = torch.optim.Adam(m.parameters(), lr=1e-3)
                stock.load_state_dict(sd)                    # interchangeable with the stock optimizer's format
                opt = make_opt(m)
                opt.load_state_dict(stock.state_dict())
            loss = crit(m(X), Y)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        with torch.no_grad():
            final = m(batches[0][0]).clone()
        return torch.stack(losses).cpu(), [p.detach().cpu().clone() for p in m.parameters()], final.cpu()

    with_model = run(lambda m: ua.FusedAdam(m.parameters(), lr=1e-3, model=m))
    alone = run(lambda m: ua.FusedAdam(m.parameters(), lr=1e-3))
    resumed = run(lambda m: ua.FusedAdam(m.parameters(), lr=1e-3), resume_at=4)
    stock = run(lambda m: torch.optim.Adam(m.parameters(), lr=1e-3))
    for other in (alone, resumed):
        assert torch.equal(other[0], with_model[0])
        assert torch.equal(other[2], with_model[2])
        for a, b in zip(other[1], with_model[1]):
            assert torch.equal(a, b)
    assert with_model[0][0] != with_model[0][-1]
    tol = 1e-5 if dtype == "fp32" else 2e-3
    assert ((alone[0] - stock[0]).abs() / stock[0].abs()).max().item() < tol


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("arch", [(12, 1, (32, 64, 128, 256), 4, 16, 16), (24, 12, (32, 64, 128, 256, 512), 2, 32, 32),
                                  (12, 1, (32, 64), 3, 8, 24), (4, 2, (48, 96), 2, 16, 16)])
def test_adam_step_fused_with_weight_packing_is_bit_identical(ua, dtype, arch):
    """``FusedAdam(..., model=m)`` steps and writes the packed MFMA weight images in ONE launch
    (mi_unet_adam_pack) where the plan allows it; the run must equal, bit for bit, the one that steps with
    the plain Adam kernel and lets the next forward repack (``fuse_pack = False``): losses (forward images),
    gradients of every later step (backward images), parameters, moments, and an evaluation pass at
    another batch size (a second plan, which must notice that its own images are stale)."""
    cin, cout, feats, B, H, W = arch
    batches = [tuple(t.cuda() for t in O.synthetic_batch(B, cin, cout, H, W, seed=s)) for s in range(5)]
    crit = ua.MSELoss()

    def run(fuse):
        torch.manual_seed(0)
        m = ua.UNet(cin, cout, feats, compute_dtype=dtype).cuda()
        opt = ua.FusedAdam(m.parameters(), lr=1e-3, model=m)
        opt.fuse_pack = fuse
        losses, gnorms = [], []
        for i, (X, Y) in enumerate(batches):
            if i % 2 == 0:
                loss = crit(m(X), Y)
                opt.zero_grad()
                loss.backward()
                gnorms.append(torch.stack([p.grad.norm() for p in m.parameters()]))
                opt.step()
                losses.append(loss.detach())
            else:
                losses.append(m.train_step(X, Y, opt).clone())
            if i == 2:
                with torch.no_grad():
                    mid = m(batches[0][0][:1]).clone()
        assert m.last_train_plan.adam_pack_ok
        with torch.no_grad():
            final = m(batches[0][0]).clone()
        named = dict(m.named_parameters())
        mom = [opt.state[named[k]][key].detach().cpu().clone() for k in sorted(named) for key in ("exp_avg", "exp_avg_sq")]
        return (torch.stack(losses).cpu(), torch.stack(gnorms).cpu(), [p.detach().cpu().clone() for p in m.parameters()],
                mom, mid.cpu(), final.cpu())

    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(a[4], b[4]) and torch.equal(a[5], b[5])
    for x, y in zip(a[2] + a[3], b[2] + b[3]):
        assert torch.equal(x, y)
    assert a[0][0] != a[0][-1]


def test_adam_pack_entry_point_states_and_limits(ua):
    """mi_unet_adam_pack through the C ABI: refused on a workspace no repacking forward has initialised
    (MI_ESTATE), refused for a plan whose layers cannot go through the fused kernel (a concatenated input whose
    halves are padded: mi_unet_plan_adam_pack_supported says so first), and FusedAdam falls back to the plain
    step there with the same results as without the model."""
    import ctypes as C
    L = ua.lib()
    torch.manual_seed(0)
    m = ua.UNet(12, 1, (32, 64), compute_dtype="bf16").cuda()
    X, Y = (t.cuda() for t in O.synthetic_batch(4, 12, 1, 16, 16, seed=1))
    loss = ua.MSELoss()(m(X), Y)
    loss.backward()
    plan = m.last_train_plan
    assert L.mi_unet_plan_adam_pack_supported(plan.handle) == 1
    flat = m.flat_parameters()
    g = torch.zeros_like(flat); mm = torch.zeros_like(flat); vv = torch.zeros_like(flat)
    other = torch.empty(plan.ws_bytes, device="cuda", dtype=torch.uint8)
    stream = torch.cuda.current_stream().cuda_stream
    rc = L.mi_unet_adam_pack(plan.handle, flat.data_ptr(), g.data_ptr(), mm.data_ptr(), vv.data_ptr(), other.data_ptr(),
                             plan.ws_bytes, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, stream)
    assert rc != 0 and b"MI_FWD_REPACK" in L.mi_last_error()
    rc = L.mi_unet_adam_pack(plan.handle, flat.data_ptr(), g.data_ptr(), mm.data_ptr(), vv.data_ptr(), plan.ws.data_ptr(),
                             plan.ws_bytes - 256, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, stream)
    assert rc != 0 and b"too small" in L.mi_last_error()
    torch.cuda.synchronize()

    torch.manual_seed(0)
    small = ua.UNet(12, 1, (8, 16), compute_dtype="bf16").cuda()          # 8 + 8 channels concatenated: padded halves
    opt = ua.FusedAdam(small.parameters(), lr=1e-3, model=small)
    l0 = small.train_step(X, Y, opt)
    assert L.mi_unet_plan_adam_pack_supported(small.last_train_plan.handle) == 0 and not small.last_train_plan.adam_pack_ok
    l1 = small.train_step(X, Y, opt)
    torch.manual_seed(0)
    ref = ua.UNet(12, 1, (8, 16), compute_dtype="bf16").cuda()
    ropt = ua.FusedAdam(ref.parameters(), lr=1e-3)
    r0 = ref.train_step(X, Y, ropt); r1 = ref.train_step(X, Y, ropt)
    assert torch.equal(l0, r0) and torch.equal(l1, r1) and l0 != l1


def test_overlapping_forward_passes_keep_their_activations(ua):
    """What stock autograd gives the reference's module: several forward passes may precede one
    backward, an evaluation may sit between a forward and its backward, gradients accumulate
    across backward calls, a retained graph can be walked twice -- and a graph whose activations
    have been overwritten fails loudly instead of returning wrong gradients."""
    torch.manual_seed(0)
    m = ua.UNet(3, 2, features=(8, 16), compute_dtype="fp32").cuda()
    x1 = torch.randn(4, 3, 16, 16, device="cuda")
    x2 = torch.randn(4, 3, 16, 16, device="cuda")
    state = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    ((O.unet_forward(state, x1.cpu()) ** 2).mean() + (O.unet_forward(state, x2.cpu()) ** 2).mean()).backward()
    ref = [state[k].grad for k, _ in m.named_parameters()]

    def run(mode):
        for p in m.parameters():
            p.grad = None
        if mode == "one_graph":
            ((m(x1) ** 2).mean() + (m(x2) ** 2).mean()).backward()
        elif mode == "accumulate":
            (m(x1) ** 2).mean().backward()
            (m(x2) ** 2).mean().backward()
        elif mode == "zero_in_place":
            (m(x1) ** 2).mean().backward()
            for p in m.parameters():
                p.grad.zero_()
            (m(x1) ** 2).mean().backward()
            (m(x2) ** 2).mean().backward()
        elif mode == "eval_between":
            loss = (m(x1) ** 2).mean()
            with torch.no_grad():
                m(x2)
            loss.backward()
            (m(x2) ** 2).mean().backward()
        elif mode == "retained":
            loss = (m(x1) ** 2).mean() + (m(x2) ** 2).mean()
            loss.backward(retain_graph=True)
            for p in m.parameters():
                p.grad = None
            loss.backward()
        return [p.grad.cpu() for p in m.parameters()]

    for mode in ("one_graph", "accumulate", "zero_in_place", "eval_between", "retained"):
        for g, r in zip(run(mode), ref):
            assert rel(g, r) < 5e-5, mode
    # the standard loop keeps to one workspace: the plan is free again once backward has run
    n_plans = sum(len(v) for v in m._plans.values())
    for _ in range(3):
        (m(x1) ** 2).mean().backward()
    assert sum(len(v) for v in m._plans.values()) == n_plans
    loss = (m(x1) ** 2).mean()
    loss.backward(retain_graph=True)
    (m(x2) ** 2).mean().backward()            # reuses the released workspace
    with pytest.raises(RuntimeError, match="activations of this forward pass are gone"):
        loss.backward()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_full_size_properties(ua, dtype):
    """Size-independent properties at the benchmark's own size (12ch 32x32, batch 512):
    bitwise run-to-run determinism, linearity of backward in the upstream gradient, and
    batch additivity (the identity data-parallel training relies on)."""
    torch.manual_seed(0)
    model = ua.UNet(12, 1, compute_dtype=dtype).to("cuda")
    X, Y = O.synthetic_batch(512, 12, 1, 32, 32)
    X, Y = X.cuda(), Y.cuda()
    crit = ua.MSELoss()

    def grads(xb, yb, scale=1.0):
        model.zero_grad(set_to_none=True)
        loss = crit(model(xb), yb) * scale
        loss.backward()
        return loss.detach(), torch.cat([p.grad.flatten() for p in model.parameters()]).clone()

    l1, g1 = grads(X, Y)
    l2, g2 = grads(X, Y)
    assert torch.equal(l1, l2) and torch.equal(g1, g2)
    _, g3 = grads(X, Y, scale=4.0)
    assert rel(g3, 4.0 * g1) < (1e-6 if dtype == "fp32" else 1e-2)
    _, ga = grads(X[:256], Y[:256])
    _, gb = grads(X[256:], Y[256:])
    assert rel(0.5 * (ga + gb), g1) < (2e-5 if dtype == "fp32" else 2e-2)
    assert torch.isfinite(g1).all()


def test_bucketed_backward_matches_single_call():
    """The data-parallel route runs backward bucket by bucket (several mi_unet_backward calls over
    segment ranges, a collective started after each): same gradients, bit for bit, as one call."""
    import unet_amd

    class Recorder(unet_amd.GradientReducer):
        def reduce_async(self, t):
            self.calls.append(t.numel())

        def wait_all(self):
            pass

    torch.manual_seed(0)
    m = unet_amd.UNet(12, 1, (8, 16, 32, 64), compute_dtype="bf16").cuda()
    X, Y = O.synthetic_batch(8, 12, 1, 32, 32)
    X, Y = X.cuda(), Y.cuda()
    crit = unet_amd.MSELoss()
    crit(m(X), Y).backward()
    g0 = [p.grad.clone() for p in m.parameters()]
    red = Recorder(bucket_bytes=64 << 10)
    red.calls, red.force = [], True
    red.attach(m)
    assert m._dp is red
    for p in m.parameters():
        p.grad = None
    crit(m(X), Y).backward()
    torch.cuda.synchronize()
    assert len(red.calls) > 1 and sum(red.calls) >= sum(p.numel() for p in m.parameters())
    for a, p in zip(g0, m.parameters()):
        assert torch.equal(a, p.grad)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("feats,cout,hw", [((8, 16, 32, 64), 1, (32, 32)), ((8, 16), 3, (17, 23))])
def test_data_parallel_prescale_reaches_every_gradient(dtype, feats, cout, hw):
    """Data-parallel runs fold 1/world into the backward pass (alpha of every weight-gradient combine)
    instead of scaling the flat buffer afterwards: with a prescale of 1/8 EVERY parameter gradient --
    3x3 convs, transposed convs, the 1x1 head, all biases -- must be one eighth of the unscaled one,
    while the input gradient (not a parameter gradient, never reduced) stays unscaled."""
    import unet_amd

    class Recorder(unet_amd.GradientReducer):
        def reduce_async(self, t):
            pass

        def wait_all(self):
            pass

    torch.manual_seed(0)
    m = unet_amd.UNet(12, cout, feats, compute_dtype=dtype).cuda()
    X, Y = O.synthetic_batch(8, 12, cout, *hw)
    X, Y = X.cuda().requires_grad_(True), Y.cuda()
    crit = unet_amd.MSELoss()
    crit(m(X), Y).backward()
    g0 = {k: p.grad.clone() for k, p in m.named_parameters()}
    dx0 = X.grad.clone()
    for bucket_bytes in (1, 64 << 10, 1 << 30):          # every segment its own bucket / a few / one
        red = Recorder(bucket_bytes=bucket_bytes)
        red.force = True
        red.grad_prescale = 0.125
        red.attach(m)
        for p in m.parameters():
            p.grad = None
        X.grad = None
        crit(m(X), Y).backward()
        torch.cuda.synchronize()
        for k, p in m.named_parameters():
            assert g0[k].abs().max().item() > 0, k
            assert rel(p.grad, 0.125 * g0[k]) < 1e-6, (k, bucket_bytes)    # a power of two: exact up to rounding
        assert torch.equal(X.grad, dx0)
    m._dp = None


def test_one_rank_rccl_group_runs_the_collective_path(monkeypatch):
    """A one-rank RCCL group: the all-reduces of the gradient buckets and the stream hand-over to the
    optimizer run for real; the averaged gradient of one rank is its own.  Also with captured launch
    graphs, where a bucket's call must join its streams (a capture cannot end with a deferred join)."""
    import torch.distributed as dist
    import unet_amd
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        torch.manual_seed(0)
        m = unet_amd.UNet(12, 1, (8, 16, 32, 64), compute_dtype="fp32").cuda()
        X, Y = O.synthetic_batch(4, 12, 1, 16, 16)
        X, Y = X.cuda(), Y.cuda()
        crit = unet_amd.MSELoss()
        crit(m(X), Y).backward()
        g0 = [p.grad.clone() for p in m.parameters()]
        finals = {}
        for mode in unet_amd.GradientReducer.MODES:      # one all-reduce per bucket / reduce-scatter + all-gather
            torch.manual_seed(0)
            m = unet_amd.UNet(12, 1, (8, 16, 32, 64), compute_dtype="fp32").cuda()
            red = unet_amd.GradientReducer(bucket_bytes=64 << 10, force=True, mode=mode).attach(m)
            assert m._dp is red and red.mode == mode
            unet_amd.broadcast_parameters(m)
            opt = unet_amd.FusedAdam(m.parameters(), lr=1e-4, model=m)
            for _ in range(3):
                loss = crit(m(X), Y)
                opt.zero_grad()
                loss.backward()
                if _ == 0:
                    for a, p in zip(g0, m.parameters()):
                        assert torch.equal(a, p.grad), mode
                opt.step()
            torch.cuda.synchronize()
            assert torch.isfinite(loss).item()
            # several buckets, reduced behind the plan's second stream without joining the streams in between
            assert len(m._dp.buckets_for(m.last_plan)) >= 2
            assert getattr(m.last_plan, "side_stream", None) is not None
            finals[mode] = [p.detach().clone() for p in m.parameters()]
        for a, b in zip(finals["allreduce"], finals["rs_ag"]):       # one rank: both leave the gradients as they were
            assert torch.equal(a, b)
        monkeypatch.setenv("UNET_TM_GRAPH", "1")
        torch.manual_seed(0)
        mg = unet_amd.UNet(12, 1, (8, 16, 32, 64), compute_dtype="fp32").cuda()
        unet_amd.GradientReducer(bucket_bytes=64 << 10, force=True).attach(mg)
        for _ in range(4):                       # replayed from the captured graphs from the third pass on
            for p in mg.parameters():
                p.grad = None
            crit(mg(X), Y).backward()
        torch.cuda.synchronize()
        for a, p in zip(g0, mg.parameters()):
            assert torch.equal(a, p.grad)
    finally:
        dist.destroy_process_group()


def test_captured_launch_graphs_give_identical_results(monkeypatch):
    """UNET_TM_GRAPH=1: the plan-owned part of forward/backward is replayed from a HIP graph from
    the third call on; predictions and gradients must equal the eager launches bit for bit, also
    when the input batch changes (the graph must not have baked the first batch in)."""
    import unet_amd
    torch.manual_seed(0)
    feats = (8, 16, 32, 64)
    crit = unet_amd.MSELoss()
    batches = [tuple(t.cuda() for t in O.synthetic_batch(8, 12, 1, 32, 32, seed=s)) for s in (1, 2, 3, 4)]

    def run(env):
        monkeypatch.setenv("UNET_TM_GRAPH", env)
        torch.manual_seed(0)
        m = unet_amd.UNet(12, 1, feats, compute_dtype="bf16").cuda()
        outs = []
        for X, Y in batches:
            for p in m.parameters():
                p.grad = None
            pred = m(X)
            crit(pred, Y).backward()
            outs.append((pred.detach().clone(), [p.grad.clone() for p in m.parameters()]))
        torch.cuda.synchronize()
        return outs

    eager, graphed = run("0"), run("1")
    for (p0, g0), (p1, g1) in zip(eager, graphed):
        assert torch.equal(p0, p1)
        for a, b in zip(g0, g1):
            assert torch.equal(a, b)


@pytest.mark.parametrize("graph", ["0", "1"])
@pytest.mark.parametrize("bucket_bytes", [1, 64 << 10, 1 << 30])
def test_dp_ordering_with_a_late_collective_that_changes_the_data(monkeypatch, graph, bucket_bytes):
    """An all-reduce in a one-rank group is the identity, so it cannot show a collective that runs before
    its bucket's weight gradients are final, or an optimizer that reads a bucket before its collective is
    done.  Here the collective's stand-in is LATE and CHANGES the data: on the stream that is current at
    the call it first runs >= 50 us of unrelated GPU work, then doubles the slice in place.
      * every p.grad the optimizer sees must be exactly twice the gradient of a run without data
        parallelism (a collective enqueued ahead of a producer would double stale data, which the
        producer then overwrites: 1x; a doubled partial result: anything else);
      * the parameters after FusedAdam.step() must equal, bit for bit, those of a run that was handed
        2*g directly (an optimizer that did not wait for the last collective would have used 1*g)."""
    import unet_amd
    monkeypatch.setenv("UNET_TM_GRAPH", graph)
    feats = (8, 16, 32, 64)
    X, Y = (t.cuda() for t in O.synthetic_batch(8, 12, 1, 32, 32))
    crit = unet_amd.MSELoss()
    busy_a = torch.randn(2048, 2048, device="cuda")
    busy_b = torch.empty_like(busy_a)

    class LateDoubler(unet_amd.GradientReducer):
        calls = 0

        def reduce_async(self, t):
            torch.mm(busy_a, busy_a, out=busy_b)      # 17 GFLOP of fp32: well over 50 us in front of the change
            t.mul_(2.0)
            LateDoubler.calls += 1

        def wait_all(self):
            pass

    def run(dp, steps=4):
        torch.manual_seed(0)
        m = unet_amd.UNet(12, 1, feats, compute_dtype="bf16").cuda()
        opt = unet_amd.FusedAdam(m.parameters(), lr=1e-3, model=m)
        if dp:
            red = LateDoubler(bucket_bytes=bucket_bytes)
            red.force = True
            red.attach(m)
            assert m._dp is red
        seen = []
        for _ in range(steps):               # with graphs on, the launch sequences replay from the third step
            loss = crit(m(X), Y)
            opt.zero_grad()
            loss.backward()
            if not dp:
                for p in m.parameters():
                    p.grad.mul_(2.0)
            seen.append([p.grad.clone() for p in m.parameters()])
            opt.step()
        torch.cuda.synchronize()
        return seen, [p.detach().clone() for p in m.parameters()], m

    # timing of the stand-in: the delay in front of the data change is what makes a missing wait visible
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.mm(busy_a, busy_a, out=busy_b)
    e0.record(); torch.mm(busy_a, busy_a, out=busy_b); e1.record()
    torch.cuda.synchronize()
    assert e0.elapsed_time(e1) >= 0.05

    ref_grads, ref_params, _ = run(dp=False)
    LateDoubler.calls = 0
    got_grads, got_params, m = run(dp=True)
    n_buckets = len(m._dp.buckets_for(m.last_plan))
    assert LateDoubler.calls == 4 * n_buckets
    assert n_buckets == (10 if bucket_bytes == 1 else 1 if bucket_bytes == 1 << 30 else n_buckets) and n_buckets >= 1
    for step, (gs, rs) in enumerate(zip(got_grads, ref_grads)):
        for i, (a, b) in enumerate(zip(gs, rs)):
            assert torch.equal(a, b), (step, i)
    for a, b in zip(got_params, ref_params):
        assert torch.equal(a, b)


def test_full_size_properties_of_the_deep_configuration(golden):
    """The 5-level 24 -> 12 channel network on 128x128 matrices at the batch the benchmark runs it with
    (64), bf16: bitwise run-to-run determinism, finite gradients, batch additivity (two halves of 32
    against the 64), and the first two rows predicted as a batch of two predicts them -- whose norm the
    reference recorded (cfg4_b2: the same two inputs)."""
    import unet_amd
    g = golden("cfg4_b2")
    cin, cout, feats, _ = meta(g)
    assert (cin, cout, len(feats)) == (24, 12, 5)
    torch.manual_seed(0)
    model = unet_amd.UNet(cin, cout, feats, compute_dtype="bf16").to("cuda")
    X, Y = O.synthetic_batch(64, cin, cout, 128, 128)
    X, Y = X.cuda(), Y.cuda()
    crit = unet_amd.MSELoss()

    def grads(xb, yb):
        model.zero_grad(set_to_none=True)
        pred = model(xb)
        loss = crit(pred, yb)
        loss.backward()
        return loss.detach(), pred.detach().clone(), torch.cat([p.grad.flatten() for p in model.parameters()]).clone()

    l1, p1, g1 = grads(X, Y)
    l2, p2, g2 = grads(X, Y)
    assert torch.equal(l1, l2) and torch.equal(p1, p2) and torch.equal(g1, g2)
    assert torch.isfinite(g1).all() and torch.isfinite(p1).all()
    _, _, ga = grads(X[:32], Y[:32])
    _, _, gb = grads(X[32:], Y[32:])
    assert rel(0.5 * (ga + gb), g1) < 2e-2
    # rows 0-1 as a batch of their own (other kernels are picked at that size): same prediction to bf16
    # rounding, and the norm the reference computed for exactly these two inputs
    with torch.no_grad():
        p_two = model(X[:2]).clone()
    assert rel(p1[:2], p_two) < BAR["bf16"]["out"]
    assert abs(p1[:2].double().norm().item() - float(g["prednorm"])) <= BAR["bf16"]["out"] * float(g["prednorm"])
    pv = p1[:2].cpu().flatten()[torch.from_numpy(g["predidx"])]
    assert rel(pv, torch.from_numpy(g["predval"])) < 3 * BAR["bf16"]["out"]
    # the loss of those two rows against the targets the fixture's run used (drawn right after ITS two inputs)
    _, Y2 = O.synthetic_batch(2, cin, cout, 128, 128)
    loss2 = crit(p1[:2].contiguous(), Y2.cuda())
    assert abs(loss2.item() - float(g["loss0"])) <= BAR["bf16"]["out"] * float(g["loss0"])
    del model
    torch.cuda.empty_cache()


# Loss-curve bars of the long run: deviations grow with the number of steps (every step's rounding
# differences feed the next), so they are looser than the 20-step bars above.
# (first n steps, bar), about three times what was observed on MI355X: fp32 1.2e-7 / 4.4e-6 / 1.6e-2 over the first
# 20 / 120 / 240 steps (past step ~200 two fp32 implementations of the same arithmetic drift apart by themselves: the
# loop amplifies every rounding difference), bf16 4.0e-4 over the first 20 steps and 2.8e-2 overall (worst at step 26,
# on the steep part of the curve).  Final test sums: fp32 5e-3 / 1e-2, bf16 2e-3 / 4e-3 (|d| / d^2).
CURVE_BAR = {"fp32": [(20, 1e-6), (120, 2e-5), (240, 5e-2)], "bf16": [(20, 1.5e-3), (240, 8e-2)]}
CURVE_SUMS_BAR = 3e-2


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_long_loss_curve_tracks_the_reference(ua, golden, dtype):
    """240 Adam steps at lr 1e-3 over four batches in rotation, recorded from the reference (the loss falls
    13-fold): every step's loss within the bar of the reference's, and the run must really have learned."""
    from conftest import curve_batches
    g = golden("curve240")
    cin, cout, feats, _ = meta(g)
    batches = [(X.cuda(), Y.cuda()) for X, Y in curve_batches(g)]
    torch.manual_seed(0)
    model = ua.UNet(cin, cout, feats, compute_dtype=dtype).to("cuda")
    opt = ua.FusedAdam(model.parameters(), lr=float(g["meta.lr"]), model=model)
    crit = ua.MSELoss()
    losses = []
    model.train()
    for s in range(len(g["losses"])):
        X, Y = batches[s % len(batches)]
        loss = crit(model(X), Y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    losses = torch.stack(losses).cpu().double().numpy()
    ref = np.asarray(g["losses"])
    dev = np.abs(losses - ref) / np.abs(ref)
    print(f"curve240[{dtype}]: max rel dev over the first 20 / 60 / 120 / 240 steps "
          f"{dev[:20].max():.3e} / {dev[:60].max():.3e} / {dev[:120].max():.3e} / {dev.max():.3e} (at step {dev.argmax()}), "
          f"mean {dev.mean():.3e}, final {losses[-1]:.5f} vs {ref[-1]:.5f}")
    for horizon, bar in CURVE_BAR[dtype]:
        assert dev[:horizon].max() < bar, (horizon, dev[:horizon].max(), int(dev[:horizon].argmax()))
    assert losses[-1] < 0.1 * losses[0]
    model.eval()
    a = s2 = 0.0
    with torch.no_grad():
        for X, Y in batches:
            sums, _ = ua.eval_sums(model(X), Y)
            a += sums[0].item(); s2 += sums[1].item()
    e_abs = abs(a - float(g["eval.abs_sum"])) / float(g["eval.abs_sum"])
    e_sq = abs(s2 - float(g["eval.sq_sum"])) / float(g["eval.sq_sum"])
    print(f"curve240[{dtype}]: final test sums rel dev |d| {e_abs:.3e}, d^2 {e_sq:.3e}")
    assert e_abs <= CURVE_SUMS_BAR and e_sq <= CURVE_SUMS_BAR


def test_k_split_across_workgroups_is_repeatable_and_leaves_its_counters_at_zero(ua):
    """Few-pixel layers with K >= 4096 (default widths, one 64x64 matrix: the bottleneck's second conv and
    the first decoder conv) split K across workgroups and exchange partial tiles through workspace scratch.
    One workspace is driven through alternating inputs and weights: the same (input, weights) must predict
    the same bits every time (a stale partial tile or a miscounted ticket would not), the split must agree
    with the unsplit kernel to fp32 summation order, and the arrival counters must read zero between launches."""
    import ctypes as C
    from unet_amd._lib import MI_OPT_NO_KSPLIT
    L = ua.lib()
    state = O.init_state_dict(12, 1, (32, 64, 128, 256), seed=0)
    model = ua.UNet(12, 1, (32, 64, 128, 256), compute_dtype="bf16")
    model.load_state_dict(state)
    model = model.to("cuda").eval()
    Xs = [O.synthetic_batch(b, 12, 1, 64, 64, seed=20 + i)[0].cuda() for i, b in enumerate((1, 1, 2))]

    def counters(plan):
        off, cb, tot = C.c_size_t(), C.c_size_t(), C.c_size_t()
        assert L.mi_unet_plan_ksplit_region(plan.handle, C.byref(off), C.byref(cb), C.byref(tot)) == 0
        assert tot.value > cb.value > 0, "this shape was expected to carry K-split scratch"
        return plan.ws[off.value: off.value + cb.value]

    first = {}
    with torch.no_grad():
        w = model.bottleneck[2].weight
        weights = {1.0: w.detach().clone(), 1.25: (w * 1.25).detach().clone()}
        for rnd in range(4):
            for i, X in enumerate(Xs):
                for scale in (1.0, 1.25):
                    w.copy_(weights[scale])                     # the version counter moves: the weights are re-packed
                    pred = model(X).clone()
                    assert int(counters(model.last_plan).to(torch.int32).abs().sum().item()) == 0
                    key = (i, scale)
                    if key in first:
                        assert torch.equal(pred, first[key]), (rnd, key)
                    else:
                        first[key] = pred
        assert not torch.equal(first[(0, 1.0)], first[(0, 1.25)]) and not torch.equal(first[(0, 1.0)], first[(1, 1.0)])
        # the unsplit kernel on a plan of its own: same sums in another order
        other = ua.UNet(12, 1, (32, 64, 128, 256), compute_dtype="bf16")
        other.load_state_dict(state)
        other = other.to("cuda").eval()
        other(Xs[0])
        assert L.mi_unet_plan_set_option(other.last_plan.handle, MI_OPT_NO_KSPLIT, 1) == 0
        for i, X in enumerate(Xs[:2]):
            assert rel(other(X), first[(i, 1.0)]) < 2e-3, i
    ref = O.unet_forward(state, Xs[0].cpu())
    assert rel(first[(0, 1.0)].cpu(), ref) < BAR["bf16"]["out"]


@pytest.mark.parametrize("B", [1, 2])
def test_few_pixel_conv_pools_behind_a_k_split(ua, B):
    """A 5-level net on one or two 128x128 matrices: the deepest encoder conv (512 -> 512 channels on 8x8
    images, K = 4608) runs on the few-pixel kernel, with one image also split in K across workgroups, and
    writes the 2x2 max-pool of its output itself -- only the workgroup that arrives last holds the sums.
    Every pooled tensor must equal the max-pool of the tensor it was taken from, bit for bit, twice in a row
    (the second pass finds the arrival counters where the first left them), and the prediction the oracle's."""
    import ctypes as C
    feats = (32, 64, 128, 256, 512)
    state = O.init_state_dict(12, 1, feats, seed=2)
    model = ua.UNet(12, 1, feats, compute_dtype="bf16")
    model.load_state_dict(state)
    model = model.to("cuda").eval()
    X = O.synthetic_batch(B, 12, 1, 128, 128, seed=31)[0]
    with torch.no_grad():
        for rnd in range(2):
            pred = model(X.cuda())
            names = model.last_plan.tensors()
            for l in range(len(feats)):
                src = model.workspace_tensor("act.downs.%d.3" % l)
                assert torch.equal(model.workspace_tensor("act.pool.%d" % l), torch.nn.functional.max_pool2d(src, 2)), (rnd, l)
            assert "act.pool.4" in names
    if B == 1:
        off, cb, tot = C.c_size_t(), C.c_size_t(), C.c_size_t()
        assert ua.lib().mi_unet_plan_ksplit_region(model.last_plan.handle, C.byref(off), C.byref(cb), C.byref(tot)) == 0
        assert tot.value > cb.value > 0, "this shape was expected to carry K-split scratch"
        assert int(model.last_plan.ws[off.value: off.value + cb.value].to(torch.int32).abs().sum().item()) == 0
    assert rel(pred.cpu(), O.unet_forward(state, X)) < BAR["bf16"]["out"]
