"""The CPU oracle must reproduce the fixtures recorded from the reference bit for bit.

This is what pins the oracle: every later GPU parity test compares the HIP path with
the oracle (and with these same fixtures)."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O

FULL = ["small4", "small5", "odd"]
SCALAR = ["cfg1_full", "cfg2_b32", "cfg4_b2"]


def _meta(g):
    feats = tuple(int(v) for v in g["meta.features"])
    osz = tuple(int(v) for v in g["meta.out_size"])
    return int(g["meta.cin"]), int(g["meta.cout"]), feats, (None if osz[0] < 0 else osz)


@pytest.mark.parametrize("name", FULL)
def test_init_is_bit_identical(golden, name):
    g = golden(name)
    cin, cout, feats, _ = _meta(g)
    state = O.init_state_dict(cin, cout, feats, seed=0)
    keys = [k[len("param."):] for k in g.files if k.startswith("param.")]
    assert list(state.keys()) == keys  # state_dict order: downs, ups, bottleneck, final_conv
    for k in keys:
        assert torch.equal(state[k], torch.from_numpy(g["param." + k])), k


@pytest.mark.parametrize("name", FULL)
def test_forward_backward_bit_identical(golden, name):
    g = golden(name)
    cin, cout, feats, osz = _meta(g)
    state = OrderedDict((k[len("param."):], torch.from_numpy(g[k])) for k in g.files if k.startswith("param."))
    X, Y = torch.from_numpy(g["X"]), torch.from_numpy(g["Y"])
    loss, taps, act_grads, pgrads, dX = O.forward_backward(state, X, Y, osz, want_input_grad=True)
    assert loss.item() == float(g["loss0"])
    assert torch.equal(taps["out"], torch.from_numpy(g["pred"]))
    n_act = 0
    for k in g.files:
        if k.startswith("act."):
            assert torch.equal(taps[k[4:]], torch.from_numpy(g[k])), k
            n_act += 1
        elif k.startswith("actgrad."):
            assert torch.equal(act_grads[k[8:]], torch.from_numpy(g[k])), k
        elif k.startswith("grad."):
            assert torch.equal(pgrads[k[5:]], torch.from_numpy(g[k])), k
    assert n_act >= 27
    assert torch.equal(dX, torch.from_numpy(g["dX"]))
    if name == "odd":  # both bilinear branches of models/unet2d.py:81-84 and :92-100 fired
        assert any(k.startswith("act.align.") for k in g.files)
        assert tuple(taps["out"].shape[2:]) == (15, 24) and tuple(taps["final_conv"].shape[2:]) == (27, 24)


@pytest.mark.parametrize("name", FULL + SCALAR)
def test_adam_loss_curve_bit_identical(golden, name):
    g = golden(name)
    cin, cout, feats, osz = _meta(g)
    b = int(g["meta.batch"]); h, w = (int(v) for v in g["meta.hw"])
    state = O.init_state_dict(cin, cout, feats, seed=0)
    oh, ow = osz if osz else (h, w)
    gen = torch.Generator().manual_seed(1)
    X = torch.rand(b, cin, h, w, generator=gen)
    Y = torch.rand(b, cout, oh, ow, generator=gen)
    assert X.double().sum().item() == float(g["Xsum"]) and Y.double().sum().item() == float(g["Ysum"])
    n = len(g["losses"]) if name != "cfg4_b2" else 3  # keep the CPU suite short
    losses, final = O.train_steps(state, [(X, Y)] * n, lr=1e-4, out_size=osz)
    assert losses == [float(v) for v in g["losses"][:n]]
    if n == len(g["losses"]):
        a, s, cnt = O.eval_sums(final, X, Y, osz)
        assert (a, s, cnt) == (float(g["eval.abs_sum"]), float(g["eval.sq_sum"]), int(g["eval.numel"]))


def test_long_loss_curve_bit_identical(golden):
    """240 Adam steps at lr 1e-3 over four batches in rotation (the loss falls 13-fold): the oracle's loop
    reproduces every loss the reference recorded, and the final test sums."""
    from conftest import curve_batches
    g = golden("curve240")
    cin, cout, feats, _ = _meta(g)
    batches = curve_batches(g)
    n = len(g["losses"])
    state = O.init_state_dict(cin, cout, feats, seed=0)
    losses, final = O.train_steps(state, [batches[s % len(batches)] for s in range(n)], lr=float(g["meta.lr"]))
    assert losses == [float(v) for v in g["losses"]]
    assert losses[-1] < 0.1 * losses[0]
    a = s = 0.0
    for X, Y in batches:
        ai, si, _ = O.eval_sums(final, X, Y)
        a += ai; s += si
    assert (a, s) == (float(g["eval.abs_sum"]), float(g["eval.sq_sum"]))


@pytest.mark.parametrize("name", ["cfg1_full", "cfg2_b32"])
def test_full_width_scalars(golden, name):
    g = golden(name)
    cin, cout, feats, osz = _meta(g)
    b = int(g["meta.batch"]); h, w = (int(v) for v in g["meta.hw"])
    state = O.init_state_dict(cin, cout, feats, seed=0)
    for k, v in state.items():
        assert v.double().norm().item() == float(g["paramnorm." + k])
    X, Y = O.synthetic_batch(b, cin, cout, h, w)
    loss, taps, act_grads, pgrads, dX = O.forward_backward(state, X, Y, osz, want_input_grad=True)
    assert loss.item() == float(g["loss0"])
    assert torch.equal(taps["out"].flatten()[torch.from_numpy(g["predidx"])], torch.from_numpy(g["predval"]))
    for k in g.files:
        if k.startswith("actval."):
            n = k[7:]
            assert torch.equal(taps[n].flatten()[torch.from_numpy(g["actidx." + n])], torch.from_numpy(g[k])), k
            assert taps[n].double().norm().item() == float(g["actnorm." + n])
        elif k.startswith("gradval."):
            n = k[8:]
            assert torch.equal(pgrads[n].flatten()[torch.from_numpy(g["gradidx." + n])], torch.from_numpy(g[k])), k
            assert pgrads[n].double().norm().item() == float(g["gradnorm." + n])
        elif k.startswith("actgradnorm."):
            assert act_grads[k[12:]].double().norm().item() == float(g[k])
    assert dX.double().norm().item() == float(g["dXnorm"])


def test_pool_ties_known_answers(golden):
    g = golden("pool_ties")
    for name in ("even", "odd"):
        x = torch.from_numpy(g[name + ".x"]).clone().requires_grad_(True)
        y = O.maxpool2(x)
        assert torch.equal(y, torch.from_numpy(g[name + ".y"]))
        y.backward(torch.from_numpy(g[name + ".gy"]))
        assert torch.equal(x.grad, torch.from_numpy(g[name + ".gx"]))
    gx = g["even.gx"][0, 0]
    gy = g["even.gy"][0, 0]
    # first maximum in row-major order takes the whole gradient
    assert gx[0, 0] == gy[0, 0] and gx[0, 1] == 0 and gx[1, 0] == 0 and gx[1, 1] == 0   # all-zero window
    assert gx[0, 2] == gy[0, 1] and gx[0, 3] == 0 and gx[1, 2] == 0 and gx[1, 3] == 0   # 4-way tie
    assert gx[0, 5] == gy[0, 2] and gx[1, 4] == 0                                       # anti-diagonal tie
    assert gx[3, 0] == gy[1, 0] and gx[3, 1] == 0                                       # bottom-row tie
    assert gx[2, 3] == gy[1, 1] and gx[3, 3] == 0                                       # right-column tie


def test_flop_counts_of_the_benchmark_configurations():
    assert O.conv_flops_per_sample(12, 1, (32, 64, 128, 256), 16, 16) == (95_633_408, 285_130_752)
    assert O.conv_flops_per_sample(12, 1, (32, 64, 128, 256), 32, 32) == (382_533_632, 1_140_523_008)
    assert O.conv_flops_per_sample(24, 12, (32, 64, 128, 256, 512), 128, 128) == (7_671_382_016, 22_787_653_632)
    assert O.conv_flops_per_sample(12, 1, (32, 64, 128, 256), 64, 64)[0] == 1_530_134_528


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference tree only exists in the build container")
def test_oracle_equals_live_reference():
    import sys
    sys.path.insert(0, "/root/reference")
    try:
        from models.unet2d import UNet as RefUNet
    finally:
        sys.path.remove("/root/reference")
    torch.manual_seed(5)
    ref = RefUNet(3, 2, features=(8, 16, 32))
    state = O.init_state_dict(3, 2, (8, 16, 32), seed=5)
    for (k, v), (k2, v2) in zip(ref.state_dict().items(), state.items()):
        assert k == k2 and torch.equal(v, v2)
    x = torch.rand(3, 3, 24, 40)
    assert torch.equal(ref(x), O.unet_forward(state, x))
    x = torch.rand(2, 3, 21, 19)  # odd sizes: resize branch
    ref.out_size = (9, 11)
    assert torch.equal(ref(x), O.unet_forward(state, x, out_size=(9, 11)))
