"""CPU oracle for the U-Net traffic-matrix hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, with stock ``torch`` CPU ops, the algorithm of the reference
network so the HIP path can be checked against it.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; nothing in the product package does.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
model from its own files, records inputs / outputs / gradients / loss curves,
and ``tests/test_oracle_golden.py`` requires this restatement to reproduce those
fixtures bit for bit (``torch.equal``).

What is restated (reference file:line):
  * parameter set, shapes and key names          models/unet2d.py:20-59
  * default init and RNG consumption order       models/unet2d.py:24-59 (construction order:
    downs -> bottleneck -> ups -> final_conv; every layer uses torch's default
    ``_ConvNd.reset_parameters``: kaiming_uniform(a=sqrt(5)) + U(+-1/sqrt(fan_in)) bias)
  * forward                                       models/unet2d.py:61-102
  * loss / optimiser step                         train.py:382-393, train.py:497-498
  * evaluation sums                               test.py:33-47

The arithmetic of each op (conv, max-pool, transposed conv, bilinear resize) is
third-party PyTorch/ATen, version 2.10.0 in this image; the functional calls below
are the same ATen ops the reference's ``nn.Module`` objects dispatch to.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

DEFAULT_FEATURES = (32, 64, 128, 256)  # models/unet2d.py:13


# --------------------------------------------------------------------------- parameters
def construction_order(in_channels: int, out_channels: int, features: Sequence[int]):
    """(key_prefix, kind, cin, cout) in the order the reference constructs layers
    (models/unet2d.py:24-59) -- this is also the order the global RNG is consumed in."""
    order = []
    ch = in_channels
    for i, f in enumerate(features):  # :26-34
        order.append((f"downs.{i}.0", "conv3", ch, f))
        order.append((f"downs.{i}.2", "conv3", f, f))
        ch = f
    fb = features[-1]
    order.append(("bottleneck.0", "conv3", fb, 2 * fb))  # :37-42
    order.append(("bottleneck.2", "conv3", 2 * fb, 2 * fb))
    for j, f in enumerate(reversed(features)):  # :45-56
        order.append((f"ups.{2 * j}", "convt2", 2 * f, f))
        order.append((f"ups.{2 * j + 1}.0", "conv3", 2 * f, f))
        order.append((f"ups.{2 * j + 1}.2", "conv3", f, f))
    order.append(("final_conv", "conv1", features[0], out_channels))  # :59
    return order


def state_dict_order(in_channels: int, out_channels: int, features: Sequence[int]) -> List[str]:
    """Key prefixes in ``state_dict()`` order: the two ModuleLists are registered
    (models/unet2d.py:20-21) before ``bottleneck`` (:37), so ups precede bottleneck."""
    pre = [p for p, *_ in construction_order(in_channels, out_channels, features)]
    downs = [p for p in pre if p.startswith("downs.")]
    ups = [p for p in pre if p.startswith("ups.")]
    bott = [p for p in pre if p.startswith("bottleneck.")]
    return downs + ups + bott + ["final_conv"]


def _weight_shape(kind: str, cin: int, cout: int) -> Tuple[int, ...]:
    if kind == "conv3":
        return (cout, cin, 3, 3)
    if kind == "conv1":
        return (cout, cin, 1, 1)
    if kind == "convt2":
        return (cin, cout, 2, 2)  # ConvTranspose2d keeps (in, out, kh, kw)
    raise ValueError(kind)


def init_state_dict(in_channels=1, out_channels=1, features=DEFAULT_FEATURES,
                    seed: Optional[int] = None, dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    """Freshly initialised parameters, bit-identical to constructing the reference
    ``UNet`` under the same global seed."""
    if seed is not None:
        torch.manual_seed(seed)
    made: Dict[str, torch.Tensor] = {}
    for prefix, kind, cin, cout in construction_order(in_channels, out_channels, features):
        w = torch.empty(_weight_shape(kind, cin, cout), dtype=torch.float32)
        torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        # torch computes fan_in as size(1) * receptive field for every _ConvNd, so the
        # transposed conv uses its *output* channel count here.
        fan_in = w.size(1) * w.size(2) * w.size(3)
        bound = 1.0 / math.sqrt(fan_in)
        b = torch.empty(cout, dtype=torch.float32)
        torch.nn.init.uniform_(b, -bound, bound)
        made[prefix + ".weight"] = w.to(dtype)
        made[prefix + ".bias"] = b.to(dtype)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for prefix in state_dict_order(in_channels, out_channels, features):
        out[prefix + ".weight"] = made[prefix + ".weight"]
        out[prefix + ".bias"] = made[prefix + ".bias"]
    return out


def infer_arch(state: Dict[str, torch.Tensor]) -> Tuple[int, int, Tuple[int, ...]]:
    """(in_channels, out_channels, features) from parameter shapes."""
    feats = []
    i = 0
    while f"downs.{i}.0.weight" in state:
        feats.append(state[f"downs.{i}.0.weight"].shape[0])
        i += 1
    return (state["downs.0.0.weight"].shape[1], state["final_conv.weight"].shape[0], tuple(feats))


# --------------------------------------------------------------------------- single ops
def conv3x3_relu(x, w, b):
    """nn.Conv2d(k=3, padding=1) followed by nn.ReLU -- models/unet2d.py:28-31."""
    return F.relu(F.conv2d(x, w, b, stride=1, padding=1))


def maxpool2(x):
    """nn.MaxPool2d(kernel_size=2, stride=2) -- models/unet2d.py:22."""
    return F.max_pool2d(x, kernel_size=2, stride=2)


def convt2x2(x, w, b):
    """nn.ConvTranspose2d(k=2, stride=2) -- models/unet2d.py:46-48."""
    return F.conv_transpose2d(x, w, b, stride=2)


def bilinear(x, size):
    """F.interpolate(mode='bilinear', align_corners=False) -- models/unet2d.py:82-84, :95-100."""
    return F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False)


def conv1x1(x, w, b):
    """nn.Conv2d(k=1) head, no activation -- models/unet2d.py:59, :89."""
    return F.conv2d(x, w, b)


def mse_loss(pred, target):
    """nn.MSELoss() (mean over every element) -- train.py:498, :387."""
    return F.mse_loss(pred, target)


# --------------------------------------------------------------------------- forward
def unet_forward(state: Dict[str, torch.Tensor], x: torch.Tensor,
                 out_size: Optional[Tuple[int, int]] = None,
                 taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """Forward pass of models/unet2d.py:61-102 over a plain parameter dict.

    ``taps`` (optional) receives every intermediate activation under these names:
      downs.{i}.1 / downs.{i}.3   post-ReLU outputs of the two encoder convs (.3 is the skip)
      pool.{i}                    max-pool output
      bottleneck.1 / .3
      ups.{2j}                    transposed-conv output (before any resize)
      align.{j}                   bilinear-aligned tensor, only when the resize fires (:81-84)
      ups.{2j+1}.1 / .3           post-ReLU outputs of the two decoder convs
      final_conv                  1x1 head output (before any forced resize)
      out                         value returned (after the out_size resize :92-100)
    """
    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    n_levels = 0
    while f"downs.{n_levels}.0.weight" in state:
        n_levels += 1

    skips = []
    for i in range(n_levels):  # :65-68
        x = tap(f"downs.{i}.1", conv3x3_relu(x, state[f"downs.{i}.0.weight"], state[f"downs.{i}.0.bias"]))
        x = tap(f"downs.{i}.3", conv3x3_relu(x, state[f"downs.{i}.2.weight"], state[f"downs.{i}.2.bias"]))
        skips.append(x)
        x = tap(f"pool.{i}", maxpool2(x))

    x = tap("bottleneck.1", conv3x3_relu(x, state["bottleneck.0.weight"], state["bottleneck.0.bias"]))  # :70
    x = tap("bottleneck.3", conv3x3_relu(x, state["bottleneck.2.weight"], state["bottleneck.2.bias"]))

    for j in range(n_levels):  # :76-87
        skip = skips[n_levels - 1 - j]
        x = tap(f"ups.{2 * j}", convt2x2(x, state[f"ups.{2 * j}.weight"], state[f"ups.{2 * j}.bias"]))
        if x.shape[2:] != skip.shape[2:]:
            x = tap(f"align.{j}", bilinear(x, skip.shape[2:]))
        x = torch.cat([skip, x], dim=1)  # skip first, upsampled second (:86)
        x = tap(f"ups.{2 * j + 1}.1",
                conv3x3_relu(x, state[f"ups.{2 * j + 1}.0.weight"], state[f"ups.{2 * j + 1}.0.bias"]))
        x = tap(f"ups.{2 * j + 1}.3",
                conv3x3_relu(x, state[f"ups.{2 * j + 1}.2.weight"], state[f"ups.{2 * j + 1}.2.bias"]))

    x = tap("final_conv", conv1x1(x, state["final_conv.weight"], state["final_conv.bias"]))  # :89
    if out_size is not None:  # :92-100
        h, w = out_size
        if x.shape[2] != h or x.shape[3] != w:
            x = bilinear(x, (h, w))
    return tap("out", x)


# --------------------------------------------------------------------------- gradients
def forward_backward(state: Dict[str, torch.Tensor], x: torch.Tensor, y: torch.Tensor,
                     out_size: Optional[Tuple[int, int]] = None, want_input_grad: bool = False):
    """One forward + MSE + backward.  Returns (loss, taps, act_grads, param_grads[, dx]).

    ``act_grads[name]`` is dLoss/d(taps[name]) -- the total gradient reaching that
    activation (for a skip tensor: pool path + concat path summed), i.e. what a
    tensor hook on the reference's activation reports."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in state.items()}
    xin = x.detach().clone().requires_grad_(want_input_grad)
    taps: Dict[str, torch.Tensor] = {}
    pred = unet_forward(params, xin, out_size, taps)
    loss = mse_loss(pred, y)
    names = [n for n in taps if taps[n].requires_grad]
    wrt = [taps[n] for n in names] + list(params.values()) + ([xin] if want_input_grad else [])
    grads = torch.autograd.grad(loss, wrt, allow_unused=True)
    act_grads = {n: g for n, g in zip(names, grads[: len(names)])}
    pg = grads[len(names): len(names) + len(params)]
    param_grads = OrderedDict((k, g) for k, g in zip(params.keys(), pg))
    taps = {k: v.detach() for k, v in taps.items()}
    if want_input_grad:
        return loss.detach(), taps, act_grads, param_grads, grads[-1]
    return loss.detach(), taps, act_grads, param_grads


# --------------------------------------------------------------------------- train / eval loops
def train_steps(state: Dict[str, torch.Tensor], batches, lr: float = 1e-4,
                out_size: Optional[Tuple[int, int]] = None):
    """Loop body of train.py:382-393 with Adam(lr) (train.py:497) and MSE (train.py:498).

    ``batches`` is an iterable of (X, Y).  Returns (losses, final_state)."""
    params = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in state.items())
    opt = torch.optim.Adam(list(params.values()), lr=lr)
    losses = []
    for X, Y in batches:
        pred = unet_forward(params, X, out_size)
        loss = mse_loss(pred, Y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return losses, OrderedDict((k, v.detach()) for k, v in params.items())


def eval_sums(state: Dict[str, torch.Tensor], X: torch.Tensor, Y: torch.Tensor,
              out_size: Optional[Tuple[int, int]] = None):
    """Per-batch quantities of test.py:33-43: (sum |diff|, sum diff^2, numel)."""
    with torch.no_grad():
        diff = unet_forward(state, X, out_size) - Y
        return diff.abs().sum().item(), (diff ** 2).sum().item(), diff.numel()


def synthetic_batch(batch: int, cin: int, cout: int, h: int, w: int, seed: int = 1):
    """Synthetic traffic matrices in [0,1) (the reference MinMax-scales its data,
    dataset.py:326-327): X then Y drawn from one seeded CPU generator."""
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(batch, cin, h, w, generator=g)
    Y = torch.rand(batch, cout, h, w, generator=g)
    return X, Y


# --------------------------------------------------------------------------- work counts
def conv_flops_per_sample(in_channels: int, out_channels: int, features: Sequence[int],
                          h: int, w: int) -> Tuple[int, int]:
    """(forward, train) dense FLOPs per sample: 2*Cout*Hout*Wout*Cin*kh*kw per conv,
    padding taps counted; train = 3*fwd minus the first layer's dgrad.  Assumes sizes
    that halve exactly (no resize branch)."""
    fwd = 0
    first = None
    hs, ws = h, w
    dims = []
    for _ in features:
        dims.append((hs, ws))
        hs, ws = hs // 2, ws // 2
    for prefix, kind, cin, cout in construction_order(in_channels, out_channels, features):
        if prefix.startswith("downs."):
            lh, lw = dims[int(prefix.split(".")[1])]
        elif prefix.startswith("bottleneck."):
            lh, lw = hs, ws
        elif prefix.startswith("ups."):
            j = int(prefix.split(".")[1]) // 2
            lh, lw = dims[len(features) - 1 - j]
        else:
            lh, lw = h, w
        if kind == "conv3":
            fl = 2 * cout * lh * lw * cin * 9
        elif kind == "conv1":
            fl = 2 * cout * lh * lw * cin
        else:  # convt2: input is (lh/2, lw/2)
            fl = 2 * (lh // 2) * (lw // 2) * cin * cout * 4
        if first is None:
            first = fl
        fwd += fl
    return fwd, 3 * fwd - first
