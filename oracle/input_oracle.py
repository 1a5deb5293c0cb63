"""CPU oracle for the input pipeline and the train/test harness.  TEST INFRASTRUCTURE ONLY.

Plain numpy / stock torch restatement of what the reference does between a raw 1-D traffic
series and the tensors its training loop consumes, and of the loop itself.  Only ``tests/``
may import this file; nothing in the product package does.

Parity status: PINNED.  ``tests/golden/make_golden_input.py`` runs the reference's own
``utils/image_transform.py`` and ``dataset.py`` functions (and scikit-learn's MinMaxScaler, the
third-party class the reference calls) on a seeded series and stores their outputs;
``tests/test_input_oracle.py`` requires this restatement to reproduce them exactly
(``numpy.array_equal``).

What is restated (reference file:line):
  * MinMax scaling                      dataset.py:323-327 (sklearn MinMaxScaler 1.x, feature_range (0, 1):
                                        scale = 1 / (max - min) [1 when the range is 0], offset = -min * scale,
                                        x * scale + offset in float64, two rounded operations)
  * sliding windows -> image pairs      utils/image_transform.py:11-65 (gray2d)
  * K consecutive windows per sample    dataset.py:151-186
  * random split, seed 42               dataset.py:367, :395-412 (single windows), :369-390 (sequences)
  * epoch loop, checkpoint, early stop  train.py:378-411, :499-540
  * test metrics                        test.py:24-47
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np


# --------------------------------------------------------------------------- scaling
def minmax_params(values: np.ndarray) -> Tuple[float, float]:
    """(scale, offset) of MinMaxScaler().fit(values) for one feature (dataset.py:326-327)."""
    v = np.asarray(values, dtype=np.float64).reshape(-1)
    lo, hi = np.nanmin(v), np.nanmax(v)
    rng = hi - lo
    if rng < 10 * np.finfo(np.float64).eps:   # sklearn's _handle_zeros_in_scale
        rng = 1.0
    scale = 1.0 / rng
    return float(scale), float(0.0 - lo * scale)


def minmax_apply(values: np.ndarray, scale: float, offset: float) -> np.ndarray:
    v = np.array(values, dtype=np.float64).reshape(-1)
    v *= scale
    v += offset
    return v


# --------------------------------------------------------------------------- windows
def num_windows(n: int, H_in: int, W_in: int, H_out: int, W_out: int, overlap_len: int, stride: int) -> int:
    """Number of (input, target) window pairs (image_transform.py:27-47); raises like the reference."""
    input_len, output_len = H_in * W_in, H_out * W_out
    if overlap_len < 0 or overlap_len > min(input_len, output_len):
        raise ValueError(f"overlap_len must be in [0, {min(input_len, output_len)}], got {overlap_len}")
    if stride <= 0:
        raise ValueError("stride must be positive")
    max_start = n - (input_len + output_len - overlap_len)
    if max_start <= 0:
        raise ValueError("Time series too short for given window sizes and overlap. "
                         f"len={n}, input_len={input_len}, output_len={output_len}, overlap_len={overlap_len}")
    return max_start // stride + 1


def image_pairs(series_norm: np.ndarray, H_in: int, W_in: int, H_out: int, W_out: int, overlap_len: int,
                stride: int) -> Tuple[np.ndarray, np.ndarray]:
    """X (N,H_in,W_in), Y (N,H_out,W_out) float32 (image_transform.py:46-65)."""
    s = np.asarray(series_norm).reshape(-1)
    N = num_windows(len(s), H_in, W_in, H_out, W_out, overlap_len, stride)
    input_len, output_len = H_in * W_in, H_out * W_out
    X = np.empty((N, H_in, W_in), dtype=np.float32)
    Y = np.empty((N, H_out, W_out), dtype=np.float32)
    for w in range(N):
        start = w * stride
        X[w] = s[start:start + input_len].reshape(H_in, W_in)
        o = start + input_len - overlap_len
        Y[w] = s[o:o + output_len].reshape(H_out, W_out)
    return X, Y


def sequences(X: np.ndarray, Y: np.ndarray, K: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """X (N,C,H,W) -> X_seq (N-K+1,K,C,H,W), Y_seq, orig_idx (dataset.py:151-186)."""
    N = X.shape[0]
    if N < K:
        raise ValueError(f"Not enough windows ({N}) to build sequences with K={K}")
    Xs = np.stack([X[t - K + 1:t + 1] for t in range(K - 1, N)], axis=0)
    Ys = np.stack([Y[t] for t in range(K - 1, N)], axis=0)
    return Xs, Ys, np.arange(K - 1, N, dtype=np.int64)


def split_ids(n: int, test_ratio: float, val_ratio: float, seed: int = 42):
    """(train, val, test) positions of a random split of n samples (dataset.py:367, :397-404)."""
    perm = np.random.RandomState(seed).permutation(n)
    test_size, val_size = int(n * test_ratio), int(n * val_ratio)
    return perm[test_size + val_size:], perm[test_size:test_size + val_size], perm[:test_size]


def batch_tensors(series: np.ndarray, targets: np.ndarray, K: int, H_in: int, W_in: int, H_out: int, W_out: int,
                  overlap_len: int, stride: int) -> Tuple[np.ndarray, np.ndarray]:
    """The network's input for the samples whose TARGET windows are `targets` (original window
    indices): X (B, K, H_in, W_in) with the K context windows t-K+1..t as channels, Y (B, 1, H_out,
    W_out) (dataset.py:176-183 stacked on the channel axis, :221-223 for Y)."""
    scale, offset = minmax_params(series)
    X, Y = image_pairs(minmax_apply(series, scale, offset), H_in, W_in, H_out, W_out, overlap_len, stride)
    t = np.asarray(targets, dtype=np.int64)
    Xb = np.stack([X[i - K + 1:i + 1] for i in t], axis=0)
    return Xb, Y[t][:, None]


# --------------------------------------------------------------------------- harness
class OracleNet:
    """The oracle network as an object with the few nn.Module methods the loops use."""

    def __init__(self, state, out_size=None):
        from collections import OrderedDict
        self.params = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in state.items())
        self.out_size = out_size

    def __call__(self, x):
        from . import unet_oracle
        return unet_oracle.unet_forward(self.params, x, self.out_size)

    def parameters(self):
        return list(self.params.values())

    def state_dict(self):
        from collections import OrderedDict
        return OrderedDict((k, v.detach()) for k, v in self.params.items())

    def train(self):
        return self

    def eval(self):
        return self


def fit(model, optimizer, criterion, train_batches, val_batches, num_epochs: int, early_stop: bool = False,
        patience: Optional[int] = None, min_delta: float = 0.0) -> Dict[str, object]:
    """The reference's epoch loop on already materialised batches (train.py:378-411, :499-540).
    `train_batches` is a callable epoch -> list of (X, Y); returns the loss history, the best
    checkpoint dict and the epoch at which the loop stopped."""
    import torch

    best_val, no_improve, best_ckpt = float("inf"), 0, None
    hist: List[Tuple[float, float]] = []
    use_early = early_stop and patience is not None
    last = 0
    for epoch in range(1, num_epochs + 1):
        last = epoch
        model.train()
        tot = 0.0
        batches = train_batches(epoch)
        for X, Y in batches:
            loss = criterion(model(X), Y)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            tot += loss.item()
        train_loss = tot / len(batches)
        model.eval()
        tot = 0.0
        with torch.no_grad():
            for X, Y in val_batches:
                tot += criterion(model(X), Y).item()
        val_loss = tot / len(val_batches)
        hist.append((train_loss, val_loss))
        if val_loss < best_val - min_delta:
            best_val, no_improve = val_loss, 0
            best_ckpt = {"model_state_dict": {k: v.detach().clone() for k, v in model.state_dict().items()},
                         "best_val_loss": best_val, "epoch": epoch}
        else:
            no_improve += 1
        if use_early and no_improve >= patience:
            break
    return {"history": hist, "best": best_ckpt, "stopped_at": last}


def test_metrics(model, batches) -> Tuple[float, float]:
    """(MAE, MSE) over all points in normalised scale (test.py:24-47)."""
    import torch

    model.eval()
    mae = mse = 0.0
    n = 0
    with torch.no_grad():
        for X, Y in batches:
            d = model(X) - Y
            mae += d.abs().sum().item()
            mse += (d ** 2).sum().item()
            n += d.numel()
    return mae / n, mse / n


# --------------------------------------------------------------------------- series reconstruction
def reconstruct_series(preds: np.ndarray, gts: np.ndarray, idx: np.ndarray, input_len: int, output_len: int,
                       stride: int):
    """Overlap-average of predicted horizons back into one series (visualize_results.py:443-476):
    window g's horizon is laid at raw positions [g*stride + input_len, + output_len) (the reference
    places it right after the input window, whatever the overlap), sums and counts are kept per
    position, and the positions with at least one prediction are returned compressed.
    Returns (pred_series, gt_series, mask) in float64."""
    idx = np.asarray(idx)
    total = int(idx.max()) * stride + input_len + output_len
    ps, gs = np.zeros(total, dtype=np.float64), np.zeros(total, dtype=np.float64)
    cnt = np.zeros(total, dtype=np.int32)
    for i in range(preds.shape[0]):
        a = int(idx[i]) * stride + input_len
        ps[a:a + output_len] += preds[i, 0].reshape(-1)
        gs[a:a + output_len] += gts[i, 0].reshape(-1)
        cnt[a:a + output_len] += 1
    mask = cnt > 0
    return ps[mask] / cnt[mask], gs[mask] / cnt[mask], mask
