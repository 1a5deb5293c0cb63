"""Input pipeline with the series resident on the GPU.

The reference materialises every (input, target) image pair on the host, stores them in an
``.npz`` and copies each batch to the device (dataset.py:295-487, train.py:383-384).  Here the
raw 1-D series is uploaded once; a batch is gathered from it by one kernel
(``mi_window_batch``) given the indices of its samples, with the MinMax scaling
(dataset.py:326-327) applied on the fly in float64 exactly as scikit-learn does, so the
tensors are bit-identical to the reference's.  Per step the host sends nothing (the index
permutation of an epoch lives on the device too).

    series = SeriesWindows(values, H_in=15, W_in=24, H_out=15, W_out=24, overlap_len=72, stride=1)
    train, val, test = series.split(test_ratio=0.2, val_ratio=0.1, seed=42)   # dataset.py:367-412
    for X, Y in series.batches(train, batch_size=128, shuffle=True):           # device tensors
        ...
"""
from __future__ import annotations

from typing import Iterator, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import check, lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def reference_shuffle(n: int) -> torch.Tensor:
    """The permutation ``DataLoader(shuffle=True)`` (dataset.py:561) would draw next from torch's
    global generator: opening the loader draws a base seed, then the sampler draws the seed of a
    private generator for ``randperm``.  Consuming the same draws keeps a run reproducible against
    the reference under the same ``torch.manual_seed``."""
    torch.empty((), dtype=torch.int64).random_()                       # the iterator's base seed
    seed = int(torch.empty((), dtype=torch.int64).random_().item())    # RandomSampler's generator seed
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g)


def reference_open_loader() -> None:
    """The draw a non-shuffling ``DataLoader`` iterator makes when it is opened (dataset.py:562-563)."""
    torch.empty((), dtype=torch.int64).random_()


class SeriesWindows:
    """A traffic series on the GPU and the window geometry that turns it into samples.

    ``context`` (K) consecutive windows are stacked as input channels (dataset.py:176-183); with the
    default ``context=1`` a sample is the single window the reference's U-Net path uses.  A sample is
    identified by the ORIGINAL index t of its target window, ``context-1 <= t < num_windows``, the
    same ids the reference stores as ``train_idx`` / ``val_idx`` / ``test_idx``."""

    def __init__(self, values, H_in: int, W_in: int, H_out: int, W_out: int, overlap_len: int, stride: int = 1,
                 context: int = 1, device="cuda"):
        v = np.ascontiguousarray(np.asarray(values, dtype=np.float64).reshape(-1))
        self.H_in, self.W_in, self.H_out, self.W_out = int(H_in), int(W_in), int(H_out), int(W_out)
        self.overlap_len, self.stride, self.context = int(overlap_len), int(stride), int(context)
        self.n = int(v.shape[0])
        L = lib()
        N = L.mi_window_count(self.n, self.H_in, self.W_in, self.H_out, self.W_out, self.overlap_len, self.stride)
        if N < 0:
            raise ValueError(L.mi_last_error().decode())          # the reference's messages (image_transform.py:27-44)
        if N < self.context:
            raise ValueError(f"Not enough windows ({N}) to build sequences with K={self.context}")   # dataset.py:168-169
        self.num_windows = int(N)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("SeriesWindows needs a GPU device: the pipeline has no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.series = torch.from_numpy(v).to(self.device)
        # MinMaxScaler.fit on the device, the two parameters derived on the host in float64
        ws = torch.empty(L.mi_series_minmax_ws_bytes(), dtype=torch.uint8, device=self.device)
        mm = torch.empty(2, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(L.mi_series_minmax(self.series.data_ptr(), self.n, mm.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
                  "series_minmax")
        lo, hi = (float(t) for t in mm.cpu())
        self.data_min, self.data_max = lo, hi
        rng = hi - lo
        if rng < 10 * np.finfo(np.float64).eps:                   # sklearn: a constant feature keeps scale 1
            rng = 1.0
        self.scale = 1.0 / rng
        self.offset = 0.0 - lo * self.scale

    # ------------------------------------------------------------------ sample ids and splits
    @property
    def sample_ids(self) -> np.ndarray:
        """Target-window index of every sample (dataset.py:176-184: t = K-1 .. N-1)."""
        return np.arange(self.context - 1, self.num_windows, dtype=np.int64)

    def split(self, test_ratio: float = 0.2, val_ratio: float = 0.1, seed: int = 42):
        """(train, val, test) sample ids of the reference's random split (dataset.py:367-412)."""
        ids = self.sample_ids
        n = ids.shape[0]
        perm = np.random.RandomState(seed).permutation(n)
        test_size, val_size = int(n * test_ratio), int(n * val_ratio)
        return ids[perm[test_size + val_size:]], ids[perm[test_size:test_size + val_size]], ids[perm[:test_size]]

    def to_device_ids(self, ids) -> torch.Tensor:
        """Validates sample ids once on the host and uploads them (int64)."""
        a = np.asarray(ids, dtype=np.int64).reshape(-1)
        if a.size and (a.min() < self.context - 1 or a.max() >= self.num_windows):
            raise IndexError(f"sample ids must lie in [{self.context - 1}, {self.num_windows})")
        return torch.from_numpy(a).to(self.device)

    # ------------------------------------------------------------------ batches
    def batch(self, ids, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """X (B, context, H_in, W_in), Y (B, 1, H_out, W_out), fp32 on the device.  `ids` is a device
        int64 tensor from ``to_device_ids`` (not re-validated) or anything array-like on the host."""
        if not (torch.is_tensor(ids) and ids.device == self.device and ids.dtype == torch.int64):
            ids = self.to_device_ids(ids.cpu().numpy() if torch.is_tensor(ids) else ids)
        ids = ids.contiguous()
        B = int(ids.numel())
        if B == 0:
            raise ValueError("empty batch")
        if out is None:
            X = torch.empty((B, self.context, self.H_in, self.W_in), dtype=torch.float32, device=self.device)
            Y = torch.empty((B, 1, self.H_out, self.W_out), dtype=torch.float32, device=self.device)
        else:
            X, Y = out
        with torch.cuda.device(self.device):
            check(lib().mi_window_batch(self.series.data_ptr(), self.n, ids.data_ptr(), B, self.context, self.H_in,
                                        self.W_in, self.H_out, self.W_out, self.overlap_len, self.stride, self.scale,
                                        self.offset, X.data_ptr(), Y.data_ptr(), _stream()), "window_batch")
        return X, Y

    def batches(self, ids, batch_size: int, shuffle: bool = False,
                order: Optional[Sequence[int]] = None) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        """One epoch over `ids` in batches (the last one may be short, drop_last=False as in
        dataset.py:561-563).  shuffle=True draws the order exactly as the reference's DataLoader would
        from torch's global generator; `order` (positions into `ids`) overrides it."""
        dev_ids = ids if torch.is_tensor(ids) and ids.device == self.device else self.to_device_ids(ids)
        n = int(dev_ids.numel())
        if order is not None:
            perm = torch.as_tensor(np.asarray(order, dtype=np.int64))
        elif shuffle:
            perm = reference_shuffle(n)
        else:
            reference_open_loader()
            perm = None
        if perm is not None:
            dev_ids = dev_ids[perm.to(self.device)]
        for lo in range(0, n, batch_size):
            yield self.batch(dev_ids[lo:lo + batch_size])

    def num_batches(self, ids, batch_size: int) -> int:
        n = int(ids.numel()) if torch.is_tensor(ids) else len(ids)
        return (n + batch_size - 1) // batch_size

    # ------------------------------------------------------------------ scaler
    def inverse_transform(self, t: torch.Tensor) -> torch.Tensor:
        """MinMaxScaler.inverse_transform (used by test.py:75-76): (x - offset) / scale."""
        return (t.double() - self.offset) / self.scale


def reconstruct_series(pred: torch.Tensor, gt: torch.Tensor, idx, input_len: int, output_len: int, stride: int):
    """Overlap-average of predicted horizons into one series, as visualize_results.py:443-476 does on
    the host: ``pred`` / ``gt`` are (N, 1, H_out, W_out) device tensors of the windows ``idx`` (original
    window indices, distinct).  Returns (pred_series, gt_series, mask): float64 device tensors holding
    the positions covered by at least one horizon (the reference's compressed series) and the boolean
    mask of those positions over the raw index space."""
    ids = np.asarray(idx.cpu().numpy() if torch.is_tensor(idx) else idx, dtype=np.int64).reshape(-1)
    N = int(pred.shape[0])
    if ids.shape[0] != N or gt.shape[0] != N or N == 0:
        raise ValueError("reconstruct_series: one window index per prediction row is required")
    if ids.min() < 0 or np.unique(ids).shape[0] != N:
        raise ValueError("reconstruct_series: window indices must be distinct and non-negative")
    if int(np.prod(pred.shape[1:])) != output_len or tuple(gt.shape) != tuple(pred.shape):
        raise ValueError("reconstruct_series: rows must hold output_len values")
    if not pred.is_cuda:
        raise RuntimeError("reconstruct_series needs device tensors: there is no CPU path")
    dev = pred.device
    n_slots = int(ids.max()) + 1
    slot = np.full(n_slots, -1, dtype=np.int32)
    slot[ids] = np.arange(N, dtype=np.int32)
    total = int(ids.max()) * stride + input_len + output_len
    p = pred.detach().float().contiguous()
    g = gt.detach().float().contiguous()
    slot_d = torch.from_numpy(slot).to(dev)
    pm = torch.empty(total, dtype=torch.float64, device=dev)
    gm = torch.empty(total, dtype=torch.float64, device=dev)
    cnt = torch.empty(total, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib().mi_series_reconstruct(p.data_ptr(), g.data_ptr(), slot_d.data_ptr(), n_slots, N, input_len, output_len,
                                          stride, total, pm.data_ptr(), gm.data_ptr(), cnt.data_ptr(), _stream()),
              "series_reconstruct")
    mask = cnt > 0
    return pm[mask], gm[mask], mask
