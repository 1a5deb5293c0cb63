"""Train / evaluate loops with the reference's bookkeeping and no per-step host sync.

Mirrors train.py:378-411 (``train_one_epoch``, ``evaluate``), train.py:499-540 (best-validation
checkpoint ``{"model_state_dict", "best_val_loss", "epoch"}``, early stopping on
``early_stop`` / ``early_stop_patience`` / ``early_stop_min_delta``) and test.py:24-47
(``compute_test_metrics``).  The reference calls ``loss.item()`` after every step, which stalls
the GPU once per batch; here the per-batch losses are accumulated in a device tensor and read
once per epoch.  The sum is formed in float64 in batch order, like the reference's Python float
accumulation, so the epoch averages agree to the last digit it prints.

Batches come from any iterable of device ``(X, Y)`` pairs, e.g. ``SeriesWindows.batches``."""
from __future__ import annotations

import os
from typing import Callable, Dict, Iterable, List, Optional, Tuple

import torch


def _accumulate(acc: Optional[torch.Tensor], loss: torch.Tensor) -> torch.Tensor:
    v = loss.detach().double().reshape(1)
    return v if acc is None else acc + v


def train_one_epoch(model, batches: Iterable, optimizer, criterion) -> float:
    """train.py:378-395 without the per-step ``.item()``."""
    model.train()
    acc, n = None, 0
    # the library's own loss on the library's own network: the whole iteration as direct calls
    # (UNet.train_step, bit-identical to the five lines below and without their host overhead)
    from .loss import MSELoss
    fused = hasattr(model, "train_step") and type(criterion) is MSELoss
    for X, Y in batches:
        if fused:
            loss = model.train_step(X, Y, optimizer)
        else:
            pred = model(X)
            loss = criterion(pred, Y)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
        acc = _accumulate(acc, loss)
        n += 1
    if n == 0:
        raise ValueError("no training batches")
    return float(acc.item()) / n


def evaluate(model, batches: Iterable, criterion) -> float:
    """train.py:398-411."""
    model.eval()
    acc, n = None, 0
    with torch.no_grad():
        for X, Y in batches:
            acc = _accumulate(acc, criterion(model(X), Y))
            n += 1
    if n == 0:
        raise ValueError("no evaluation batches")
    return float(acc.item()) / n


def compute_test_metrics(model, batches: Iterable) -> Tuple[float, float]:
    """(MAE, MSE) over every predicted point in normalised scale (test.py:24-47)."""
    from .loss import eval_sums

    model.eval()
    acc, n_points = None, 0
    with torch.no_grad():
        for X, Y in batches:
            sums, n = eval_sums(model(X), Y)                     # [sum |d|, sum d^2] from one kernel pass
            sums = sums.double()
            acc = sums if acc is None else acc + sums
            n_points += n
    if n_points == 0:
        raise ValueError("no test batches")
    mae, mse = (float(v) / n_points for v in acc.cpu())
    return mae, mse


def fit(model, optimizer, criterion, train_batches: Callable[[int], Iterable], val_batches: Callable[[int], Iterable],
        num_epochs: int, ckpt_path: Optional[str] = None, early_stop: bool = False, patience: Optional[int] = None,
        min_delta: float = 0.0, log: Optional[Callable[[str], None]] = print) -> Dict[str, object]:
    """The epoch loop of train.py:499-540.  ``train_batches(epoch)`` / ``val_batches(epoch)`` return the
    epoch's batch iterables.  Writes the best-validation checkpoint to ``ckpt_path`` in the reference's
    format and returns {"history": [(train, val), ...], "best_val_loss", "best_epoch", "stopped_at"}."""
    best_val, no_improve, best_epoch = float("inf"), 0, 0
    use_early = bool(early_stop) and patience is not None
    history: List[Tuple[float, float]] = []
    stopped_at = 0
    if ckpt_path:
        os.makedirs(os.path.dirname(os.path.abspath(ckpt_path)), exist_ok=True)
    for epoch in range(1, num_epochs + 1):
        stopped_at = epoch
        train_loss = train_one_epoch(model, train_batches(epoch), optimizer, criterion)
        val_loss = evaluate(model, val_batches(epoch), criterion)
        history.append((train_loss, val_loss))
        if log:
            log(f"Epoch [{epoch}/{num_epochs}] Train Loss: {train_loss:.6f} | Val Loss: {val_loss:.6f}")
        if val_loss < best_val - min_delta:
            best_val, no_improve, best_epoch = val_loss, 0, epoch
            if ckpt_path:
                torch.save({"model_state_dict": model.state_dict(), "best_val_loss": best_val, "epoch": epoch}, ckpt_path)
                if log:
                    log(f"  -> Saved best model to {ckpt_path}")
        else:
            no_improve += 1
        if use_early and no_improve >= patience:
            if log:
                log(f"Early stopping triggered at epoch {epoch}. "
                    f"No improvement in validation loss for {patience} consecutive epochs.")
            break
    return {"history": history, "best_val_loss": best_val, "best_epoch": best_epoch, "stopped_at": stopped_at}


def fit_from_cfg(cfg, series, splits=None, model=None, log: Optional[Callable[[str], None]] = print) -> Dict[str, object]:
    """``train.py:main`` for the U-Net route on a ``SeriesWindows``: model from ``build_model_from_cfg``,
    Adam(lr=cfg.lr), MSE, batches of ``cfg.batch_size`` shuffled like the reference's DataLoader, the
    checkpoint at ``cfg.model_ckpt``."""
    from . import FusedAdam, MSELoss, build_model_from_cfg

    train_ids, val_ids, _ = splits if splits is not None else series.split(
        getattr(cfg, "test_ratio", 0.2), getattr(cfg, "val_ratio", 0.1), getattr(cfg, "random_seed", 42))
    if model is None:
        model = build_model_from_cfg(cfg).to(series.device)
    optimizer = FusedAdam(model.parameters(), lr=cfg.lr, model=model)
    criterion = MSELoss()
    tr, va = series.to_device_ids(train_ids), series.to_device_ids(val_ids)
    bs = cfg.batch_size
    res = fit(model, optimizer, criterion,
              lambda epoch: series.batches(tr, bs, shuffle=True), lambda epoch: series.batches(va, bs, shuffle=False),
              cfg.num_epochs, getattr(cfg, "model_ckpt", None), getattr(cfg, "early_stop", False),
              getattr(cfg, "early_stop_patience", None), getattr(cfg, "early_stop_min_delta", 0.0), log)
    res["model"] = model
    return res
