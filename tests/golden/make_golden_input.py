#!/usr/bin/env python3
"""Generate the input-pipeline and harness fixtures FROM THE REFERENCE ITSELF.

Run in the build container only (the reference tree does not exist on the GPU box):

    python /root/repo/tests/golden/make_golden_input.py

It changes into a scratch directory (the reference's ``config.py`` creates ``results/`` and
``checkpoints/`` relative to the working directory), writes a seeded synthetic CSV, points the
reference's ``cfg`` at it and calls the reference's own functions:

  * ``dataset.generate_and_save_dataset()``  (CSV -> MinMaxScaler -> image pairs -> seed-42 split)
  * ``utils.image_transform.build_image_pairs`` and ``dataset._build_sequences`` on other geometries
  * ``train.main()`` and ``test.compute_test_metrics`` on a reduced-width U-Net for a few epochs on the CPU

Only DATA is stored: series, scaler parameters, selected windows, checksums, index lists, loss
values, checkpoint contents.  Files written next to this script:

  input_ref15x24.npz   the reference's default geometry through generate_and_save_dataset
  input_cases.npz      three other geometries (stride > 1, maximal overlap, K = 12)
  harness_small.npz    epoch losses, best checkpoint and test metrics of a 3-epoch reference run
"""
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def synthetic_series(n, seed):
    """Positive, bursty, float64: a random walk plus spikes, like packet counts."""
    rng = np.random.RandomState(seed)
    walk = np.cumsum(rng.randn(n)) * 40.0
    spikes = (rng.rand(n) < 0.02) * rng.rand(n) * 5000.0
    return np.abs(walk + 3000.0 + spikes + rng.rand(n) * 100.0)


def pick(n):
    return np.unique(np.array([0, 1, 2, n // 3, n // 2, n - 2, n - 1], dtype=np.int64))


def main():
    scratch = tempfile.mkdtemp(prefix="golden_input_")
    os.chdir(scratch)
    sys.path.insert(0, REF)
    import pandas as pd
    import torch
    from config import cfg                                    # noqa: E402
    import dataset as ref_dataset                             # noqa: E402
    from utils.image_transform import build_image_pairs       # noqa: E402
    from sklearn.preprocessing import MinMaxScaler

    # ---------------------------------------------------------------- default geometry, full pipeline
    series = synthetic_series(1400, 7)
    os.makedirs("data/raw", exist_ok=True)
    pd.DataFrame({cfg.value_col: series}).to_csv("in.csv", index=False)
    series = pd.read_csv("in.csv")[cfg.value_col].values      # what the reference will read back
    cfg.raw_csv_path = "in.csv"
    cfg.model_type = "unet"
    cfg.processed_dir = os.path.join(scratch, "processed")
    cfg.checkpoint_dir = os.path.join(scratch, "ckpt")
    cfg.scaler_path = os.path.join(cfg.checkpoint_dir, "scaler.pkl")
    cfg.model_ckpt = os.path.join(cfg.checkpoint_dir, "best.pt")
    ref_dataset.generate_and_save_dataset()
    d = np.load(ref_dataset._get_dataset_path(), allow_pickle=False)
    scaler = MinMaxScaler().fit(series.reshape(-1, 1))
    out = {
        "series": series, "scale": scaler.scale_, "offset": scaler.min_,
        "geometry": np.array([cfg.H_in, cfg.W_in, cfg.H_out, cfg.W_out, cfg.overlap_rows * cfg.W_in, cfg.stride]),
        "ratios": np.array([cfg.test_ratio, cfg.val_ratio]), "seed": np.array(cfg.random_seed),
        "train_idx": d["train_idx"], "val_idx": d["val_idx"], "test_idx": d["test_idx"],
    }
    for split in ("train", "val", "test"):
        X, Y = d[f"X_{split}"], d[f"Y_{split}"]
        sel = pick(X.shape[0])
        out[f"{split}_sel"] = sel
        out[f"X_{split}_sel"] = X[sel]
        out[f"Y_{split}_sel"] = Y[sel]
        out[f"X_{split}_sum"] = np.array(X.astype(np.float64).sum())
        out[f"Y_{split}_sum"] = np.array(Y.astype(np.float64).sum())
        out[f"X_{split}_shape"] = np.array(X.shape)
    np.savez_compressed(os.path.join(HERE, "input_ref15x24.npz"), **out)

    # ---------------------------------------------------------------- other geometries
    cases = {}
    for name, (n, seed, hi, wi, ho, wo, ov, st, K) in {
        "stride7_k12": (2600, 11, 12, 12, 12, 12, 0, 7, 12),
        "maxoverlap_k1": (400, 12, 4, 6, 3, 5, 15, 3, 1),
        "wide_k5": (3000, 13, 23, 23, 23, 23, 46, 5, 5),
    }.items():
        s = synthetic_series(n, seed)
        sc = MinMaxScaler()
        norm = sc.fit_transform(s.reshape(-1, 1)).flatten()
        X, Y = build_image_pairs(norm, "gray2d", hi, wi, ho, wo, ov, st)
        Xs, Ys, idx = ref_dataset._build_sequences(X[:, None], Y, K)
        sel = pick(Xs.shape[0])
        cases[f"{name}.series"] = s
        cases[f"{name}.geometry"] = np.array([hi, wi, ho, wo, ov, st, K])
        cases[f"{name}.scale"] = sc.scale_
        cases[f"{name}.offset"] = sc.min_
        cases[f"{name}.n_windows"] = np.array(X.shape[0])
        cases[f"{name}.orig_idx"] = idx
        cases[f"{name}.sel"] = sel
        cases[f"{name}.Xseq_sel"] = Xs[sel]          # (n, K, 1, H, W)
        cases[f"{name}.Yseq_sel"] = Ys[sel]
        cases[f"{name}.X_sum"] = np.array(X.astype(np.float64).sum())
        cases[f"{name}.Y_sum"] = np.array(Y.astype(np.float64).sum())
    np.savez_compressed(os.path.join(HERE, "input_cases.npz"), **cases)

    # ---------------------------------------------------------------- the harness on a small network
    import train as ref_train                                  # noqa: E402
    import test as ref_test                                    # noqa: E402
    cfg.unet_features = (4, 8, 16)          # 15x24 halves three times at most
    cfg.in_channels, cfg.out_channels = 1, 1
    cfg.batch_size, cfg.num_epochs, cfg.lr = 64, 4, 1e-3
    cfg.device = "cpu"
    cfg.early_stop, cfg.early_stop_patience, cfg.early_stop_min_delta = True, 2, 0.0
    torch.manual_seed(5)
    torch.set_num_threads(4)
    import io
    import contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ref_train.main()
    log = buf.getvalue()
    losses = []
    for line in log.splitlines():
        if line.startswith("Epoch ["):
            parts = line.replace("|", " ").split()
            losses.append((float(parts[parts.index("Loss:") + 1]), float(parts[-1])))
    ckpt = torch.load(cfg.model_ckpt, map_location="cpu", weights_only=True)
    # the shuffled order the reference's DataLoader drew: replay the same seed over an index dataset
    torch.manual_seed(5)
    model = ref_train.build_model_from_cfg(cfg)                # consumes the init draws, as main() did
    init_state = {k: v.clone() for k, v in model.state_dict().items()}
    n_train = len(d["train_idx"])
    order = []
    ds = torch.utils.data.TensorDataset(torch.arange(n_train))
    loader = torch.utils.data.DataLoader(ds, batch_size=cfg.batch_size, shuffle=True, drop_last=False)
    val_like = torch.utils.data.DataLoader(ds, batch_size=cfg.batch_size, shuffle=False)
    for _ in range(len(losses)):
        order.append(torch.cat([b[0] for b in loader]).numpy())
        iter(val_like)       # evaluate() opens the validation loader: one more draw from the global generator
    model.load_state_dict(ckpt["model_state_dict"])
    _, _, test_loader, _, _, _, _ = ref_dataset.prepare_dataloaders()
    mae, mse = ref_test.compute_test_metrics(model, test_loader, "cpu")
    h = {
        "printed_losses": np.array(losses), "epochs_run": np.array(len(losses)),
        "best_val_loss": np.array(ckpt["best_val_loss"]), "best_epoch": np.array(ckpt["epoch"]),
        "ckpt_keys": np.array(sorted(ckpt.keys())), "epoch_order": np.stack(order),
        "test_mae": np.array(mae), "test_mse": np.array(mse),
        "features": np.array(cfg.unet_features), "batch_size": np.array(cfg.batch_size), "lr": np.array(cfg.lr),
        "seed": np.array(5), "patience": np.array(2),
    }
    for k, v in init_state.items():
        h["init." + k] = v.numpy()
    for k, v in ckpt["model_state_dict"].items():
        h["best." + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "harness_small.npz"), **h)
    print("wrote fixtures; epochs:", losses, "best", ckpt["epoch"], "mae/mse", mae, mse)


if __name__ == "__main__":
    main()
