#!/usr/bin/env python3
"""The reference's ``python train.py`` for the U-Net route, on the device pipeline and the HIP network.

    python scripts/train.py --csv data/raw/Pangyo_traffic_volume.csv --value-col "All packets" \
        --h-in 15 --w-in 24 --h-out 15 --w-out 24 --overlap-rows 3 --stride 1 --features 32,64,128 \
        --batch-size 128 --epochs 200 --lr 1e-4 --ckpt checkpoints/unet_best.pt

Defaults follow the reference's config.py.  The checkpoint has the reference's format
(train.py:522-529) and loads into either implementation; scripts/test.py reports the test metrics."""
import argparse
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--csv", required=True)
    ap.add_argument("--value-col", default="All packets")
    ap.add_argument("--h-in", type=int, default=15)
    ap.add_argument("--w-in", type=int, default=24)
    ap.add_argument("--h-out", type=int, default=15)
    ap.add_argument("--w-out", type=int, default=24)
    ap.add_argument("--overlap-rows", type=int, default=3)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--context", type=int, default=1, help="consecutive windows stacked as input channels")
    ap.add_argument("--features", default="32,64,128,256")
    ap.add_argument("--batch-size", type=int, default=128)
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--patience", type=int, default=15)
    ap.add_argument("--min-delta", type=float, default=0.0)
    ap.add_argument("--no-early-stop", action="store_true")
    ap.add_argument("--test-ratio", type=float, default=0.2)
    ap.add_argument("--val-ratio", type=float, default=0.1)
    ap.add_argument("--split-seed", type=int, default=42)
    ap.add_argument("--seed", type=int, default=None, help="torch.manual_seed before the model is built")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--ckpt", default="checkpoints/unet_best.pt")
    ap.add_argument("--device", default="cuda:0")
    return ap.parse_args(argv)


def make(args):
    import pandas as pd
    import unet_amd
    df = pd.read_csv(args.csv)
    if args.value_col not in df.columns:
        raise KeyError(f"Column '{args.value_col}' not found in CSV. Available: {list(df.columns)}")   # dataset.py:317-321
    series = unet_amd.SeriesWindows(df[args.value_col].values, args.h_in, args.w_in, args.h_out, args.w_out,
                                    args.overlap_rows * args.w_in, args.stride, context=args.context, device=args.device)
    cfg = types.SimpleNamespace(model_type="unet", in_channels=args.context, out_channels=1,
                                unet_features=tuple(int(v) for v in args.features.split(",")), H_out=args.h_out,
                                W_out=args.w_out, compute_dtype=args.dtype, lr=args.lr, batch_size=args.batch_size,
                                num_epochs=args.epochs, model_ckpt=args.ckpt, early_stop=not args.no_early_stop,
                                early_stop_patience=args.patience, early_stop_min_delta=args.min_delta,
                                test_ratio=args.test_ratio, val_ratio=args.val_ratio, random_seed=args.split_seed)
    return series, cfg


def main(argv=None):
    import torch
    import unet_amd
    args = parse(argv)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    series, cfg = make(args)
    tr, va, te = series.split(cfg.test_ratio, cfg.val_ratio, cfg.random_seed)
    print(f"Train windows: {len(tr)}, Val windows: {len(va)}")
    res = unet_amd.harness.fit_from_cfg(cfg, series, splits=(tr, va, te))
    print(f"best val loss {res['best_val_loss']:.6f} at epoch {res['best_epoch']}, stopped at epoch {res['stopped_at']}")


if __name__ == "__main__":
    main()
