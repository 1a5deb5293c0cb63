#!/usr/bin/env python3
"""The reference's ``python test.py`` for the U-Net route: loads the best checkpoint
(test.py:133-135) and prints MAE / MSE on the test split in normalised scale (test.py:187-190).
Takes the same arguments as scripts/train.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    import torch
    import unet_amd
    import train as train_script
    args = train_script.parse(argv)
    series, cfg = train_script.make(args)
    _, _, te = series.split(cfg.test_ratio, cfg.val_ratio, cfg.random_seed)
    model = unet_amd.build_model_from_cfg(cfg).to(series.device)
    print(f"Loading best model from {cfg.model_ckpt} ...")
    ckpt = torch.load(cfg.model_ckpt, map_location=series.device, weights_only=True)
    model.load_state_dict(ckpt["model_state_dict"])
    mae, mse = unet_amd.harness.compute_test_metrics(model, series.batches(te, cfg.batch_size))
    print(f"\nTest MAE (normalized): {mae:.6f}")
    print(f"Test MSE (normalized): {mse:.6f}")


if __name__ == "__main__":
    main()
