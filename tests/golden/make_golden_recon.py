#!/usr/bin/env python3
"""Generate the series-reconstruction fixture FROM THE REFERENCE ITSELF (build container only):

    MPLBACKEND=Agg python /root/repo/tests/golden/make_golden_recon.py

Calls the reference's ``visualize_results.visualize_test_windows_sec_original`` (the function that
averages overlapping predicted horizons back into one series, visualize_results.py:443-476) with a
stand-in model, synthetic windows and a recording object in place of the scaler: the function hands
the reconstructed (compressed) prediction and ground-truth series to ``scaler.inverse_transform``,
which is where they are captured.  Only data is stored (recon_cases.npz)."""
import os
import sys
import tempfile

import numpy as np

os.environ.setdefault("MPLBACKEND", "Agg")
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


class Recorder:
    def __init__(self):
        self.seen = []

    def inverse_transform(self, a):
        self.seen.append(np.array(a, dtype=np.float64).reshape(-1))
        return a


def main():
    os.chdir(tempfile.mkdtemp(prefix="golden_recon_"))
    sys.path.insert(0, REF)
    import torch
    from config import cfg
    import visualize_results as V

    class Affine(torch.nn.Module):
        def forward(self, x):
            return x * 0.5 + 0.1

    out = {}
    for name, (H, W, overlap_rows, stride, n_windows, n_pick, seed) in {
        "ref15x24": (15, 24, 3, 1, 400, 60, 1), "stride5": (12, 12, 0, 5, 300, 90, 2), "dense": (6, 8, 2, 2, 120, 120, 3),
    }.items():
        cfg.H_in = cfg.H_out = H
        cfg.W_in = cfg.W_out = W
        cfg.overlap_rows, cfg.stride = overlap_rows, stride
        cfg.visualized_dir = os.path.join(os.getcwd(), "viz_" + name)
        rng = np.random.RandomState(seed)
        idx = rng.permutation(n_windows)[:n_pick].astype(np.int64)
        g = torch.Generator().manual_seed(seed)
        X = torch.rand(n_pick, 1, H, W, generator=g)
        Y = torch.rand(n_pick, 1, H, W, generator=g)
        loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=32, shuffle=False)
        rec = Recorder()
        V.visualize_test_windows_sec_original(Affine(), loader, idx, "cpu", rec, seq_len=1, num_windows=1)
        assert len(rec.seen) == 2
        out[name + ".geometry"] = np.array([H, W, H, W, stride])
        out[name + ".idx"] = idx
        out[name + ".pred"] = Affine()(X).numpy()
        out[name + ".gt"] = Y.numpy()
        out[name + ".pred_series"] = rec.seen[0]
        out[name + ".gt_series"] = rec.seen[1]
        print(name, "L =", rec.seen[0].shape[0])
    np.savez_compressed(os.path.join(HERE, "recon_cases.npz"), **out)


if __name__ == "__main__":
    main()
