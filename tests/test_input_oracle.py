"""The input-pipeline / harness oracle must reproduce what the reference's own functions produced
(fixtures from tests/golden/make_golden_input.py), and the host-side pieces of the product
(window counting, error texts, shuffling order, epoch bookkeeping) must agree with it.  No GPU."""
import numpy as np
import pytest
import torch

from oracle import input_oracle as IO
from oracle import unet_oracle as UO

CASES = ["stride7_k12", "maxoverlap_k1", "wide_k5"]


def test_minmax_matches_sklearn_fixture(golden):
    g = golden("input_ref15x24")
    scale, offset = IO.minmax_params(g["series"])
    assert scale == float(g["scale"][0]) and offset == float(g["offset"][0])
    c = golden("input_cases")
    for name in CASES:
        scale, offset = IO.minmax_params(c[name + ".series"])
        assert scale == float(c[name + ".scale"][0]) and offset == float(c[name + ".offset"][0]), name
    assert IO.minmax_params(np.full(10, 3.5)) == (1.0, -3.5)     # constant series: scale stays 1


def test_reference_dataset_reproduced(golden):
    """generate_and_save_dataset of the reference: scaling, windows, seed-42 split."""
    g = golden("input_ref15x24")
    hi, wi, ho, wo, ov, st = (int(v) for v in g["geometry"])
    scale, offset = IO.minmax_params(g["series"])
    X, Y = IO.image_pairs(IO.minmax_apply(g["series"], scale, offset), hi, wi, ho, wo, ov, st)
    tr, va, te = IO.split_ids(X.shape[0], float(g["ratios"][0]), float(g["ratios"][1]), int(g["seed"]))
    for split, ids in (("train", tr), ("val", va), ("test", te)):
        assert np.array_equal(ids, g[f"{split}_idx"]), split
        sel = g[f"{split}_sel"]
        assert np.array_equal(X[ids][sel][:, None], g[f"X_{split}_sel"]), split
        assert np.array_equal(Y[ids][sel], g[f"Y_{split}_sel"]), split
        assert X[ids].astype(np.float64).sum() == float(g[f"X_{split}_sum"])
        assert Y[ids].astype(np.float64).sum() == float(g[f"Y_{split}_sum"])
        assert tuple(g[f"X_{split}_shape"]) == (len(ids), 1, hi, wi)


@pytest.mark.parametrize("name", CASES)
def test_windows_and_sequences_reproduced(golden, name):
    c = golden("input_cases")
    hi, wi, ho, wo, ov, st, K = (int(v) for v in c[name + ".geometry"])
    s = c[name + ".series"]
    scale, offset = IO.minmax_params(s)
    X, Y = IO.image_pairs(IO.minmax_apply(s, scale, offset), hi, wi, ho, wo, ov, st)
    assert X.shape[0] == int(c[name + ".n_windows"]) == IO.num_windows(len(s), hi, wi, ho, wo, ov, st)
    assert X.astype(np.float64).sum() == float(c[name + ".X_sum"]) and Y.astype(np.float64).sum() == float(c[name + ".Y_sum"])
    Xs, Ys, idx = IO.sequences(X[:, None], Y, K)
    assert np.array_equal(idx, c[name + ".orig_idx"])
    sel = c[name + ".sel"]
    assert np.array_equal(Xs[sel], c[name + ".Xseq_sel"]) and np.array_equal(Ys[sel], c[name + ".Yseq_sel"])
    # the batch form the product produces: context windows as channels, target with a channel axis
    Xb, Yb = IO.batch_tensors(s, idx[sel], K, hi, wi, ho, wo, ov, st)
    assert np.array_equal(Xb, c[name + ".Xseq_sel"][:, :, 0]) and np.array_equal(Yb[:, 0], c[name + ".Yseq_sel"])


def test_window_count_and_errors_match_library():
    """mi_window_count is host code: same counts and the reference's ValueError texts."""
    import unet_amd
    L = unet_amd.lib()
    for args in [(1400, 15, 24, 15, 24, 72, 1), (2600, 12, 12, 12, 12, 0, 7), (400, 4, 6, 3, 5, 15, 3), (721, 15, 24, 15, 24, 0, 1)]:
        assert L.mi_window_count(*args) == IO.num_windows(*args)
    for args in [(100, 15, 24, 15, 24, 72, 1), (720, 15, 24, 15, 24, 0, 1), (1400, 15, 24, 15, 24, 361, 1),
                 (1400, 15, 24, 15, 24, -1, 1), (1400, 15, 24, 15, 24, 72, 0)]:
        with pytest.raises(ValueError) as e:
            IO.num_windows(*args)
        assert L.mi_window_count(*args) == -1
        assert L.mi_last_error().decode() == str(e.value)


def test_reference_shuffle_equals_dataloader_order():
    """Same draws from torch's global generator as DataLoader(shuffle=True) + a plain loader (dataset.py:561-562)."""
    import unet_amd
    n, bs = 97, 16
    ds = torch.utils.data.TensorDataset(torch.arange(n))
    torch.manual_seed(11)
    ref = []
    for _ in range(3):
        ref.append(torch.cat([b[0] for b in torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=True)]))
        iter(torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=False))
    torch.manual_seed(11)
    for e in range(3):
        assert torch.equal(unet_amd.reference_shuffle(n), ref[e])
        unet_amd.reference_open_loader()


def _fixture_batches(g, h):
    hi, wi, ho, wo, ov, st = (int(v) for v in g["geometry"])
    scale, offset = IO.minmax_params(g["series"])
    X, Y = IO.image_pairs(IO.minmax_apply(g["series"], scale, offset), hi, wi, ho, wo, ov, st)
    X, Y = torch.from_numpy(X[:, None]), torch.from_numpy(Y[:, None])
    bs = int(h["batch_size"])

    def chunks(ids):
        return [(X[ids[i:i + bs]], Y[ids[i:i + bs]]) for i in range(0, len(ids), bs)]
    tr = g["train_idx"]
    return (lambda epoch: chunks(tr[h["epoch_order"][epoch - 1]])), chunks(g["val_idx"]), chunks(g["test_idx"]), (ho, wo)


def test_harness_oracle_reproduces_reference_run(golden):
    """train.main() + test.compute_test_metrics of the reference on the small network."""
    g, h = golden("input_ref15x24"), golden("harness_small")
    train_b, val_b, test_b, osz = _fixture_batches(g, h)
    state = {k[5:]: torch.from_numpy(h[k]) for k in h.files if k.startswith("init.")}
    order = UO.state_dict_order(1, 1, tuple(int(v) for v in h["features"]))
    assert list(state) == [p + sfx for p in order for sfx in (".weight", ".bias")]
    net = IO.OracleNet(state, osz)
    opt = torch.optim.Adam(net.parameters(), lr=float(h["lr"]))
    res = IO.fit(net, opt, torch.nn.MSELoss(), train_b, val_b, int(h["epochs_run"]), True, int(h["patience"]), 0.0)
    printed = h["printed_losses"]
    assert len(res["history"]) == printed.shape[0]
    for (tl, vl), (ptl, pvl) in zip(res["history"], printed):
        assert abs(tl - ptl) < 1.5e-6 and abs(vl - pvl) < 1.5e-6            # the reference prints 6 decimals
    assert res["best"]["epoch"] == int(h["best_epoch"])
    assert sorted(res["best"].keys()) == list(h["ckpt_keys"])
    assert res["best"]["best_val_loss"] == pytest.approx(float(h["best_val_loss"]), rel=2e-6)
    for k, v in res["best"]["model_state_dict"].items():
        np.testing.assert_allclose(v.numpy(), h["best." + k], rtol=2e-4, atol=2e-6)
    mae, mse = IO.test_metrics(net, test_b)
    assert mae == pytest.approx(float(h["test_mae"]), rel=5e-6) and mse == pytest.approx(float(h["test_mse"]), rel=5e-6)


class _Scripted:
    """A stand-in 'model' whose validation loss follows a script: exercises the bookkeeping only."""

    def __init__(self, val_losses):
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.vals, self.epoch, self.training = val_losses, 0, True

    def parameters(self):
        return [self.p]

    def state_dict(self):
        return {"p": self.p.detach().clone()}

    def train(self):
        self.training = True
        self.epoch += 1

    def eval(self):
        self.training = False

    def __call__(self, x):
        return self.p * 0 + (1.0 if self.training else self.vals[self.epoch - 1]) + x * 0


def test_fit_bookkeeping_early_stop_and_checkpoint(tmp_path):
    """train.py:504-540: improvement test with min_delta, patience counter, checkpoint dict."""
    import unet_amd
    vals = [0.9, 0.5, 0.6, 0.55, 0.7, 0.1, 0.05]
    m = _Scripted(vals)
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    crit = lambda pred, y: (pred - y).mean()          # noqa: E731  loss == the scripted value
    x, y = torch.zeros(1), torch.zeros(1)
    path = str(tmp_path / "ck" / "best.pt")
    logs = []
    res = unet_amd.harness.fit(m, opt, crit, lambda e: [(x, y)] * 3, lambda e: [(x, y)] * 2, 7, path, True, 3, 0.0, logs.append)
    assert res["stopped_at"] == 5 and res["best_epoch"] == 2 and res["best_val_loss"] == pytest.approx(0.5)
    assert [round(v, 6) for _, v in res["history"]] == vals[:5] and all(t == pytest.approx(1.0) for t, _ in res["history"])
    ck = torch.load(path, weights_only=True)
    assert sorted(ck.keys()) == ["best_val_loss", "epoch", "model_state_dict"] and ck["epoch"] == 2
    assert logs[0] == "Epoch [1/7] Train Loss: 1.000000 | Val Loss: 0.900000" and logs[-1].startswith("Early stopping triggered at epoch 5")
    # the same script against the oracle loop
    m2 = _Scripted(vals)
    ref = IO.fit(m2, torch.optim.SGD(m2.parameters(), lr=0.0), crit, lambda e: [(x, y)] * 3, [(x, y)] * 2, 7, True, 3, 0.0)
    assert ref["stopped_at"] == 5 and ref["best"]["epoch"] == 2 and ref["history"] == res["history"]
    # min_delta: an improvement smaller than it does not count; without early_stop the loop runs out
    m3 = _Scripted(vals)
    res3 = unet_amd.harness.fit(m3, torch.optim.SGD(m3.parameters(), lr=0.0), crit, lambda e: [(x, y)], lambda e: [(x, y)], 7,
                                None, False, 3, 0.42, None)
    assert res3["stopped_at"] == 7 and res3["best_epoch"] == 6 and res3["best_val_loss"] == pytest.approx(0.1)


@pytest.mark.parametrize("name", ["ref15x24", "stride5", "dense"])
def test_series_reconstruction_oracle_reproduces_reference(golden, name):
    """visualize_results.visualize_test_windows_sec_original: the two series it hands to the scaler."""
    c = golden("recon_cases")
    H, W, Ho, Wo, st = (int(v) for v in c[name + ".geometry"])
    p, g, mask = IO.reconstruct_series(c[name + ".pred"], c[name + ".gt"], c[name + ".idx"], H * W, Ho * Wo, st)
    assert np.array_equal(p, c[name + ".pred_series"]) and np.array_equal(g, c[name + ".gt_series"])
    assert mask.sum() == p.shape[0]
