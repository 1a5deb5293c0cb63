"""On-GPU input pipeline and train/test harness against the fixtures recorded from the reference
(tests/golden/make_golden_input.py) and against the CPU oracle.

Bars: the pipeline's tensors are float32 images of float64-scaled samples and must be
BIT-IDENTICAL to the reference's (torch.equal / array_equal).  The harness run is a float
computation: fp32 kernels, epoch losses within 5e-5 relative of the values the reference printed
(6 decimals, so an absolute floor of 1e-6 is added), test metrics within 1e-4."""
import types
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import input_oracle as IO

pytestmark = pytest.mark.gpu

CASES = ["stride7_k12", "maxoverlap_k1", "wide_k5"]


@pytest.fixture(scope="module")
def ua():
    import unet_amd
    unet_amd.lib()
    return unet_amd


def _series(ua, g):
    hi, wi, ho, wo, ov, st = (int(v) for v in g["geometry"])
    return ua.SeriesWindows(g["series"], hi, wi, ho, wo, ov, st)


def test_reference_dataset_bit_identical(ua, golden):
    g = golden("input_ref15x24")
    sw = _series(ua, g)
    assert sw.scale == float(g["scale"][0]) and sw.offset == float(g["offset"][0])     # MinMaxScaler.fit on the device
    tr, va, te = sw.split(float(g["ratios"][0]), float(g["ratios"][1]), int(g["seed"]))
    for split, ids in (("train", tr), ("val", va), ("test", te)):
        assert np.array_equal(ids, g[f"{split}_idx"]), split
        sel = g[f"{split}_sel"]
        X, Y = sw.batch(ids[sel])
        assert X.dtype == torch.float32 and tuple(X.shape) == (len(sel), 1, sw.H_in, sw.W_in)
        assert torch.equal(X.cpu(), torch.from_numpy(g[f"X_{split}_sel"])), split
        assert torch.equal(Y.cpu()[:, 0], torch.from_numpy(g[f"Y_{split}_sel"])), split
        # the whole split in one batch: float64 checksums of the reference's arrays
        X, Y = sw.batch(ids)
        assert np.array_equal(X.cpu().numpy().astype(np.float64).sum(), g[f"X_{split}_sum"])
        assert np.array_equal(Y.cpu().numpy().astype(np.float64).sum(), g[f"Y_{split}_sum"])


@pytest.mark.parametrize("name", CASES)
def test_context_windows_as_channels_bit_identical(ua, golden, name):
    c = golden("input_cases")
    hi, wi, ho, wo, ov, st, K = (int(v) for v in c[name + ".geometry"])
    sw = ua.SeriesWindows(c[name + ".series"], hi, wi, ho, wo, ov, st, context=K)
    assert sw.num_windows == int(c[name + ".n_windows"])
    assert sw.scale == float(c[name + ".scale"][0]) and sw.offset == float(c[name + ".offset"][0])
    assert np.array_equal(sw.sample_ids, c[name + ".orig_idx"])
    sel = c[name + ".sel"]
    X, Y = sw.batch(sw.sample_ids[sel])
    assert torch.equal(X.cpu(), torch.from_numpy(c[name + ".Xseq_sel"][:, :, 0])), name     # (n, K, 1, H, W) -> channels
    assert torch.equal(Y.cpu()[:, 0], torch.from_numpy(c[name + ".Yseq_sel"])), name


@pytest.mark.parametrize("geom", [(32, 32, 32, 32, 64, 3, 12, 512), (23, 23, 23, 23, 0, 1, 1, 100), (5, 7, 3, 3, 9, 2, 4, 33)])
def test_random_batches_equal_oracle(ua, geom):
    """Sizes up to the benched one (batch 512, 12 context windows of 32x32), ids in random order with repeats."""
    hi, wi, ho, wo, ov, st, K, B = geom
    rng = np.random.RandomState(3)
    n = hi * wi + ho * wo - ov + st * 700
    s = np.abs(np.cumsum(rng.randn(n)) * 17.0 + 900.0) + rng.rand(n)
    s[5] = s.max() * 3.0            # an outlier fixes the range
    sw = ua.SeriesWindows(s, hi, wi, ho, wo, ov, st, context=K)
    ids = rng.randint(K - 1, sw.num_windows, size=B)
    ids[:2] = [K - 1, sw.num_windows - 1]                      # first and last valid sample
    X, Y = sw.batch(ids)
    Xo, Yo = IO.batch_tensors(s, ids, K, hi, wi, ho, wo, ov, st)
    assert torch.equal(X.cpu(), torch.from_numpy(Xo)) and torch.equal(Y.cpu(), torch.from_numpy(Yo))
    # every value of the normalised series lies in [0, 1] and the extremes are hit exactly
    full, _ = sw.batch(sw.sample_ids[:: max(1, hi * wi // st)])
    assert full.min().item() >= 0.0 and full.max().item() <= 1.0
    np.testing.assert_allclose(sw.inverse_transform(X[:1]).cpu().numpy(), (Xo[:1].astype(np.float64) - sw.offset) / sw.scale,
                               rtol=1e-12)


def test_errors_and_edge_cases(ua):
    s = np.arange(2000, dtype=np.float64)
    with pytest.raises(ValueError, match="Time series too short"):
        ua.SeriesWindows(s[:700], 15, 24, 15, 24, 0, 1)
    with pytest.raises(ValueError, match=r"overlap_len must be in \[0, 360\], got 400"):
        ua.SeriesWindows(s, 15, 24, 15, 24, 400, 1)
    with pytest.raises(ValueError, match="stride must be positive"):
        ua.SeriesWindows(s, 15, 24, 15, 24, 0, 0)
    with pytest.raises(ValueError, match=r"Not enough windows \(3\) to build sequences with K=5"):
        ua.SeriesWindows(s[:722], 15, 24, 15, 24, 0, 1, context=5)
    with pytest.raises(RuntimeError):
        ua.SeriesWindows(s, 15, 24, 15, 24, 0, 1, device="cpu")
    sw = ua.SeriesWindows(s, 15, 24, 15, 24, 72, 1, context=3)
    for bad in ([1], [sw.num_windows], [-1]):
        with pytest.raises(IndexError):
            sw.batch(bad)
    with pytest.raises(ValueError):
        sw.batch([])
    # a constant series: scikit-learn keeps scale 1, so every value maps to 0
    c = ua.SeriesWindows(np.full(900, 7.25), 15, 24, 15, 24, 72, 1)
    assert c.scale == 1.0 and c.offset == -7.25
    X, Y = c.batch([0, c.num_windows - 1])
    assert X.abs().max().item() == 0.0 and Y.abs().max().item() == 0.0
    # NaNs are skipped by the fit, as numpy.nanmin / nanmax do
    t = s.copy()
    t[10] = np.nan
    assert ua.SeriesWindows(t, 15, 24, 15, 24, 72, 1).scale == 1.0 / 1999.0
    # one epoch, unshuffled and with an explicit order; the last batch is short (drop_last=False)
    ids = sw.sample_ids[:70]
    got = [x for x, _ in sw.batches(ids, 32)]
    assert [b.shape[0] for b in got] == [32, 32, 6] and sw.num_batches(ids, 32) == 3
    order = np.random.RandomState(0).permutation(70)
    a = torch.cat([x for x, _ in sw.batches(ids, 32, order=order)])
    assert torch.equal(a, sw.batch(ids[order])[0])


def _cfg(h, g, ckpt):
    return types.SimpleNamespace(model_type="unet", in_channels=1, out_channels=1,
                                 unet_features=tuple(int(v) for v in h["features"]), H_out=int(g["geometry"][2]),
                                 W_out=int(g["geometry"][3]), compute_dtype="fp32", lr=float(h["lr"]),
                                 batch_size=int(h["batch_size"]), num_epochs=int(h["epochs_run"]), model_ckpt=ckpt,
                                 early_stop=True, early_stop_patience=int(h["patience"]), early_stop_min_delta=0.0)


def test_harness_reproduces_the_reference_run(ua, golden, tmp_path):
    """The reference's train.main() / compute_test_metrics on a 3-level U-Net over 15x24 windows (both
    bilinear resizes fire), replayed through the device pipeline and the HIP network in fp32."""
    g, h = golden("input_ref15x24"), golden("harness_small")
    sw = _series(ua, g)
    ckpt = str(tmp_path / "best.pt")
    cfg = _cfg(h, g, ckpt)
    model = ua.build_model_from_cfg(cfg)
    model.load_state_dict(OrderedDict((k[5:], torch.from_numpy(h[k])) for k in h.files if k.startswith("init.")))
    model = model.to("cuda")
    tr, va, te = (sw.to_device_ids(g[k]) for k in ("train_idx", "val_idx", "test_idx"))
    opt = ua.FusedAdam(model.parameters(), lr=cfg.lr, model=model)
    logs = []
    res = ua.harness.fit(model, opt, ua.MSELoss(),
                         lambda e: sw.batches(tr, cfg.batch_size, order=h["epoch_order"][e - 1]),
                         lambda e: sw.batches(va, cfg.batch_size), cfg.num_epochs, ckpt, True, cfg.early_stop_patience, 0.0,
                         logs.append)
    printed = h["printed_losses"]
    assert len(res["history"]) == printed.shape[0] == res["stopped_at"]
    for (tl, vl), (ptl, pvl) in zip(res["history"], printed):
        assert abs(tl - ptl) <= 5e-5 * ptl + 1e-6 and abs(vl - pvl) <= 5e-5 * pvl + 1e-6, (res["history"], printed)
    assert res["best_epoch"] == int(h["best_epoch"])
    ck = torch.load(ckpt, map_location="cpu", weights_only=True)                      # test.py:134-135
    assert sorted(ck.keys()) == list(h["ckpt_keys"]) and ck["epoch"] == int(h["best_epoch"])
    assert ck["best_val_loss"] == pytest.approx(float(h["best_val_loss"]), rel=5e-5)
    assert list(ck["model_state_dict"].keys()) == [k[5:] for k in h.files if k.startswith("best.")]
    for k, v in ck["model_state_dict"].items():
        ref = torch.from_numpy(h["best." + k])
        assert ((v - ref).norm() / (ref.norm() + 1e-12)).item() < 2e-3, k
    mae, mse = ua.harness.compute_test_metrics(model, sw.batches(te, cfg.batch_size))
    assert mae == pytest.approx(float(h["test_mae"]), rel=1e-4) and mse == pytest.approx(float(h["test_mse"]), rel=1e-4)
    assert logs[0].startswith("Epoch [1/4] Train Loss: 0.1726") and "Saved best model" in logs[1]


def test_fit_from_cfg_with_reference_shuffling(ua, golden, tmp_path):
    """Same seed, same draws: the shuffled epoch orders equal the ones the reference's DataLoader drew."""
    g, h = golden("input_ref15x24"), golden("harness_small")
    sw = _series(ua, g)
    cfg = _cfg(h, g, str(tmp_path / "b.pt"))
    cfg.num_epochs = 2
    torch.manual_seed(int(h["seed"]))
    res = ua.harness.fit_from_cfg(cfg, sw, log=None)                 # builds the model first, as train.py:495 does
    for (tl, vl), (ptl, pvl) in zip(res["history"], h["printed_losses"]):
        assert abs(tl - ptl) <= 5e-5 * ptl + 1e-6 and abs(vl - pvl) <= 5e-5 * pvl + 1e-6


@pytest.mark.parametrize("name", ["ref15x24", "stride5", "dense"])
def test_series_reconstruction_matches_reference(ua, golden, name):
    """Overlap-averaged series on the device against what the reference's plotting function computed.
    Float64 averages of float32 terms; the device sums in window order, the reference in batch order:
    relative 1e-12."""
    c = golden("recon_cases")
    H, W, Ho, Wo, st = (int(v) for v in c[name + ".geometry"])
    pred, gt = torch.from_numpy(c[name + ".pred"]).cuda(), torch.from_numpy(c[name + ".gt"]).cuda()
    ps, gs, mask = ua.reconstruct_series(pred, gt, c[name + ".idx"], H * W, Ho * Wo, st)
    assert ps.dtype == torch.float64 and ps.shape[0] == c[name + ".pred_series"].shape[0] == int(mask.sum().item())
    np.testing.assert_allclose(ps.cpu().numpy(), c[name + ".pred_series"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(gs.cpu().numpy(), c[name + ".gt_series"], rtol=1e-12, atol=0)
    _, _, omask = IO.reconstruct_series(c[name + ".pred"], c[name + ".gt"], c[name + ".idx"], H * W, Ho * Wo, st)
    assert np.array_equal(mask.cpu().numpy(), omask)


def test_series_reconstruction_large_and_errors(ua):
    rng = np.random.RandomState(5)
    H = W = 32
    for stride, n_win, n_pick in ((1, 5000, 2048), (7, 3000, 700), (1500, 40, 40)):     # dense overlap ... disjoint horizons
        idx = rng.permutation(n_win)[:n_pick]
        pred = torch.rand(n_pick, 1, H, W)
        gt = torch.rand(n_pick, 1, H, W)
        ps, gs, mask = ua.reconstruct_series(pred.cuda(), gt.cuda(), idx, H * W, H * W, stride)
        po, go, mo = IO.reconstruct_series(pred.numpy(), gt.numpy(), idx, H * W, H * W, stride)
        assert np.array_equal(mask.cpu().numpy(), mo)
        np.testing.assert_allclose(ps.cpu().numpy(), po, rtol=1e-12)
        np.testing.assert_allclose(gs.cpu().numpy(), go, rtol=1e-12)
    p = torch.rand(4, 1, 4, 4).cuda()
    with pytest.raises(ValueError):
        ua.reconstruct_series(p, p, [0, 1, 1, 2], 16, 16, 1)           # duplicate window
    with pytest.raises(ValueError):
        ua.reconstruct_series(p, p, [0, 1, 2], 16, 16, 1)              # one index per row
    with pytest.raises(ValueError):
        ua.reconstruct_series(p, p, [0, 1, 2, 3], 16, 15, 1)           # row length
    with pytest.raises(RuntimeError):
        ua.reconstruct_series(p.cpu(), p.cpu(), [0, 1, 2, 3], 16, 16, 1)
