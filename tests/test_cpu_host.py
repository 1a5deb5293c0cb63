"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol the
header declares, plan metadata (parameter layout, backward segments) matches the reference's
naming, the Python boundary mirrors the reference factory, and the data-parallel gradient
averaging reproduces the global-batch gradient (two gloo processes, oracle gradients)."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import unet_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "unet_tm_hip.h")


@pytest.fixture(scope="module")
def ua():
    import unet_amd
    return unet_amd


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(ua):
    from unet_amd import _lib
    names = declared_symbols()
    assert len(names) >= 40
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} is declared in the header but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names          # and nothing is bound that the header does not declare
    assert raw.mi_version() >= 1


def make_plan(L, cin, cout, feats, B, H, W, dtype=0, out=(0, 0), need_dx=0):
    from unet_amd._lib import UNetDesc
    d = UNetDesc()
    d.in_channels, d.out_channels, d.n_levels = cin, cout, len(feats)
    for i, f in enumerate(feats):
        d.features[i] = f
    d.batch, d.height, d.width, d.out_h, d.out_w, d.dtype, d.need_input_grad = B, H, W, out[0], out[1], dtype, need_dx
    h = C.c_void_p()
    rc = L.mi_unet_plan_create(C.byref(d), C.byref(h))
    return rc, h


def plan_params(L, h):
    out = []
    for i in range(L.mi_unet_plan_num_params(h)):
        name = C.create_string_buffer(96)
        off, shape, nd = C.c_int64(), (C.c_int64 * 4)(), C.c_int()
        assert L.mi_unet_plan_param_info(h, i, name, 96, C.byref(off), shape, C.byref(nd)) == 0
        out.append((name.value.decode(), off.value, tuple(shape[:nd.value])))
    return out


def plan_segments(L, h):
    segs = []
    for s in range(L.mi_unet_plan_num_segments(h)):
        lo, hi = C.c_int64(), C.c_int64()
        assert L.mi_unet_plan_segment_range(h, s, C.byref(lo), C.byref(hi)) == 0
        segs.append((lo.value, hi.value))
    return segs


@pytest.mark.parametrize("cfg", [(12, 1, (32, 64, 128, 256), 4, 32, 32), (24, 12, (8, 16, 32, 64, 128), 2, 32, 32),
                                 (3, 2, (4, 8), 1, 27, 24)])
def test_plan_metadata_matches_reference_naming(ua, cfg):
    cin, cout, feats, B, H, W = cfg
    L = ua.lib()
    rc, h = make_plan(L, cin, cout, feats, B, H, W)
    assert rc == 0
    try:
        params = plan_params(L, h)
        ref = O.init_state_dict(cin, cout, feats, seed=0)
        # construction order (the RNG order of the reference's constructor), reference shapes
        expect = [f"{pre}.{leaf}" for pre, *_ in O.construction_order(cin, cout, feats) for leaf in ("weight", "bias")]
        assert [p[0] for p in params] == expect
        for name, off, shape in params:
            assert shape == tuple(ref[name].shape)
            assert off % 64 == 0
        flat = L.mi_unet_plan_flat_elems(h)
        ends = sorted((off + int(np.prod(shape)), off) for _, off, shape in params)
        assert ends[-1][0] <= flat
        for (e0, _), (_, o1) in zip(ends, ends[1:]):
            assert o1 >= e0                                   # no overlap
        # backward segments: head, decoder blocks, bottleneck, encoder levels; descending, covering all
        segs = plan_segments(L, h)
        assert len(segs) == 2 * len(feats) + 2
        assert segs[0][1] == flat and segs[-1][0] == 0
        for (lo0, hi0), (lo1, hi1) in zip(segs, segs[1:]):
            assert lo0 < hi0 and hi1 == lo0
        assert L.mi_unet_plan_workspace_bytes(h) > 0
    finally:
        L.mi_unet_plan_destroy(h)


def test_plan_rejects_what_the_reference_cannot_build(ua):
    L = ua.lib()
    rc, _ = make_plan(L, 12, 1, (8, 16, 24), 1, 16, 16)      # features must double (unet2d.py:46-56 channel arithmetic)
    assert rc != 0 and b"double" in L.mi_last_error()
    rc, _ = make_plan(L, 0, 1, (8, 16), 1, 16, 16)
    assert rc != 0
    rc, _ = make_plan(L, 12, 1, (8, 16), 1, 1, 16)             # too small to pool once
    assert rc != 0


class Cfg:
    pass


def test_factory_mirrors_reference_builders(ua):
    cfg = Cfg()
    cfg.in_channels = 5
    torch.manual_seed(3)
    m = ua.build_model_from_cfg(cfg)                            # defaults: out_channels 1, features (32,64,128,256)
    ref = O.init_state_dict(5, 1, (32, 64, 128, 256), seed=3)
    sd = m.state_dict()
    assert list(sd.keys()) == [f"{pre}.{leaf}" for pre in O.state_dict_order(5, 1, (32, 64, 128, 256))
                               for leaf in ("weight", "bias")]
    for k in sd:
        assert sd[k].dtype == torch.float32 and torch.equal(sd[k], ref[k]), k
    cfg.model_type = "UNet"
    cfg.out_channels, cfg.unet_features, cfg.H_out, cfg.W_out = 3, (4, 8), 15, 24
    m2 = ua.build_model_from_cfg(cfg)
    assert m2.out_size == (15, 24) and tuple(m2.final_conv.weight.shape) == (3, 4, 1, 1)
    cfg.model_type = "coregan"
    with pytest.raises(ValueError):
        ua.build_model_from_cfg(cfg)
    cfg2 = Cfg()
    with pytest.raises(AttributeError):
        ua.build_model_from_cfg(cfg2)                           # in_channels is required (builders.py:93)
    # checkpoint interchange on the host, and no silent CPU execution
    m.load_state_dict({k: v + 1 for k, v in ref.items()})
    assert torch.equal(m.state_dict()["final_conv.bias"], ref["final_conv.bias"] + 1)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 5, 16, 16))


def test_bucket_planning(ua):
    segs = [(900, 1000), (600, 900), (300, 600), (20, 300), (0, 20)]
    b = ua.distributed.plan_buckets(segs, bucket_bytes=4 * 350)
    assert [x[:2] for x in b] == [(0, 2), (2, 5)]               # the 20-element tail rides in the last bucket
    assert b[0][2:] == (600, 1000) and b[1][2:] == (0, 600)
    covered = sorted((lo, hi) for _, _, lo, hi in b)
    assert covered[0][0] == 0 and covered[-1][1] == 1000
    one = ua.distributed.plan_buckets(segs, bucket_bytes=1 << 30)
    assert one == [(0, 5, 0, 1000)]
    with pytest.raises(ValueError):
        ua.shard_batch(torch.zeros(5, 1), 0, 2)


def _dp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import unet_amd
    torch.set_num_threads(2)
    r, w, _ = unet_amd.init_process_group_from_env(backend="gloo")
    cin, cout, feats, B, H, W = 12, 1, (4, 8, 16, 32), 4, 16, 16
    L = unet_amd.lib()
    rc, h = make_plan(L, cin, cout, feats, B // w, H, W)
    assert rc == 0
    params, segs, flat_n = plan_params(L, h), plan_segments(L, h), L.mi_unet_plan_flat_elems(h)
    L.mi_unet_plan_destroy(h)
    state = O.init_state_dict(cin, cout, feats, seed=0)
    X, Y = O.synthetic_batch(B, cin, cout, H, W)
    xs, ys = unet_amd.shard_batch(X, r, w), unet_amd.shard_batch(Y, r, w)
    grads = O.forward_backward(state, xs, ys)[3]                 # this rank's mean-loss gradient
    flat = torch.zeros(flat_n)
    for name, off, shape in params:
        flat[off:off + grads[name].numel()] = grads[name].reshape(-1)
    red = unet_amd.GradientReducer(bucket_bytes=16 << 10)
    assert red.world == w and len(unet_amd.distributed.plan_buckets(segs, 16 << 10)) > 1
    red.reduce_flat(flat, segs)
    torch.save({"flat": flat, "params": params}, os.path.join(out_dir, f"rank{r}.pt"))
    torch.distributed.destroy_process_group()


def test_data_parallel_average_equals_global_batch_gradient(tmp_path):
    """World size 2 over gloo: each rank's shard gradient, averaged bucket by bucket, equals the
    gradient of the mean loss over the global batch (what one process with the whole batch computes)."""
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    assert torch.equal(r0["flat"], r1["flat"])                   # every rank ends with the same buffer
    cin, cout, feats, B, H, W = 12, 1, (4, 8, 16, 32), 4, 16, 16
    state = O.init_state_dict(cin, cout, feats, seed=0)
    X, Y = O.synthetic_batch(B, cin, cout, H, W)
    gfull = O.forward_backward(state, X, Y)[3]
    for name, off, shape in r0["params"]:
        got = r0["flat"][off:off + gfull[name].numel()].reshape(gfull[name].shape)
        err = (got - gfull[name]).norm() / (gfull[name].norm() + 1e-30)
        assert err < 5e-6, (name, err.item())


def _collective_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import types
    import unet_amd
    torch.set_num_threads(2)
    unet_amd.init_process_group_from_env(backend="gloo")
    L = unet_amd.lib()
    rc, h = make_plan(L, 12, 1, (32, 64, 128, 256), 4, 32, 32)          # the default 4-level network
    assert rc == 0
    segs, flat_n = plan_segments(L, h), L.mi_unet_plan_flat_elems(h)
    L.mi_unet_plan_destroy(h)
    plan = types.SimpleNamespace(segments=segs)
    g = torch.Generator().manual_seed(100 + rank)
    mine = torch.randn(flat_n, generator=g)
    out = {}
    for mode in unet_amd.GradientReducer.MODES:
        red = unet_amd.GradientReducer(bucket_bytes=8 << 20, mode=mode)
        buckets = red.buckets_for(plan)
        assert len(buckets) >= 3 and buckets[0][0] == 0 and buckets[-1][1] == len(segs)
        flat = mine * red.grad_prescale                                  # what backward's alpha does on the device
        covered = 0
        for (seg_lo, seg_hi, lo, hi) in buckets:                         # the order the backward pass produces them in
            red.reduce_async(flat[lo:hi])
            covered += hi - lo
        red.wait_all()
        assert covered == flat_n and not red._pending
        out[mode] = flat
    # an odd-sized range: the part that does not split into whole shards rides in its own all-reduce
    red = unet_amd.GradientReducer(mode="rs_ag")
    odd = (mine[:1003] * 0.5).clone()
    red.reduce_async(odd)
    red.wait_all()
    out["odd"] = odd
    torch.save(out, os.path.join(out_dir, f"coll{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_reducer_modes_agree_in_backward_bucket_order(tmp_path):
    """World size 2 over gloo on CPU tensors: GradientReducer.reduce_async / wait_all driven through the
    bucket order of the default 4-level plan, once per collective mode.  All-reduce and reduce-scatter +
    all-gather must leave every rank with the same averaged buffer."""
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_collective_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "coll0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "coll1.pt"))
    n = r0["allreduce"].numel()
    expect = 0.5 * (torch.randn(n, generator=torch.Generator().manual_seed(100))
                    + torch.randn(n, generator=torch.Generator().manual_seed(101)))
    for mode in ("allreduce", "rs_ag"):
        assert torch.equal(r0[mode], r1[mode]), mode
        assert torch.allclose(r0[mode], expect, rtol=0, atol=1e-6), mode
    assert torch.equal(r0["allreduce"], r0["rs_ag"])                   # two-operand sums: the same bits either way
    assert torch.equal(r0["odd"], r1["odd"]) and torch.allclose(r0["odd"], expect[:1003], atol=1e-6)
    with pytest.raises(ValueError, match="unknown mode"):
        import unet_amd
        unet_amd.GradientReducer(mode="ring")


def test_bench_launches_its_own_ranks(monkeypatch, capsys):
    """`python bench.py --gpus N` without a launcher around it starts N workers through
    torch.distributed.run before touching a GPU, and refuses clearly when the machine has fewer GPUs."""
    import types
    import bench
    monkeypatch.delenv("RANK", raising=False)
    seen = {}

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)

    args = types.SimpleNamespace(gpus=4)
    rc = bench.self_launch(args, ["--gpus", "4", "--steps", "3"], device_count=lambda: 8, run=fake_run)
    assert rc == 7                                                      # the workers' status is the bench's status
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    seen.clear()
    rc = bench.self_launch(args, ["--gpus", "4"], device_count=lambda: 1, run=fake_run)
    assert rc != 0 and not seen
    assert "needs 4 GPUs" in capsys.readouterr().err
    # the whole entry point: no launcher environment, one visible GPU at most in this container
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2"])
    assert e.value.code not in (0, None)
