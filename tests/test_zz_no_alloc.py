"""The allocation-free weight-gradient entry point (mi_conv3x3_wgrad_static: partial sums in a scratch
array built into the library) against the entry point it is a variant of (mi_conv3x3_wgrad, partial
sums in a caller workspace).  Same kernels and the same fixed-order combine, so dW and db must be
bit-identical (torch.equal) -- on two calls in a row that share the scratch, the second on a smaller
input, and again after a call of another shape has left its partials behind."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def rnd(shape, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) - 0.4).to(dtype).cuda()


def both(L, ops, B, H, W, Cin, Cout, dtype, alpha=1.0, beta=0.0, seed=0):
    """(dW, db) from the workspace entry point and from the static one, each through the C ABI."""
    dt = ops.mi_dtype(dtype)
    x = torch.zeros((B, H, W, ops.pad_ch(Cin)), device="cuda", dtype=dtype)
    dy = torch.zeros((B, H, W, ops.pad_ch(Cout)), device="cuda", dtype=dtype)
    x[..., :Cin] = rnd((B, H, W, Cin), seed + 1, dtype)
    dy[..., :Cout] = rnd((B, H, W, Cout), seed + 2, dtype)
    init_w, init_b = rnd((Cout, Cin, 3, 3), seed + 3, torch.float32), rnd((Cout,), seed + 4, torch.float32)
    stream = torch.cuda.current_stream().cuda_stream
    res = []
    for static in (False, True):
        dW, db = init_w.clone(), init_b.clone()
        view = lambda t: (t.data_ptr(), t.shape[3], 0)
        if static:
            free0 = torch.cuda.mem_get_info()[0]
            rc = L.mi_conv3x3_wgrad_static(*view(x), *view(dy), dW.data_ptr(), db.data_ptr(), alpha, beta,
                                           B, H, W, Cin, 0, Cout, dt, stream)
            assert torch.cuda.mem_get_info()[0] == free0          # nothing was allocated on the device
        else:
            nbytes = L.mi_conv3x3_wgrad_ws_bytes(B, H, W, Cin, 0, Cout, dt)
            ws = torch.empty(nbytes, device="cuda", dtype=torch.uint8)
            rc = L.mi_conv3x3_wgrad(*view(x), *view(dy), dW.data_ptr(), db.data_ptr(), alpha, beta, ws.data_ptr(), nbytes,
                                    B, H, W, Cin, 0, Cout, dt, stream)
        assert rc == 0, L.mi_last_error()
        res.append((dW, db))
    torch.cuda.synchronize()
    return res


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_static_scratch_wgrad_equals_workspace_wgrad(dtype):
    import unet_amd
    L, ops = unet_amd.lib(), unet_amd.ops
    assert L.mi_conv3x3_wgrad_static_max_bytes() == 32 << 20
    # first call: a tiling shape (halo-tile kernel in bf16, many slabs); second call right after it:
    # a smaller input of another kernel family (ragged, generic kernel) over the scratch the first
    # left behind; then back to a third, tiny-image shape
    shapes = [(64, 16, 16, 64, 64), (3, 9, 7, 20, 24), (70, 4, 4, 128, 64), (64, 16, 16, 64, 64)]
    for i, (B, H, W, Cin, Cout) in enumerate(shapes):
        (w0, b0), (w1, b1) = both(L, ops, B, H, W, Cin, Cout, dtype, seed=10 * i)
        assert w0.abs().max().item() > 0
        assert torch.equal(w0, w1) and torch.equal(b0, b1), (dtype, shapes[i])
    # accumulate form: dW = alpha * sum + beta * dW
    (w0, b0), (w1, b1) = both(L, ops, 8, 8, 8, 32, 48, dtype, alpha=0.25, beta=1.0, seed=77)
    assert torch.equal(w0, w1) and torch.equal(b0, b1)


def test_static_scratch_rejects_shapes_beyond_its_size():
    import unet_amd
    L = unet_amd.lib()
    B, H, W, Cin, Cout = 64, 4, 4, 1024, 1024                     # 36 MiB of partials
    assert L.mi_conv3x3_wgrad_ws_bytes(B, H, W, Cin, 0, Cout, 1) > L.mi_conv3x3_wgrad_static_max_bytes()
    x = torch.zeros((B, H, W, Cin), device="cuda", dtype=torch.bfloat16)
    dW = torch.zeros((Cout, Cin, 3, 3), device="cuda")
    db = torch.zeros(Cout, device="cuda")
    rc = L.mi_conv3x3_wgrad_static(x.data_ptr(), Cin, 0, x.data_ptr(), Cout, 0, dW.data_ptr(), db.data_ptr(), 1.0, 0.0,
                                   B, H, W, Cin, 0, Cout, 1, torch.cuda.current_stream().cuda_stream)
    assert rc == -2                                               # MI_ESIZE, nothing launched
    assert b"static scratch holds" in L.mi_last_error()
    torch.cuda.synchronize()
    assert dW.abs().max().item() == 0
