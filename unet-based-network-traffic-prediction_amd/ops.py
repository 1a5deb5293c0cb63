"""Thin tensor-level wrappers over the single-operator C entry points.

Activations are torch tensors in the library's layout: NHWC ``(B, H, W, Cpad)`` with
``Cpad = pad_ch(C)`` and dtype ``torch.float32`` or ``torch.bfloat16``.  Parameters and
parameter gradients are fp32 in the reference's layouts.  Every function enqueues on the
current torch stream and returns torch tensors it allocated; nothing here computes
with torch ops."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from ._lib import MI_BF16, MI_F32, check, lib


def pad_ch(c: int) -> int:
    """Stored channel count of a C-channel tensor (same rule as ``mi_pad_channels``)."""
    return 32 if c <= 32 else (c + 15) // 16 * 16


def mi_dtype(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return MI_F32
    if dt == torch.bfloat16:
        return MI_BF16
    raise TypeError(f"unsupported compute dtype {dt}: use torch.float32 or torch.bfloat16")


def torch_dtype(name) -> torch.dtype:
    if isinstance(name, torch.dtype):
        return name
    table = {"fp32": torch.float32, "float32": torch.float32, "f32": torch.float32,
             "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}
    try:
        return table[str(name).lower()]
    except KeyError:
        raise ValueError(f"unknown compute dtype {name!r} (fp32 or bf16)") from None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("this operator runs only as a HIP kernel on the GPU; got a CPU tensor "
                               "(there is no CPU fallback)")


def _view(t: torch.Tensor, coff: int = 0) -> Tuple[int, int, int]:
    """(pointer, channel stride, channel offset) of an NHWC tensor."""
    assert t.dim() == 4 and t.is_contiguous()
    return t.data_ptr(), t.shape[3], coff


# ------------------------------------------------------------------ layout
def to_nhwc(x: torch.Tensor, dtype=torch.float32, out: Optional[torch.Tensor] = None, coff: int = 0):
    _need_cuda(x)
    x = x.contiguous().float()
    B, Cc, H, W = x.shape
    if out is None:
        out = torch.empty((B, H, W, pad_ch(Cc)), device=x.device, dtype=dtype)
    check(lib().mi_nchw_to_nhwc(x.data_ptr(), out.data_ptr(), out.shape[3], coff, B, Cc, H, W,
                                mi_dtype(out.dtype), _stream()), "nchw_to_nhwc")
    return out


def from_nhwc(t: torch.Tensor, channels: int, coff: int = 0) -> torch.Tensor:
    _need_cuda(t)
    B, H, W, cs = t.shape
    out = torch.empty((B, channels, H, W), device=t.device, dtype=torch.float32)
    check(lib().mi_nhwc_to_nchw(t.data_ptr(), cs, coff, out.data_ptr(), B, channels, H, W,
                                mi_dtype(t.dtype), _stream()), "nhwc_to_nchw")
    return out


# ------------------------------------------------------------------ packing
def pack_conv3x3(weight: torch.Tensor, dtype, split: int = 0, want_bwd: bool = True):
    _need_cuda(weight)
    Cout, Cin = weight.shape[:2]
    L, dt = lib(), mi_dtype(dtype)
    es = 2 if dtype == torch.bfloat16 else 4
    wf = torch.empty(L.mi_packed_conv3x3_bytes(Cout, Cin, split, dt, 0) // es, device=weight.device, dtype=dtype)
    wb = torch.empty(L.mi_packed_conv3x3_bytes(Cout, Cin, split, dt, 1) // es, device=weight.device,
                     dtype=dtype) if want_bwd else None
    check(L.mi_pack_conv3x3(weight.contiguous().data_ptr(), wf.data_ptr(), wb.data_ptr() if want_bwd else None,
                            Cout, Cin, split, dt, _stream()), "pack_conv3x3")
    return wf, wb


def pack_convt2x2(weight: torch.Tensor, dtype, want_bwd: bool = True):
    _need_cuda(weight)
    Cin, Cout = weight.shape[:2]
    L, dt = lib(), mi_dtype(dtype)
    es = 2 if dtype == torch.bfloat16 else 4
    wf = torch.empty(L.mi_packed_convt2x2_bytes(Cin, Cout, dt, 0) // es, device=weight.device, dtype=dtype)
    wb = torch.empty(L.mi_packed_convt2x2_bytes(Cin, Cout, dt, 1) // es, device=weight.device,
                     dtype=dtype) if want_bwd else None
    check(L.mi_pack_convt2x2(weight.contiguous().data_ptr(), wf.data_ptr(), wb.data_ptr() if want_bwd else None,
                             Cin, Cout, dt, _stream()), "pack_convt2x2")
    return wf, wb


# ------------------------------------------------------------------ 3x3 conv
def conv3x3_relu_fwd(x, wfwd, bias, Cin, Cout, split=0, relu=True, out=None, out_coff=0):
    B, H, W, _ = x.shape
    if out is None:
        out = torch.empty((B, H, W, pad_ch(Cout)), device=x.device, dtype=x.dtype)
    check(lib().mi_conv3x3_relu_fwd(*_view(x), wfwd.data_ptr(), bias.data_ptr(), *_view(out, out_coff),
                                    B, H, W, Cin, split, Cout, int(relu), mi_dtype(x.dtype), _stream()), "conv3x3_fwd")
    return out


def conv3x3_dgrad(dy, wbwd, Cin, Cout, split=0, mask=None, dy_coff=0):
    B, H, W, _ = dy.shape
    cp = pad_ch(split) + pad_ch(Cin - split) if split else pad_ch(Cin)
    dx = torch.empty((B, H, W, cp), device=dy.device, dtype=dy.dtype)
    mv = _view(mask) if mask is not None else (None, 0, 0)
    check(lib().mi_conv3x3_dgrad(*_view(dy, dy_coff), wbwd.data_ptr(), *_view(dx), *mv,
                                 B, H, W, Cin, split, Cout, mi_dtype(dy.dtype), _stream()), "conv3x3_dgrad")
    return dx


def conv3x3_wgrad(x, dy, Cin, Cout, split=0, dW=None, db=None, alpha=1.0, beta=0.0, dy_coff=0):
    B, H, W, _ = x.shape
    dt = mi_dtype(x.dtype)
    if dW is None:
        dW = torch.empty((Cout, Cin, 3, 3), device=x.device, dtype=torch.float32)
        db = torch.empty((Cout,), device=x.device, dtype=torch.float32)
    nbytes = lib().mi_conv3x3_wgrad_ws_bytes(B, H, W, Cin, split, Cout, dt)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    check(lib().mi_conv3x3_wgrad(*_view(x), *_view(dy, dy_coff), dW.data_ptr(), db.data_ptr(), alpha, beta,
                                 ws.data_ptr(), nbytes, B, H, W, Cin, split, Cout, dt, _stream()), "conv3x3_wgrad")
    return dW, db


# ------------------------------------------------------------------ max-pool
def maxpool2_fwd(x, C, x_coff=0):
    B, H, W, _ = x.shape
    y = torch.empty((B, H // 2, W // 2, pad_ch(C)), device=x.device, dtype=x.dtype)
    check(lib().mi_maxpool2_fwd(*_view(x, x_coff), *_view(y), B, H, W, C, mi_dtype(x.dtype), _stream()), "maxpool2_fwd")
    return y


def maxpool2_bwd(x, dy_pool, C, dskip=None, relu_mask=False, x_coff=0):
    B, H, W, _ = x.shape
    dx = torch.empty((B, H, W, pad_ch(C)), device=x.device, dtype=x.dtype)
    sv = _view(dskip) if dskip is not None else (None, 0, 0)
    check(lib().mi_maxpool2_bwd(*_view(x, x_coff), *_view(dy_pool), *sv, *_view(dx), B, H, W, C, int(relu_mask),
                                mi_dtype(x.dtype), _stream()), "maxpool2_bwd")
    return dx


# ------------------------------------------------------------------ transposed conv
def convt2x2_fwd(x, wfwd, bias, Cin, Cout, out=None, out_coff=0):
    B, H, W, _ = x.shape
    if out is None:
        out = torch.empty((B, 2 * H, 2 * W, pad_ch(Cout)), device=x.device, dtype=x.dtype)
    check(lib().mi_convt2x2_fwd(*_view(x), wfwd.data_ptr(), bias.data_ptr(), *_view(out, out_coff), B, H, W, Cin, Cout,
                                mi_dtype(x.dtype), _stream()), "convt2x2_fwd")
    return out


def convt2x2_dgrad(dy, wbwd, Cin, Cout, mask=None, dy_coff=0):
    B, H2, W2, _ = dy.shape
    H, W = H2 // 2, W2 // 2
    dx = torch.empty((B, H, W, pad_ch(Cin)), device=dy.device, dtype=dy.dtype)
    mv = _view(mask) if mask is not None else (None, 0, 0)
    check(lib().mi_convt2x2_dgrad(*_view(dy, dy_coff), wbwd.data_ptr(), *_view(dx), *mv, B, H, W, Cin, Cout,
                                  mi_dtype(dy.dtype), _stream()), "convt2x2_dgrad")
    return dx


def convt2x2_wgrad(x, dy, Cin, Cout, alpha=1.0, beta=0.0, dy_coff=0):
    B, H, W, _ = x.shape
    dt = mi_dtype(x.dtype)
    dW = torch.empty((Cin, Cout, 2, 2), device=x.device, dtype=torch.float32)
    db = torch.empty((Cout,), device=x.device, dtype=torch.float32)
    nbytes = lib().mi_convt2x2_wgrad_ws_bytes(B, H, W, Cin, Cout, dt)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    check(lib().mi_convt2x2_wgrad(*_view(x), *_view(dy, dy_coff), dW.data_ptr(), db.data_ptr(), alpha, beta,
                                  ws.data_ptr(), nbytes, B, H, W, Cin, Cout, dt, _stream()), "convt2x2_wgrad")
    return dW, db


# ------------------------------------------------------------------ head
def head_fwd(x, weight, bias, Cin, Cout):
    B, H, W, _ = x.shape
    pred = torch.empty((B, Cout, H, W), device=x.device, dtype=torch.float32)
    check(lib().mi_conv1x1_head_fwd(*_view(x), weight.data_ptr(), bias.data_ptr(), pred.data_ptr(), B, H, W, Cin, Cout,
                                    mi_dtype(x.dtype), _stream()), "head_fwd")
    return pred


def head_bwd(x, weight, dpred, Cin, Cout, relu_mask=True, alpha=1.0, beta=0.0):
    B, H, W, _ = x.shape
    dx = torch.empty_like(x)
    dW = torch.empty((Cout, Cin, 1, 1), device=x.device, dtype=torch.float32)
    db = torch.empty((Cout,), device=x.device, dtype=torch.float32)
    nbytes = lib().mi_conv1x1_head_ws_bytes(Cin, Cout)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    check(lib().mi_conv1x1_head_bwd(*_view(x), weight.data_ptr(), dpred.contiguous().data_ptr(), *_view(dx),
                                    dW.data_ptr(), db.data_ptr(), alpha, beta, ws.data_ptr(), nbytes, B, H, W, Cin, Cout,
                                    int(relu_mask), mi_dtype(x.dtype), _stream()), "head_bwd")
    return dx, dW, db


# ------------------------------------------------------------------ loss
def mse_fwd_bwd(pred, target, want_grad=True, gscale=1.0, gscale_dev: Optional[torch.Tensor] = None, want_sums=True):
    """Returns (out3, dpred): out3 = [mean sq err, sum |d|, sum d^2] (device fp32), or None without want_sums."""
    _need_cuda(pred, target)
    if pred.shape != target.shape:
        raise ValueError(f"mse: shape mismatch {tuple(pred.shape)} vs {tuple(target.shape)}")
    pred, target = pred.contiguous(), target.contiguous()
    n = pred.numel()
    out3 = torch.empty(3, device=pred.device, dtype=torch.float32) if want_sums else None
    dpred = torch.empty_like(pred) if want_grad else None
    nbytes = lib().mi_mse_ws_bytes(n)
    ws = torch.empty(nbytes, device=pred.device, dtype=torch.uint8)
    check(lib().mi_mse_fwd_bwd(pred.data_ptr(), target.data_ptr(), n, out3.data_ptr() if want_sums else None,
                               dpred.data_ptr() if want_grad else None, gscale,
                               gscale_dev.data_ptr() if gscale_dev is not None else None,
                               ws.data_ptr(), nbytes, _stream()), "mse")
    return out3, dpred


# ------------------------------------------------------------------ bilinear
def _strides_nchw(t):
    return t.stride(0), t.stride(1), t.stride(2), t.stride(3)


def bilinear_fwd_nchw(x: torch.Tensor, size) -> torch.Tensor:
    _need_cuda(x)
    x = x.contiguous()
    B, Cc, H, W = x.shape
    out = torch.empty((B, Cc, size[0], size[1]), device=x.device, dtype=x.dtype)
    check(lib().mi_bilinear_fwd(x.data_ptr(), *_strides_nchw(x), mi_dtype(x.dtype), out.data_ptr(), *_strides_nchw(out),
                                mi_dtype(out.dtype), B, Cc, H, W, size[0], size[1], _stream()), "bilinear_fwd")
    return out


def bilinear_bwd_nchw(dout: torch.Tensor, in_size) -> torch.Tensor:
    _need_cuda(dout)
    dout = dout.contiguous()
    B, Cc, Ho, Wo = dout.shape
    din = torch.empty((B, Cc, in_size[0], in_size[1]), device=dout.device, dtype=dout.dtype)
    check(lib().mi_bilinear_bwd(dout.data_ptr(), *_strides_nchw(dout), mi_dtype(dout.dtype), din.data_ptr(),
                                *_strides_nchw(din), mi_dtype(din.dtype), B, Cc, in_size[0], in_size[1], Ho, Wo,
                                _stream()), "bilinear_bwd")
    return din


# ------------------------------------------------------------------ optimiser
def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    _need_cuda(p, g, m, v)
    check(lib().mi_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps,
                             step, grad_scale, _stream()), "adam_step")
