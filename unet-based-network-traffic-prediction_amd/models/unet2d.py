"""Drop-in ``UNet`` whose forward and backward run as hand-written gfx950 HIP kernels.

Same constructor, parameter names, shapes, initialisation and call signature as the
reference ``UNet`` (models/unet2d.py:8-102; the class the reference factory actually
builds is the identical copy at models/cells.py:425-519).  What differs is everything
under ``forward``: the module holds no compute of its own -- one C call enqueues the
whole forward launch sequence, one (segmented) C call the whole backward.

Memory model
  * all parameters are views into ONE flat fp32 buffer (construction order, 64-element
    aligned) so the optimiser step and the data-parallel gradient reduction run over
    contiguous ranges;
  * each (batch, H, W) shape gets a plan and a workspace holding every activation, every
    activation gradient, the packed MFMA weight operands and the weight-gradient partials;
  * gradients of a backward pass land in one flat fp32 buffer, handed to autograd as views.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .._lib import (MI_FWD_REPACK, MI_FWD_TRAIN, MI_OPT_DEBUG_STREAMS, MI_OPT_GRAPH, MI_OPT_PACK_ONE_STREAM,
                    MI_OPT_SERIAL_BACKWARD, UNetDesc, check, lib)
from ..ops import mi_dtype, torch_dtype

# The reference trains and evaluates in fp32 with no autocast (train.py:495-498), so the drop-in default
# is the exact-fp32 path; bf16 storage + bf16 MFMA is opt-in (compute_dtype="bf16", cfg.compute_dtype,
# or UNET_TM_DTYPE=bf16 in the environment).
DEFAULT_COMPUTE_DTYPE = os.environ.get("UNET_TM_DTYPE", "fp32")


class _Plan:
    """Shape-specific launch plan plus the workspace it runs in."""

    def __init__(self, model: "UNet", B: int, H: int, W: int, device: torch.device, need_input_grad: bool):
        L = lib()
        d = UNetDesc()
        d.in_channels, d.out_channels = model.in_channels, model.out_channels
        d.n_levels = len(model.features)
        for i, f in enumerate(model.features):
            d.features[i] = f
        d.batch, d.height, d.width = B, H, W
        d.out_h, d.out_w = model.out_size if model.out_size is not None else (0, 0)
        d.dtype = mi_dtype(model.compute_dtype)
        d.need_input_grad = int(need_input_grad)
        handle = C.c_void_p()
        check(L.mi_unet_plan_create(C.byref(d), C.byref(handle)), "plan_create")
        self.handle = handle
        self.B, self.H, self.W = B, H, W
        self.need_input_grad = bool(need_input_grad)
        self.ws_bytes = L.mi_unet_plan_workspace_bytes(handle)
        self.ws = torch.empty(self.ws_bytes, device=device, dtype=torch.uint8)
        self.out_hw = model.out_size if model.out_size is not None else (H, W)
        self.n_segments = L.mi_unet_plan_num_segments(handle)
        self.segments = []
        for s in range(self.n_segments):
            lo, hi = C.c_int64(), C.c_int64()
            check(L.mi_unet_plan_segment_range(handle, s, C.byref(lo), C.byref(hi)), "segment_range")
            self.segments.append((lo.value, hi.value))
        self.packed_version = None   # parameter version the packed weights correspond to
        self.packed_train = False
        self.ws_ready = False        # a forward pass has packed into this workspace at least once
        self.adam_pack_ok = bool(L.mi_unet_plan_adam_pack_supported(handle))   # FusedAdam may step and pack in one launch
        # A plan's workspace holds the activations of ONE forward pass.  `generation` counts the
        # passes run on it; `leased` is the generation whose backward is still to come (None = free).
        self.generation = 0
        self.leased: Optional[int] = None
        # The library reads no environment itself: the switches are plan options, set here.
        self.graph_mode = _env_flag("UNET_TM_GRAPH")
        for opt, on in ((MI_OPT_GRAPH, self.graph_mode), (MI_OPT_SERIAL_BACKWARD, _env_flag("UNET_TM_SERIAL_BACKWARD")),
                        (MI_OPT_DEBUG_STREAMS, _env_flag("UNET_TM_DEBUG_STREAMS")),
                        (MI_OPT_PACK_ONE_STREAM, _env_flag("UNET_TM_PACK_ONE_STREAM"))):
            if on:
                check(L.mi_unet_plan_set_option(handle, opt, 1), "plan_set_option")

    def tensors(self) -> Dict[str, dict]:
        """Named activation / gradient views inside the workspace (for parity checks)."""
        L = lib()
        out = {}
        for i in range(L.mi_unet_plan_num_tensors(self.handle)):
            name = C.create_string_buffer(96)
            off, dims, cs, co, lay = C.c_int64(), (C.c_int * 4)(), C.c_int(), C.c_int(), C.c_int()
            check(L.mi_unet_plan_tensor_info(self.handle, i, name, 96, C.byref(off), C.byref(dims), C.byref(cs),
                                             C.byref(co), C.byref(lay)), "tensor_info")
            out[name.value.decode()] = dict(offset=off.value, dims=tuple(dims), cstride=cs.value, coff=co.value,
                                            layout=lay.value)
        return out

    def __del__(self):
        try:
            if self.handle:
                lib().mi_unet_plan_destroy(self.handle)
        except Exception:
            pass


def _env_flag(name: str) -> bool:
    """Environment switch read when a plan is created (UNET_TM_GRAPH, UNET_TM_SERIAL_BACKWARD,
    UNET_TM_DEBUG_STREAMS, UNET_TM_PACK_ONE_STREAM): "1" turns it on."""
    return os.environ.get(name, "")[:1] == "1"


class _Lease:
    """Keeps a plan's activations reserved for the autograd node of one forward pass: released
    when its backward has run, or when the node dies without one."""

    __slots__ = ("plan", "generation")

    def __init__(self, plan: _Plan):
        plan.generation += 1
        plan.leased = plan.generation
        self.plan, self.generation = plan, plan.generation

    def intact(self) -> bool:
        return self.plan.generation == self.generation

    def release(self):
        if self.plan.leased == self.generation:
            self.plan.leased = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class _NoGuard:
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on_device(device: torch.device):
    """Context that makes ``device`` the current one; free when it already is (the usual case)."""
    if device.index is None or device.index == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


def _contiguous_strides(shape):
    st = [1] * len(shape)
    for i in range(len(shape) - 2, -1, -1):
        st[i] = st[i + 1] * int(shape[i + 1])
    return tuple(st)


class _UNetFunction(torch.autograd.Function):
    """Autograd node for the whole network: forward = one launch sequence, backward = one
    segmented launch sequence writing a flat gradient buffer."""

    @staticmethod
    def forward(ctx, model, x, *params):
        ctx.model = model
        ctx.x_requires_grad = bool(ctx.needs_input_grad[1])
        ctx.plan = model._forward_impl(x, train=True, need_input_grad=ctx.x_requires_grad)
        ctx.lease = _Lease(ctx.plan)
        return model._last_pred

    @staticmethod
    def backward(ctx, dpred):
        model, plan = ctx.model, ctx.plan
        if not ctx.lease.intact():
            raise RuntimeError("UNet: the activations of this forward pass are gone (its backward already ran and "
                               "a later forward pass reused the workspace); run forward again before backward")
        flat_grad, dx = model._backward_impl(plan, dpred, out=None, beta=0.0)
        ctx.lease.release()
        spec = model._grad_view_spec
        if spec is None or len(spec) != len(model._flat_layout):
            spec = model._grad_view_spec = [(tuple(shape), _contiguous_strides(shape), o) for (o, n, shape) in model._flat_layout]
        # one as_strided per parameter: half the host time of slice + view (46 tensors every step)
        grads = tuple(flat_grad.as_strided(shape, strides, o) for (shape, strides, o) in spec)
        return (None, dx if ctx.x_requires_grad else None) + grads


class UNet(nn.Module):
    """U-Net for traffic-matrix prediction, MI355X-native.

    Arguments are the reference's (models/unet2d.py:9-15) plus ``compute_dtype``:
    ``"fp32"`` (the default, as the reference computes: exact-fp32 MFMA throughout) or ``"bf16"``
    (bf16 storage + bf16 MFMA with fp32 accumulation; parameters, gradients and optimiser state
    stay fp32)."""

    def __init__(self, in_channels=1, out_channels=1, features=(32, 64, 128, 256), out_size=None,
                 compute_dtype=None):
        super().__init__()
        self.in_channels, self.out_channels = int(in_channels), int(out_channels)
        self.features = tuple(int(f) for f in features)
        self.out_size = tuple(out_size) if out_size is not None else None
        self.compute_dtype = torch_dtype(compute_dtype if compute_dtype is not None else DEFAULT_COMPUTE_DTYPE)

        # Parameter containers.  They exist so that state_dict keys, shapes, default
        # initialisation and RNG consumption order equal the reference's; forward() never
        # calls them.
        self.downs = nn.ModuleList()
        self.ups = nn.ModuleList()
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2)
        ch = self.in_channels
        for f in self.features:
            self.downs.append(nn.Sequential(nn.Conv2d(ch, f, 3, padding=1), nn.ReLU(inplace=True),
                                            nn.Conv2d(f, f, 3, padding=1), nn.ReLU(inplace=True)))
            ch = f
        fb = self.features[-1]
        self.bottleneck = nn.Sequential(nn.Conv2d(fb, 2 * fb, 3, padding=1), nn.ReLU(inplace=True),
                                        nn.Conv2d(2 * fb, 2 * fb, 3, padding=1), nn.ReLU(inplace=True))
        for f in reversed(self.features):
            self.ups.append(nn.ConvTranspose2d(2 * f, f, kernel_size=2, stride=2))
            self.ups.append(nn.Sequential(nn.Conv2d(2 * f, f, 3, padding=1), nn.ReLU(inplace=True),
                                          nn.Conv2d(f, f, 3, padding=1), nn.ReLU(inplace=True)))
        self.final_conv = nn.Conv2d(self.features[0], self.out_channels, kernel_size=1)

        self._plans: Dict[Tuple, List[_Plan]] = {}
        self._flat: Optional[torch.Tensor] = None
        self._flat_layout: List[Tuple[int, int, torch.Size]] = []
        self._last_pred: Optional[torch.Tensor] = None
        self._dp = None               # set by distributed.GradientReducer.attach
        self._grad_stage: Optional[torch.Tensor] = None
        self._grad_view_spec = None
        self.last_plan: Optional[_Plan] = None
        self.last_train_plan: Optional[_Plan] = None      # the plan of the last training-mode forward (FusedAdam packs into it)
        self._step_grad: Optional[torch.Tensor] = None     # train_step(): flat gradients and their views
        self._step_views = None

    # ------------------------------------------------------------------ parameters
    def _construction_order(self) -> List[nn.Parameter]:
        """Parameters in the order the layers are constructed = forward execution order
        (downs, bottleneck, ups, final_conv); the flat buffer uses this order so that
        gradients become ready back-to-front during backward.  Cached: walking the module
        containers costs more host time per step than everything else in forward()."""
        cached = self.__dict__.get("_param_list")
        if cached is not None:
            return cached
        mods = []
        for blk in self.downs:
            mods += [blk[0], blk[2]]
        mods += [self.bottleneck[0], self.bottleneck[2]]
        for i in range(0, len(self.ups), 2):
            mods += [self.ups[i], self.ups[i + 1][0], self.ups[i + 1][2]]
        mods.append(self.final_conv)
        out = []
        for m in mods:
            out += [m.weight, m.bias]
        self.__dict__["_param_list"] = out
        return out

    def _apply(self, fn, *a, **kw):
        # .to()/.cuda()/.float() replace parameter storage: the flat buffer must be rebuilt
        self._flat = None
        self._grad_stage = None
        self._step_grad = None
        self._step_views = None
        self._plans.clear()
        self.last_plan = self.last_train_plan = None
        self.__dict__.pop("_param_list", None)
        try:
            return super()._apply(fn, *a, **kw)
        finally:
            self.__dict__.pop("_param_list", None)   # _apply may replace Parameter objects

    def _ensure_flat(self, device: torch.device) -> torch.Tensor:
        params = self._construction_order()
        if self._flat is not None and self._flat.device == device:
            base = self._flat.data_ptr()
            if all(p.data_ptr() == base + 4 * o for p, (o, _, _) in zip(params, self._flat_layout)):
                return self._flat
        for p in params:
            if p.device != device or p.dtype != torch.float32:
                raise RuntimeError(f"UNet parameters must be float32 on {device} (found {p.dtype} on {p.device}); "
                                   "call model.to(device) first")
        layout, off = [], 0
        for p in params:
            layout.append((off, p.numel(), p.shape))
            off = (off + p.numel() + 63) // 64 * 64
        flat = torch.zeros(off, device=device, dtype=torch.float32)
        with torch.no_grad():
            for p, (o, n, shape) in zip(params, layout):
                view = flat[o:o + n].view(shape)
                view.copy_(p.data)
                p.data = view
        self._flat, self._flat_layout = flat, layout
        self._grad_view_spec = None
        self._step_grad = None
        self._step_views = None
        for plans in self._plans.values():
            for plan in plans:
                plan.packed_version = None
        return flat

    def flat_parameters(self) -> torch.Tensor:
        """The flat fp32 buffer all parameters are views of (built on first use)."""
        return self._ensure_flat(next(self.parameters()).device)

    def _param_version(self) -> int:
        return sum(p._version for p in self._construction_order())

    def mark_parameters_changed(self):
        """Tell the module its parameters were updated through raw pointers (the fused
        optimiser does this) so packed weights are refreshed on the next forward."""
        for plans in self._plans.values():
            for plan in plans:
                plan.packed_version = None

    # ------------------------------------------------------------------ plans
    def _get_plan(self, x: torch.Tensor, need_input_grad: bool) -> _Plan:
        B, Cc, H, W = x.shape
        key = (B, H, W, x.device.index, bool(need_input_grad), self.compute_dtype)
        # Normally one plan per shape; a forward pass that arrives while an earlier one still
        # awaits its backward (two passes in one graph, gradient checkpoints, an evaluation between
        # forward and backward) gets a plan -- and a workspace -- of its own.
        plans = self._plans.setdefault(key, [])
        for plan in plans:
            if plan.leased is None:
                return plan
        plan = _Plan(self, B, H, W, x.device, need_input_grad)
        flat_elems = lib().mi_unet_plan_flat_elems(plan.handle)
        if flat_elems != self._flat.numel():
            raise RuntimeError("internal error: flat parameter layout disagrees with the plan")
        plans.append(plan)
        return plan

    # ------------------------------------------------------------------ forward / backward
    def _check_input(self, x: torch.Tensor):
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise ValueError("UNet expects a 4-D (B, C, H, W) tensor")
        if not x.is_cuda:
            raise RuntimeError("this UNet runs only on an AMD GPU through its HIP kernels; got a CPU tensor "
                               "(there is no CPU fallback -- move the model and the input to 'cuda')")
        if x.shape[1] != self.in_channels:
            raise ValueError(f"expected {self.in_channels} input channels, got {x.shape[1]}")

    def _forward_impl(self, x: torch.Tensor, train: bool, need_input_grad: bool = False) -> _Plan:
        self._check_input(x)
        flat = self._ensure_flat(x.device)
        x = x.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        plan = self._get_plan(x, need_input_grad)
        flags = MI_FWD_TRAIN if train else 0
        version = self._param_version()
        if plan.packed_version != version or (train and not plan.packed_train):
            flags |= MI_FWD_REPACK
        oh, ow = plan.out_hw
        pred = torch.empty((plan.B, self.out_channels, oh, ow), device=x.device, dtype=torch.float32)
        # the library launches on the CURRENT device (its zero page, its stream choice): make that the tensors' device
        with _on_device(x.device):
            check(lib().mi_unet_forward(plan.handle, x.data_ptr(), flat.data_ptr(), plan.ws.data_ptr(), plan.ws_bytes,
                                        pred.data_ptr(), flags, torch.cuda.current_stream(x.device).cuda_stream),
                  "unet_forward")
        if flags & MI_FWD_REPACK:
            plan.packed_version, plan.packed_train = version, train
            plan.ws_ready = True
        self._last_pred = pred
        self.last_plan = plan
        if train:
            self.last_train_plan = plan
        return plan

    def _backward_impl(self, plan: _Plan, dpred: torch.Tensor, out: Optional[torch.Tensor], beta: float,
                       alpha: float = 1.0):
        """Run backward over all segments into ``out`` (a flat fp32 buffer; allocated when
        None).  With a gradient reducer attached, each finished bucket's all-reduce is
        started while the remaining segments are still being enqueued."""
        with _on_device(self._flat.device):
            return self._backward_on_device(plan, dpred, out, beta, alpha)

    def _backward_on_device(self, plan: _Plan, dpred: torch.Tensor, out: Optional[torch.Tensor], beta: float,
                            alpha: float):
        flat = self._flat
        dpred = dpred.contiguous()
        if dpred.dtype != torch.float32:
            dpred = dpred.float()
        # Autograd may keep, or accumulate into, what it is handed: every pass gets a flat buffer of
        # its own (the caching allocator hands the previous step's back).  Only with captured launch
        # graphs (UNET_TM_GRAPH=1) are gradients produced in a buffer that lives as long as the
        # module and copied out, because its address is part of the key of the captured graphs.
        staged = out is None and plan.graph_mode
        if staged:
            if self._grad_stage is None or self._grad_stage.shape != flat.shape or self._grad_stage.device != flat.device:
                self._grad_stage = torch.zeros_like(flat)
            out = self._grad_stage
        elif out is None:
            out = torch.empty_like(flat)
        dx = None
        if plan.need_input_grad:
            dx = torch.empty((plan.B, self.in_channels, plan.H, plan.W), device=flat.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(flat.device).cuda_stream
        L = lib()
        dp = self._dp
        if dp is None:
            check(L.mi_unet_backward(plan.handle, dpred.data_ptr(), flat.data_ptr(), plan.ws.data_ptr(), plan.ws_bytes,
                                     out.data_ptr(), alpha, beta, 0, plan.n_segments,
                                     dx.data_ptr() if dx is not None else None, stream), "unet_backward")
        else:
            alpha = alpha * dp.grad_prescale
            # A finished bucket is reduced behind the plan's weight-gradient stream (which has caught up
            # with this call's work on the current stream) so the next call's dgrad chain does not wait
            # for the bucket's weight gradients; the last call joins the streams, its bucket goes on the
            # current stream, and the waits below are issued there.
            side = L.mi_unet_plan_side_stream(plan.handle, stream)
            ext = None
            if side and not plan.graph_mode:      # a captured launch sequence must end joined: no deferred join there
                ext = plan.side_stream if getattr(plan, "side_stream", None) is not None else torch.cuda.ExternalStream(side)
                plan.side_stream = ext
            check(L.mi_unet_plan_defer_join(plan.handle, 1 if ext is not None else 0), "unet_plan_defer_join")
            for (seg_lo, seg_hi, lo, hi) in dp.buckets_for(plan):
                check(L.mi_unet_backward(plan.handle, dpred.data_ptr(), flat.data_ptr(), plan.ws.data_ptr(),
                                         plan.ws_bytes, out.data_ptr(), alpha, beta, seg_lo, seg_hi,
                                         dx.data_ptr() if dx is not None else None, stream), "unet_backward")
                # the library says where the finished range is complete: behind the side stream (deferred
                # join) or, when the call joined the streams, in the current stream's order
                if ext is not None and L.mi_unet_plan_last_call_deferred(plan.handle):
                    with torch.cuda.stream(ext):
                        dp.reduce_async(out[lo:hi])
                else:
                    dp.reduce_async(out[lo:hi])
            dp.wait_all()
        if staged:
            out = out.clone()
        return out, dx

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        params = self._construction_order()
        needs_graph = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        if needs_graph:
            self._check_input(x)
            self._ensure_flat(x.device)      # before autograd records the parameter tensors
            return _UNetFunction.apply(self, x, *self._construction_order())
        plan = self._forward_impl(x, train=False)
        plan.generation += 1      # whatever an earlier pass left in this workspace is overwritten
        return self._last_pred

    # ------------------------------------------------------------------ whole step without autograd
    def train_step(self, x: torch.Tensor, y: torch.Tensor, optimizer=None) -> torch.Tensor:
        """One iteration of the reference's loop body (train.py:382-393: forward, MSE, zero_grad,
        backward, optimizer step) as direct library calls: no autograd graph, no per-parameter
        bookkeeping -- the same kernels in the same order, so parameters and loss come out bit for
        bit as from ``loss = MSELoss()(model(x), y); optimizer.zero_grad(); loss.backward();
        optimizer.step()``.  For small batches, where that bookkeeping (about 0.5 ms of host time per
        step for 46 tensors) exceeds the GPU time of the step.

        Gradients land in a flat buffer the module keeps; every ``p.grad`` is set to its view of that
        buffer (overwritten by each call), so any optimizer works.  Returns the loss (device scalar)."""
        from ..ops import mse_fwd_bwd
        if not torch.is_tensor(y) or y.device != x.device:
            raise RuntimeError("train_step: target must be a tensor on the input's device")
        plan = self._forward_impl(x, train=True)
        plan.generation += 1
        pred = self._last_pred
        if y.dtype != torch.float32:
            y = y.float()
        out3, dpred = mse_fwd_bwd(pred, y, want_grad=True, gscale=1.0)
        flat = self._flat
        g = self._step_grad
        if g is None or g.shape != flat.shape or g.device != flat.device:
            g = self._step_grad = torch.zeros_like(flat)
            self._step_views = None
        self._backward_impl(plan, dpred, out=g, beta=0.0)
        params = self._construction_order()
        views = self._step_views
        if views is None:
            views = self._step_views = [g[o:o + n].view(shape) for (o, n, shape) in self._flat_layout]
        for p, v in zip(params, views):
            if p.grad is not v:
                p.grad = v
        if optimizer is not None:
            optimizer.step()
        return out3[0]

    # ------------------------------------------------------------------ introspection (tests, debugging)
    def workspace_tensor(self, name: str, plan: Optional[_Plan] = None) -> torch.Tensor:
        """Copy of a named activation ("act.*") or gradient ("grad.*") of the last pass as an
        fp32 NCHW tensor."""
        from ..ops import from_nhwc
        plan = plan or self.last_plan
        info = plan.tensors()[name]
        B, H, W, Cc = info["dims"]
        if info["layout"] == 1:
            n = B * Cc * H * W
            return plan.ws[info["offset"]: info["offset"] + 4 * n].view(torch.float32).view(B, Cc, H, W).clone()
        es = 2 if self.compute_dtype == torch.bfloat16 else 4
        n = B * H * W * info["cstride"]
        t = plan.ws[info["offset"]: info["offset"] + es * n].view(self.compute_dtype).view(B, H, W, info["cstride"])
        return from_nhwc(t, Cc, coff=info["coff"])


def build_unet_from_cfg(cfg) -> UNet:
    """Build the UNet from a config object; same attribute names and defaults as the
    reference (models/unet2d.py:104-122): ``in_channels`` (required), ``out_channels`` = 1,
    ``unet_features`` = (32, 64, 128, 256), ``H_out``/``W_out`` -> out_size or None.
    ``compute_dtype`` ("bf16" | "fp32") is this build's one extra, optional attribute."""
    features = getattr(cfg, "unet_features", (32, 64, 128, 256))
    out_size = (cfg.H_out, cfg.W_out) if getattr(cfg, "H_out", None) is not None else None
    return UNet(in_channels=cfg.in_channels, out_channels=getattr(cfg, "out_channels", 1), features=features,
                out_size=out_size, compute_dtype=getattr(cfg, "compute_dtype", None))
