"""``FusedAdam``: ``torch.optim.Adam`` semantics (train.py:497: default betas/eps, no weight
decay, no amsgrad) as one HIP kernel launch over the model's flat parameter buffer.  Given the
model (``model=``), the same launch also writes the packed MFMA weight images the next forward and
backward read, from the updated values (``mi_unet_adam_pack``): the parameters are read once per
step instead of three times.

State lives where ``torch.optim.Adam`` keeps it -- ``self.state[p]`` with the keys ``step``,
``exp_avg`` and ``exp_avg_sq`` -- so ``state_dict()`` / ``load_state_dict()`` round-trip and
interchange with the stock optimizer.  On the flat path the two moment tensors of every
parameter are views of one flat buffer each."""
from __future__ import annotations

import os

import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, model=None):
        """``model`` is optional: every step bumps the version counter of the parameters it wrote,
        which is what the U-Net module watches to refresh its packed weights.  With the model, a
        step over its whole flat buffer also packs the weights into the workspace of the model's
        last training forward (bit-identical parameters and images; ``fuse_pack = False`` turns
        that off)."""
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.grad_scale = grad_scale
        self._model = model
        self.fuse_pack = os.environ.get("UNET_TM_FUSE_PACK", "1")[:1] != "0"     # A/B switch for measurements
        self._flat_state = {}      # per group: step count, flat moment buffers, cached layout

    @staticmethod
    def _flat_span(params, attr):
        """(start pointer, end pointer) if the tensors tile one ascending, 256-byte-granular
        flat fp32 range (the layout UNet gives its parameters and gradients), else None."""
        ts = [getattr(p, attr) if attr else p for p in params]
        if any(t is None or t.dtype != torch.float32 or not t.is_contiguous() for t in ts):
            return None
        start = ts[0].data_ptr()
        cur = start
        for t in ts:
            if t.data_ptr() != cur:
                return None
            cur += (t.numel() + 63) // 64 * 64 * 4
        return start, cur

    @staticmethod
    def _storage_as_flat(t: torch.Tensor) -> torch.Tensor:
        """fp32 1-D tensor over the whole storage ``t`` lives in (the flat buffer it is a view of)."""
        st = t.untyped_storage()
        return torch.empty(0, dtype=torch.float32, device=t.device).set_(st, 0, (st.nbytes() // 4,))

    def _written(self, params) -> None:
        """The kernel wrote the parameters through raw pointers: bump their version counters, as an
        in-place torch op would have (the U-Net module re-packs its MFMA weight images when the
        versions move; autograd's saved-tensor checks see the write too)."""
        torch._C._increment_version(params)
        if self._model is not None:
            self._model.mark_parameters_changed()

    def _flat_adam(self, params, flat_p, flat_g, m, v, group, step) -> None:
        """One launch over the flat range; fused with the weight packing when the range is the
        model's whole flat buffer and its last training plan can take it."""
        b1, b2 = group["betas"]
        model = self._model
        plan = getattr(model, "last_train_plan", None) if (model is not None and self.fuse_pack) else None
        if (plan is not None and plan.adam_pack_ok and plan.ws_ready and plan.leased is None
                and model._flat is not None and model._flat.data_ptr() == flat_p.data_ptr()
                and model._flat.numel() == flat_p.numel() and plan.ws.device == flat_p.device):
            from ._lib import check, lib
            from .models.unet2d import _on_device
            with _on_device(flat_p.device):
                check(lib().mi_unet_adam_pack(plan.handle, flat_p.data_ptr(), flat_g.data_ptr(), m.data_ptr(), v.data_ptr(),
                                              plan.ws.data_ptr(), plan.ws_bytes, group["lr"], b1, b2, group["eps"], step,
                                              self.grad_scale, torch.cuda.current_stream(flat_p.device).cuda_stream),
                      "unet_adam_pack")
            self._written(params)          # versions move, every plan's images are marked stale ...
            plan.packed_version, plan.packed_train = model._param_version(), True      # ... except the one just written
            return
        ops.adam_step(flat_p, flat_g, m, v, group["lr"], b1, b2, group["eps"], step, self.grad_scale)
        self._written(params)

    def _group_step(self, gi, group) -> dict:
        """The group's running record; the step count is taken from ``self.state`` the first time
        (a loaded checkpoint continues its bias correction)."""
        st = self._flat_state.get(gi)
        if st is None:
            step = 0
            for p in group["params"]:
                ps = self.state.get(p)
                if ps and "step" in ps:
                    step = max(step, int(float(ps["step"])))
            st = self._flat_state[gi] = {"step": step}
        return st

    def _step_cached(self, gi, group) -> bool:
        """The usual step: same parameter tensors at the same addresses as last time, every gradient
        an fp32 view at its parameter's offset inside one flat gradient buffer.  Confirms exactly
        that (one pointer per tensor, ~20 us of host time instead of ~150 for the full analysis)
        and launches; returns False to send anything else down the general path."""
        st = self._flat_state.get(gi)
        layout = st.get("layout") if st else None
        if layout is None:
            return False
        ids, params, pstart, offsets, flat_p = layout
        gp = group["params"]
        if len(gp) != len(ids) or params[0].data_ptr() != pstart:
            st.pop("layout", None)
            return False
        for p, i in zip(gp, ids):
            if id(p) != i:
                st.pop("layout", None)
                return False
        g0 = params[0].grad
        if g0 is None:
            return False
        gstart = g0.data_ptr()
        f32 = torch.float32
        for p, off in zip(params, offsets):
            g = p.grad
            if g is None or g.dtype is not f32 or g.data_ptr() != gstart + off or p.data_ptr() != pstart + off:
                return False
        n = flat_p.numel()
        flat_g = self._storage_as_flat(g0)
        go = (gstart - flat_g.data_ptr()) // 4
        if go + n > flat_g.numel():
            return False
        st["step"] += 1
        self._flat_adam(params, flat_p, flat_g[go:go + n], st["m"], st["v"], group, st["step"])
        return True

    def _adopt_flat_moments(self, st, params, pspan, n, device) -> None:
        """Flat moment buffers for the range; whatever ``self.state`` already holds for a parameter
        (a loaded checkpoint, steps taken tensor by tensor) is carried into its slice, and from
        here on ``self.state[p]`` holds views of the flat buffers."""
        m = torch.zeros(n, device=device, dtype=torch.float32)
        v = torch.zeros(n, device=device, dtype=torch.float32)
        for p in params:
            o = (p.data_ptr() - pspan[0]) // 4
            mv, vv = m[o:o + p.numel()].view(p.shape), v[o:o + p.numel()].view(p.shape)
            ps = self.state[p]
            if "exp_avg" in ps:
                mv.copy_(ps["exp_avg"])
                vv.copy_(ps["exp_avg_sq"])
            ps["exp_avg"], ps["exp_avg_sq"] = mv, vv
        st["m"], st["v"] = m, v

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            if self._step_cached(gi, group):
                continue
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            # model.parameters() yields state_dict order (ups before bottleneck); the flat
            # buffer is laid out in construction order, so walk the tensors by address
            params.sort(key=lambda p: p.data_ptr())
            b1, b2 = group["betas"]
            pspan = self._flat_span(params, None)
            gspan = self._flat_span(params, "grad")
            flat_ok = pspan is not None and gspan is not None and len(params) == len(group["params"])
            st = self._group_step(gi, group)
            st["step"] += 1
            if flat_ok:
                n = (pspan[1] - pspan[0]) // 4
                flat_p = self._storage_as_flat(params[0])
                flat_g = self._storage_as_flat(params[0].grad)
                po = (pspan[0] - flat_p.data_ptr()) // 4
                go = (gspan[0] - flat_g.data_ptr()) // 4
                if "m" not in st or st["m"].numel() != n or st.get("m_base") != pspan[0]:
                    self._adopt_flat_moments(st, params, pspan, n, flat_p.device)
                    st["m_base"] = pspan[0]
                self._flat_adam(params, flat_p[po:po + n], flat_g[go:go + n], st["m"], st["v"], group, st["step"])
                # remember the layout: later steps only confirm that nothing moved (see _step_cached)
                st["layout"] = (tuple(id(p) for p in group["params"]), params, pspan[0],
                                [p.data_ptr() - pspan[0] for p in params], flat_p[po:po + n])
            else:
                # tensor by tensor, on the same moments (views of the flat buffers if the flat path ran before)
                st.pop("layout", None)
                for p in params:
                    ps = self.state[p]
                    if "exp_avg" not in ps:
                        ps["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                        ps["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    g = p.grad.contiguous()
                    if not p.is_contiguous():
                        raise RuntimeError("FusedAdam needs contiguous parameters")
                    ops.adam_step(p, g, ps["exp_avg"], ps["exp_avg_sq"], group["lr"], b1, b2, group["eps"], st["step"],
                                  self.grad_scale)
                self._written(params)
        return loss

    # ------------------------------------------------------------------ checkpoints
    def state_dict(self):
        """torch.optim.Adam's format: per parameter ``step`` (a float tensor), ``exp_avg``,
        ``exp_avg_sq``."""
        for gi, group in enumerate(self.param_groups):
            st = self._flat_state.get(gi)
            if not st:
                continue
            for p in group["params"]:
                ps = self.state.get(p)
                if ps and "exp_avg" in ps:
                    ps["step"] = torch.tensor(float(st["step"]))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # the loaded moments are fresh tensors: the next step re-adopts them into flat buffers and
        # continues from the loaded step count
        self._flat_state = {}
