"""Data-parallel training: one process per GPU, gradients averaged across ranks.

The reference is single-device (train.py:420-423, :495); this is the one collective the
path needs.  Samples are independent (the network has no cross-sample statistics), so
rank r simply takes rows [r*B/W, (r+1)*B/W) of the global batch and the per-rank MSE
means average to the global-batch mean: gradients are SUMMED across ranks after each
rank's are scaled by 1/world_size (folded into the weight-gradient kernels' alpha).

Bucketing.  Backward runs in segments whose flat gradient ranges are contiguous and
descend through one flat fp32 buffer (parameters are laid out in forward order), so a
bucket is a run of consecutive segments = one contiguous slice.  As soon as a bucket's
kernels are enqueued its collective is issued as a STREAM-ORDERED operation
(``async_op=False`` on device tensors) on the plan's weight-gradient stream, which has
caught up with the bucket's kernels: the caller's stream goes on with the next bucket's
dgrad chain and only later weight gradients queue up behind the exchange; the optimizer
runs after the last bucket's call has joined the two streams.  (Host tensors -- the gloo
tests -- keep asynchronous handles, waited for in ``wait_all``.)  Over xGMI
(point-to-point links) a ring all-reduce is bound by one link, so buckets are few and
large (default 8 MiB) and the tiny early-encoder tail rides in the last bucket.

Collective per bucket (``mode``): ``"allreduce"`` issues one all-reduce and leaves the
algorithm to RCCL; ``"rs_ag"`` issues a reduce-scatter followed by an all-gather of the
bucket (each rank owns 1/world of it in between), the direct form that can drive all
point-to-point links of a fully connected node at once.  The two give every rank the same
sums; which is faster on a given node is a measurement (``bench.py --collective``).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


_DEBUG = os.environ.get("UNET_TM_DEBUG_DP", "0") == "1"


def init_process_group_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*.
    ``backend`` defaults to "nccl" (= RCCL on ROCm) when a GPU is visible, else "gloo".
    Returns (rank, world_size, local_rank).

    ``UNET_TM_DIST_BACKEND`` selects another backend and ``UNET_TM_FORCE_DEVICE=<index>`` another
    device.  Several gloo ranks sharing ONE GPU is not a supported rehearsal: ProcessGroupGloo stages
    device tensors through host memory, and with two processes on one device its asynchronous
    all-reduce stalls intermittently inside ``Work.wait()`` even in a stock-torch program with no
    code of this package in it (RCCL itself refuses two ranks per GPU).  Multi-rank logic is covered
    by gloo on CPU tensors, the RCCL calls by a one-rank group on the GPU."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "UNET_TM_FORCE_DEVICE" in os.environ:
        local = int(os.environ["UNET_TM_FORCE_DEVICE"])
    backend = backend or os.environ.get("UNET_TM_DIST_BACKEND") or None
    if (world > 1 or os.environ.get("UNET_TM_FORCE_DP") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(X: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rows of the global batch owned by ``rank`` (even split; the global batch must divide)."""
    if X.shape[0] % world != 0:
        raise ValueError(f"global batch {X.shape[0]} is not divisible by world size {world}")
    per = X.shape[0] // world
    return X[rank * per:(rank + 1) * per]


def plan_buckets(segments: List[Tuple[int, int]], bucket_bytes: int) -> List[Tuple[int, int, int, int]]:
    """Group consecutive backward segments into buckets of at least ``bucket_bytes``.

    ``segments[s] = (lo, hi)`` are flat element ranges in execution order (descending).
    Returns [(seg_lo, seg_hi, lo, hi)]: segments [seg_lo, seg_hi) form one bucket covering
    flat elements [lo, hi)."""
    buckets = []
    start = 0
    while start < len(segments):
        end = start + 1
        while end < len(segments) and (segments[start][1] - segments[end - 1][0]) * 4 < bucket_bytes:
            end += 1
        lo = min(s[0] for s in segments[start:end])
        hi = max(s[1] for s in segments[start:end])
        buckets.append((start, end, lo, hi))
        start = end
    # a small trailing bucket is merged into its predecessor (one fewer exposed collective)
    if len(buckets) >= 2 and (buckets[-1][3] - buckets[-1][2]) * 4 < bucket_bytes // 4:
        a, b = buckets[-2], buckets[-1]
        buckets[-2:] = [(a[0], b[1], min(a[2], b[2]), max(a[3], b[3]))]
    return buckets


class GradientReducer:
    """Averages a flat gradient buffer across the process group, bucket by bucket."""

    MODES = ("allreduce", "rs_ag")

    def __init__(self, process_group=None, bucket_bytes: int = 8 << 20, force: bool = False, mode: Optional[str] = None):
        """``force`` keeps the bucketed path (and its collectives) active in a one-rank group:
        a way to exercise the RCCL calls on a single GPU.  ``mode`` picks the collective per bucket
        (default "allreduce", or the UNET_TM_COLLECTIVE environment variable)."""
        mode = mode or os.environ.get("UNET_TM_COLLECTIVE") or "allreduce"
        if mode not in self.MODES:
            raise ValueError(f"GradientReducer: unknown mode {mode!r} (expected one of {self.MODES})")
        self.mode = mode
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force = force and dist.is_initialized()
        self.bucket_bytes = bucket_bytes
        self.grad_prescale = 1.0 / self.world
        self._pending = []
        self._bucket_cache = {}

    def attach(self, model) -> "GradientReducer":
        model._dp = self if (self.world > 1 or self.force) else None
        return self

    def buckets_for(self, plan) -> List[Tuple[int, int, int, int]]:
        key = id(plan)
        b = self._bucket_cache.get(key)
        if b is None:
            b = plan_buckets(plan.segments, self.bucket_bytes)
            self._bucket_cache[key] = b
        return b

    def reduce_async(self, flat_slice: torch.Tensor) -> None:
        """Start summing ``flat_slice`` (a contiguous fp32 range of the flat gradient buffer) across
        the group, in place; ``wait_all`` completes it.

        Device tensors: the collective is issued as a stream-ordered operation on the stream that is
        current at the call (``async_op=False``: that stream, and only that stream, continues behind
        the collective) -- in the bucketed backward that is the weight-gradient stream, so the
        caller's stream runs the next bucket's dgrad chain meanwhile and only the weight gradients
        of later buckets queue up behind the exchange.  Measured in a one-rank RCCL group, three
        buckets: 1.26 ms per step against 1.25 without collectives; the process group's
        ``async_op=True`` handles waited for at the end cost 1.39 ms (the same with 4-byte
        tensors), and a communication stream of the reducer's own 1.42 ms: every further stream
        shares one of the process's few hardware queues with the compute streams.
        Host tensors (gloo) keep the handle form."""
        if not (self.world > 1 or self.force):
            return
        if _DEBUG:
            print(f"[dp rank {dist.get_rank(self.group)}] {self.mode} {flat_slice.numel()} elements", flush=True)
        self._collective(flat_slice, async_op=not flat_slice.is_cuda)

    def _collective(self, flat_slice: torch.Tensor, async_op: bool) -> None:
        def keep(work):
            if async_op:
                self._pending.append(work)
        if self.mode == "allreduce":
            keep(dist.all_reduce(flat_slice, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op))
            return
        # reduce-scatter + all-gather over the part of the bucket that splits evenly into `world`
        # shards of whole 8-element (32-byte) groups; the few elements left over (segment ranges are
        # multiples of 64 elements, so there are none for world sizes dividing 8) ride in a small
        # all-reduce.  RCCL runs the collectives of one group in issue order on the group's stream, so
        # the all-gather is simply enqueued behind the reduce-scatter; a host backend (gloo) runs
        # asynchronous operations on worker threads in no particular order, so there the
        # reduce-scatter is waited for first.
        n = flat_slice.numel()
        rank = dist.get_rank(self.group)
        shard = n // (8 * self.world) * 8
        body = shard * self.world
        if shard > 0:
            mine = flat_slice[rank * shard:(rank + 1) * shard]
            rs = dist.reduce_scatter_tensor(mine, flat_slice[:body], op=dist.ReduceOp.SUM, group=self.group,
                                            async_op=async_op)
            if async_op:
                rs.wait()
            keep(dist.all_gather_into_tensor(flat_slice[:body], mine, group=self.group, async_op=async_op))
        if body < n:
            keep(dist.all_reduce(flat_slice[body:], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op))

    def wait_all(self) -> None:
        for i, w in enumerate(self._pending):
            w.wait()
            if _DEBUG:
                print(f"[dp rank {dist.get_rank(self.group)}] collective {i} done", flush=True)
        self._pending.clear()

    def reduce_flat(self, flat: torch.Tensor, segments: List[Tuple[int, int]], prescaled: bool = False) -> None:
        """Synchronous form used by tests and by callers that computed gradients elsewhere:
        average ``flat`` in place, bucket by bucket."""
        if self.world == 1:
            return
        if not prescaled:
            flat.mul_(self.grad_prescale)
        for (_, _, lo, hi) in plan_buckets(segments, self.bucket_bytes):
            self.reduce_async(flat[lo:hi])
        self.wait_all()


def broadcast_parameters(model, src: int = 0, group=None) -> None:
    """Make every rank start from rank ``src``'s parameters (one broadcast of the flat buffer
    when the model has one, else per tensor)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    flat = getattr(model, "_flat", None)
    if flat is not None:
        dist.broadcast(flat, src=src, group=group)
        model.mark_parameters_changed()
    else:
        for p in model.parameters():
            dist.broadcast(p.data, src=src, group=group)
