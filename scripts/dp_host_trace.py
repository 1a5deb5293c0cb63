"""Host-side cost of the data-parallel training step on ONE rank (RCCL group of one, the bucketed
path forced on): time spent by the host in each phase of a step and inside every collective call,
with the GPU left to run behind (no synchronisation inside the timed loop).

    UNET_TM_FORCE_DP=1 python scripts/dp_host_trace.py [--bucket-mb 8] [--steps 40]
"""
import argparse
import os
import sys
import time
from collections import defaultdict

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import unet_amd  # noqa: E402
from bench import synthetic_batch, WORKLOADS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bucket-mb", type=float, default=8.0)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--collective", default="allreduce")
    ap.add_argument("--no-collective", action="store_true", help="keep the bucketed backward calls, skip the collectives")
    ap.add_argument("--tiny", action="store_true", help="all-reduce a 4-byte tensor per bucket instead of the bucket")
    ap.add_argument("--sync-op", action="store_true", help="async_op=False")
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device("cuda", 0)
    cin, cout, feats, H, W, batch, _ = WORKLOADS["cfg2_geant32_b512"]
    torch.manual_seed(0)
    model = unet_amd.UNet(cin, cout, feats, compute_dtype="bf16").to(dev)
    X, Y = (t.to(dev) for t in synthetic_batch(batch, cin, cout, H, W))
    red = unet_amd.GradientReducer(bucket_bytes=int(a.bucket_mb * (1 << 20)), force=True, mode=a.collective).attach(model)
    model.flat_parameters()
    opt, crit = unet_amd.FusedAdam(model.parameters(), lr=1e-4, model=model), unet_amd.MSELoss()
    acc = defaultdict(float)
    cnt = defaultdict(int)

    def timed(name, fn):
        def w(*args, **kw):
            t = time.perf_counter()
            r = fn(*args, **kw)
            acc[name] += time.perf_counter() - t; cnt[name] += 1
            return r
        return w
    if a.no_collective:
        red.reduce_async = lambda t: None
    if a.tiny:
        dummy = torch.zeros(1, device=dev)
        red.reduce_async = lambda t: red._pending.append(dist.all_reduce(dummy, async_op=True))
    if a.sync_op:
        red.reduce_async = lambda t: dist.all_reduce(t)
    L = unet_amd.lib()
    for fn in ("mi_unet_backward", "mi_unet_forward", "mi_adam_step", "mi_mse_fwd_bwd"):
        if hasattr(L, fn):
            setattr(L, fn, timed("C " + fn, getattr(L, fn)))
    red.reduce_async = timed("reduce_async", red.reduce_async)
    red.wait_all = timed("wait_all", red.wait_all)

    def step(record):
        t0 = time.perf_counter(); pred = model(X)
        t1 = time.perf_counter(); loss = crit(pred, Y); opt.zero_grad()
        t2 = time.perf_counter(); loss.backward()
        t3 = time.perf_counter(); opt.step()
        t4 = time.perf_counter()
        if record:
            for k, v in (("forward", t1 - t0), ("loss+zero_grad", t2 - t1), ("backward", t3 - t2), ("optimizer", t4 - t3)):
                acc[k] += v; cnt[k] += 1
    for _ in range(10):
        step(False)
    torch.cuda.synchronize()
    acc.clear(); cnt.clear()
    t = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    host = time.perf_counter() - t
    torch.cuda.synchronize()
    total = time.perf_counter() - t
    print(f"buckets of {a.bucket_mb} MiB, {a.collective}: {total / a.steps * 1e3:.4f} ms/step, host {host / a.steps * 1e3:.4f} ms/step")
    for k in ("forward", "C mi_unet_forward", "loss+zero_grad", "C mi_mse_fwd_bwd", "backward", "C mi_unet_backward", "reduce_async",
              "wait_all", "optimizer", "C mi_adam_step"):
        if cnt[k]:
            print(f"  {k:16s} {acc[k] / a.steps * 1e6:8.1f} us/step in {cnt[k] / a.steps:.1f} calls")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
