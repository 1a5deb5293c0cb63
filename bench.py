#!/usr/bin/env python3
"""Throughput of the U-Net training step on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is the loop body of the reference's train_one_epoch (train.py:382-393) on one batch of
synthetic traffic matrices already resident in HBM: forward, MSE loss, zero_grad, backward,
Adam step.  The default workload is the headline configuration: unet2d on GEANT-shaped 23x23
matrices padded to 32x32, 12 past steps as channels -> 1 step, batch 512 per GPU, bf16
compute with fp32 accumulation/master weights.  With N > 1 (one process per GPU, launched by
torch.distributed.run) every rank runs the same per-GPU batch on its own shard (weak
scaling) and gradients are averaged with bucketed RCCL all-reduces overlapped with backward.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (largest share of GPU
time): algorithmic FLOPs per launch / mean launch duration, both taken from HIP events the
library records around every kernel launch on the launch stream during extra instrumented
steps.  `cpu_baseline` is the CPU oracle (the reference's algorithm on stock torch CPU ops)
running the same step on the host cores, rank 0 at N=1 only.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOADS = {
    # name: (in_ch, out_ch, features, H, W, per-GPU batch, train?)
    "cfg1_abilene16_b32": (12, 1, (32, 64, 128, 256), 16, 16, 32, True),
    "cfg2_geant32_b512": (12, 1, (32, 64, 128, 256), 32, 32, 512, True),
    "cfg4_deep128_b64": (24, 12, (32, 64, 128, 256, 512), 128, 128, 64, True),
    "cfg5_infer64": (12, 1, (32, 64, 128, 256), 64, 64, 256, False),
}
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # dense MFMA peaks, MI355X


def synthetic_batch(batch, cin, cout, h, w, seed=1):
    """Synthetic traffic matrices in [0,1) (the reference MinMax-scales its data, dataset.py:326-327):
    X then Y drawn from one seeded CPU generator."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(batch, cin, h, w, generator=g), torch.rand(batch, cout, h, w, generator=g)


def conv_flops_per_sample(cin, cout, feats, h, w):
    """(forward, train) dense FLOPs per sample of UNet.forward (models/unet2d.py:61-102) on sizes that halve
    exactly: 2*Cout*Hout*Wout*Cin*kh*kw per conv with padding taps counted, 2*Hin*Win*Cin*Cout*4 per
    transposed conv; train = forward + weight gradients + input gradients of every layer but the first."""
    fwd, first, ch = 0, None, cin
    dims = [(h >> l, w >> l) for l in range(len(feats) + 1)]
    for l, f in enumerate(feats):                                   # encoder: ch -> f, f -> f at level l
        lh, lw = dims[l]
        fl = 2 * lh * lw * 9 * (ch * f + f * f)
        first = 2 * lh * lw * 9 * ch * f if first is None else first
        fwd, ch = fwd + fl, f
    bh, bw = dims[len(feats)]
    fwd += 2 * bh * bw * 9 * (ch * 2 * ch + 2 * ch * 2 * ch)        # bottleneck
    for l in reversed(range(len(feats))):                          # decoder: convT 2f -> f, conv 2f -> f, f -> f
        f, (lh, lw) = feats[l], dims[l]
        fwd += 2 * (lh // 2) * (lw // 2) * 2 * f * f * 4 + 2 * lh * lw * 9 * (2 * f * f + f * f)
    fwd += 2 * h * w * feats[0] * cout                              # 1x1 head
    return fwd, 3 * fwd - first


def self_launch(args, argv, device_count=None, run=subprocess.run):
    """``python bench.py --gpus N`` with N > 1 and no launcher around it: start N fresh worker processes
    through torch.distributed.run (one per GPU, RCCL rendezvous on 127.0.0.1) BEFORE this process touches
    a GPU, relay their output (rank 0 prints the JSON line) and return their exit status."""
    n = device_count() if device_count is not None else torch.cuda.device_count()   # counting does not initialise the GPU
    if n < args.gpus:
        print(f"bench: --gpus {args.gpus} needs {args.gpus} GPUs, this machine shows {n}", file=sys.stderr)
        return 3
    port = os.environ.get("MASTER_PORT") or str(29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return run(cmd, env=env).returncode


def read_profile(model):
    """[(kernel, label, flops, bytes, ms)] of the last instrumented forward+backward."""
    import unet_amd
    L = unet_amd.lib()
    h = model.last_plan.handle
    out = []
    for i in range(L.mi_unet_profile_count(h)):
        k, lab = C.create_string_buffer(64), C.create_string_buffer(96)
        fl, by, ms = C.c_double(), C.c_double(), C.c_float()
        rc = L.mi_unet_profile_read(h, i, k, 64, lab, 96, C.byref(fl), C.byref(by), C.byref(ms))
        if rc != 0:
            raise RuntimeError("profile_read failed")
        out.append((k.value.decode(), lab.value.decode(), fl.value, by.value, ms.value))
    return out


def pmc_traffic(kernel):
    """HBM bytes per launch of ``kernel`` from the newest committed PMC summary (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes over this same workload, see scripts/pmc_traffic.py), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            k = json.load(f)["kernels"].get(kernel)
        return (k["traffic"], os.path.basename(files[-1])) if k else (None, None)
    except Exception:
        return None, None


def cpu_baseline(cin, cout, feats, H, W, batch, budget_s=12.0):
    from oracle import unet_oracle as O
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    torch.set_num_threads(cores)
    # the workload's own batch and inputs (the same seeded synthetic batch the GPU run uses); the sample is
    # bounded by the number of steps, not by shrinking the batch: 1 warm-up + 10 timed steps where a step
    # is short, fewer (never under 2) where the budget would be exceeded
    b = batch
    state = O.init_state_dict(cin, cout, feats, seed=0)
    X, Y = synthetic_batch(b, cin, cout, H, W)
    t0 = time.perf_counter()
    O.train_steps(state, [(X, Y)], lr=1e-4)            # warm-up step
    warm = time.perf_counter() - t0
    n = max(2, min(10, int(budget_s / max(warm, 1e-3))))
    t0 = time.perf_counter()
    O.train_steps(state, [(X, Y)] * n, lr=1e-4)
    dt = time.perf_counter() - t0
    return {"value": b * n / dt, "unit": "TMs/s", "cores": cores, "kind": "port",
            "sample": f"{n} train steps (fwd+MSE+bwd+Adam, fp32, torch CPU) on the workload's own batch of {b}, "
                      f"after 1 warm-up step; {dt / n * 1e3:.1f} ms/step"}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2_geant32_b512", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--layers-out", default="", help="write per-launch records of the last instrumented step here")
    ap.add_argument("--bucket-mb", type=float, default=8.0, help="gradient all-reduce bucket size (MiB), N > 1 only")
    ap.add_argument("--stock-optim", action="store_true", help="torch.optim.Adam + nn.MSELoss instead of the fused ones")
    ap.add_argument("--fused-step", action="store_true",
                    help="UNet.train_step (the same kernels as direct calls, no autograd bookkeeping) instead of the five-line loop body")
    ap.add_argument("--input-pipeline", action="store_true",
                    help="gather every step's batch from a device-resident series (SeriesWindows) inside the timed region")
    ap.add_argument("--collective", default="allreduce", choices=["allreduce", "rs_ag"],
                    help="gradient exchange per bucket: one all-reduce, or reduce-scatter + all-gather (N > 1 only)")
    args = ap.parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:] if argv is None else argv))

    import unet_amd
    import torch.distributed as dist

    rank, world, local = unet_amd.init_process_group_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"bench: WORLD_SIZE={world} but --gpus {args.gpus}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cin, cout, feats, H, W, batch, train = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    torch.manual_seed(0)
    model = unet_amd.UNet(cin, cout, feats, compute_dtype=args.dtype).to(dev)
    # every rank draws the global batch from the same seed and keeps its own rows
    Xg, Yg = synthetic_batch(batch * world, cin, cout, H, W)
    X = unet_amd.shard_batch(Xg, rank, world).to(dev)
    Y = unet_amd.shard_batch(Yg, rank, world).to(dev)
    del Xg, Yg

    pipe = None
    if args.input_pipeline:
        if cout != 1:
            sys.exit("bench: --input-pipeline needs a one-channel target (the series has one feature)")
        import numpy as np
        rng = np.random.RandomState(1 + rank)
        sw = unet_amd.SeriesWindows(np.abs(np.cumsum(rng.randn(H * W * 2 + 200000)) * 30 + 2000), H, W, H, W, 0, 1, context=cin, device=dev)
        pipe = (sw, sw.to_device_ids(rng.randint(cin - 1, sw.num_windows, size=batch * 64)), [0])

    force_dp = os.environ.get("UNET_TM_FORCE_DP") == "1"      # one-rank rehearsal of the RCCL path
    reducer = unet_amd.GradientReducer(bucket_bytes=int(args.bucket_mb * (1 << 20)), force=force_dp,
                                       mode=args.collective).attach(model) if (world > 1 or force_dp) else None
    if world > 1 or force_dp:
        model.flat_parameters()
        unet_amd.broadcast_parameters(model)
    if args.stock_optim:
        opt, crit = torch.optim.Adam(model.parameters(), lr=1e-4), torch.nn.MSELoss()
    else:
        opt, crit = unet_amd.FusedAdam(model.parameters(), lr=1e-4, model=model), unet_amd.MSELoss()
    loss_acc = torch.zeros((), device=dev)

    def step():
        if pipe is not None:                   # this step's batch: one gather kernel, no host copy
            sw, ids, cursor = pipe
            lo = cursor[0] % 64 * batch
            cursor[0] += 1
            sw.batch(ids[lo:lo + batch], out=(X, Y))
        if train and args.fused_step:
            loss_acc.add_(model.train_step(X, Y, opt))
        elif train:
            pred = model(X)
            loss = crit(pred, Y)
            opt.zero_grad()
            loss.backward()
            opt.step()
            loss_acc.add_(loss.detach())       # train.py:393 without the per-step host sync
        else:
            with torch.no_grad():
                model(X)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    model.train(train)
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_ms = (time.perf_counter() - t0) / args.steps * 1e3    # host time to enqueue a step
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms_per_step = dt / args.steps * 1e3
    value = batch * world * args.steps / dt

    # ---- instrumented steps: HIP events around every kernel launch (same stream)
    L = unet_amd.lib()
    per_kernel = {}
    gpu_ms = 0.0
    pair_ms = 0.0
    if args.profile_steps > 0:
        # An event pair brackets a launch AND the markers' own turnaround.  A pair around nothing,
        # recorded behind queued work like the real ones, reads two turnarounds back to back; around
        # a launch one of them hides behind the kernel, so half the empty pair is taken off every
        # bracketed launch.  The profiler's kernel durations are the check: dconv, hwgrad and the
        # combine agree with them to 3 % after the correction, and read 12 % long without it.
        step()
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
        for e0, e1 in pairs:
            e0.record(); e1.record()
        torch.cuda.synchronize()
        pair_ms = 0.5 * sorted(e0.elapsed_time(e1) for e0, e1 in pairs)[len(pairs) // 2]
        for _ in range(args.profile_steps):
            L.mi_unet_profile_enable(model.last_plan.handle, 1)
            step()
            torch.cuda.synchronize()
            records = read_profile(model)
            records = [(k, lab, fl, by, max(ms - pair_ms, 0.1 * ms), ms) for k, lab, fl, by, ms in records]
            for k, lab, fl, by, ms, raw in records:
                e = per_kernel.setdefault(k, dict(launches=0, ms=0.0, raw_ms=0.0, flops=0.0, bytes=0.0))
                e["launches"] += 1; e["ms"] += ms; e["raw_ms"] += raw; e["flops"] += fl; e["bytes"] += by
                gpu_ms += ms
            L.mi_unet_profile_enable(model.last_plan.handle, 0)
        gpu_ms /= args.profile_steps
        if args.layers_out and rank == 0:
            with open(args.layers_out, "w") as f:
                for k, lab, fl, by, ms, _raw in records:
                    f.write(f"{lab:28s} {k:16s} {ms * 1e3:9.1f} us  {fl / (ms * 1e-3) / 1e12 if ms > 0 else 0:8.1f} TF/s  "
                            f"{by / (ms * 1e-3) / 1e9 if ms > 0 else 0:8.1f} GB/s\n")
    fence()

    if rank == 0:
        fwd_fl, train_fl = conv_flops_per_sample(cin, cout, feats, H, W)
        flops_per_sample = train_fl if train else fwd_fl
        roof = None
        if per_kernel:
            dom = max(per_kernel, key=lambda k: per_kernel[k]["ms"])
            e = per_kernel[dom]
            avg_ms = e["ms"] / e["launches"]
            if e["flops"] > 0:
                ach = e["flops"] / e["launches"] / (avg_ms * 1e-3) / 1e12
                peak = PEAK_TFLOPS[args.dtype]
                traffic, traffic_src = (pmc_traffic(dom) if args.workload == "cfg2_geant32_b512" and args.dtype == "bf16"
                                        and batch == 512 else (None, None))
                roof = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "launches_per_step": e["launches"] // args.profile_steps, "avg_launch_ms": round(avg_ms, 5),
                        "share_of_gpu_time": round(e["ms"] / args.profile_steps / gpu_ms, 3),
                        # as measured, before the marker-turnaround correction (see event_overhead_us_removed_per_launch)
                        "avg_launch_ms_uncorrected": round(e["raw_ms"] / e["launches"], 5),
                        "achieved_uncorrected": round(e["flops"] / (e["raw_ms"] * 1e-3) / 1e12, 2),
                        "frac_uncorrected": round(e["flops"] / (e["raw_ms"] * 1e-3) / 1e12 / peak, 4)}
            else:
                ach = e["bytes"] / e["launches"] / (avg_ms * 1e-3) / 1e9
                roof = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(ach / 8000.0, 4), "traffic": None,
                        "launches_per_step": e["launches"] // args.profile_steps, "avg_launch_ms": round(avg_ms, 5),
                        "share_of_gpu_time": round(e["ms"] / args.profile_steps / gpu_ms, 3)}
        out = {
            "metric": "U-Net train TMs/sec (fwd+bwd)" if train else "U-Net inference TMs/sec",
            "value": round(value, 1), "unit": "TMs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "host_enqueue_ms_per_step": round(host_ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic series, batch gathered on the device every step" if pipe is not None else "synthetic",
            "config": {"workload": args.workload, "in_channels": cin, "out_channels": cout, "features": list(feats),
                       "height": H, "width": W, "per_gpu_batch": batch, "global_batch": batch * world,
                       "step": "fwd+MSE+zero_grad+bwd+Adam" if train else "fwd (no_grad)",
                       "optimizer": "torch.optim.Adam" if args.stock_optim else "FusedAdam (HIP)",
                       "parallelism": f"dp{world}"},
            "rccl_world_size": dist.get_world_size() if dist.is_initialized() else 1,
            "dist_backend": dist.get_backend() if dist.is_initialized() else None,
            "collective": (reducer.mode if reducer is not None else None),
            "conv_mfma_frac_whole_step": round(flops_per_sample * value / world / (PEAK_TFLOPS[args.dtype] * 1e12), 4),
            "gpu_kernel_ms_per_step": round(gpu_ms, 4),
            "event_overhead_us_removed_per_launch": round(pair_ms * 1e3, 2),
            "roofline": roof,
            "kernels": {k: {"launches_per_step": v["launches"] // max(args.profile_steps, 1),
                            "ms_per_step": round(v["ms"] / max(args.profile_steps, 1), 4),
                            "ms_per_step_uncorrected": round(v["raw_ms"] / max(args.profile_steps, 1), 4),
                            "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] > 0 and v["flops"] > 0 else None}
                        for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["ms"])},
        }
        if world == 1 and not args.no_cpu_baseline and train:
            out["cpu_baseline"] = cpu_baseline(cin, cout, feats, H, W, batch)
        elif world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    finally:
        import torch.distributed as _dist
        if _dist.is_available() and _dist.is_initialized():
            _dist.destroy_process_group()
