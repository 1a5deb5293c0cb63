"""Two gloo ranks on one GPU, no library code: asynchronous all-reduce of a device tensor between
batches of stock kernels, the pattern scripts/rehearse_2rank.sh exercises through the bucketed backward."""
import os, sys, torch, torch.distributed as dist
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)
torch.cuda.set_device(0)
a = torch.randn(2048, 2048, device="cuda")
g = torch.randn(3_000_000, device="cuda")
for it in range(200):
    for _ in range(20):
        b = a @ a
    w = dist.all_reduce(g, async_op=True)
    for _ in range(20):
        b = a @ a
    w.wait()
    g.mul_(0.5)
    if it % 50 == 0:
        print(f"rank {rank} iteration {it}", flush=True)
torch.cuda.synchronize()
print(f"rank {rank} done", flush=True)
dist.destroy_process_group()
