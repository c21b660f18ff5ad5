"""RCCL smoke on whatever GPUs this launch has (one rank per GPU): the collectives bench.py / render_cli use."""
import os, time, torch, torch.distributed as dist
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); lr = int(os.environ.get("LOCAL_RANK", "0"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29641")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(lr)
dev = torch.device("cuda", lr)
dist.init_process_group("nccl", device_id=dev)
mine = torch.full((17 * 8, 1920, 3), rank + 1, dtype=torch.uint8, device=dev)
allb = torch.zeros((world * 17 * 8, 1920, 3), dtype=torch.uint8, device=dev)
dist.barrier()
torch.cuda.synchronize()
for it in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        dist.all_gather_into_tensor(allb, mine)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
stats = torch.tensor([1.0, float(rank)], dtype=torch.float64, device=dev)
dist.all_reduce(stats, op=dist.ReduceOp.SUM)
if rank == 0:
    print("RCCL ok: world %d, all_gather_into_tensor of %d B per rank: %.1f us, sum %s, version %s" % (world, mine.numel(), dt * 1e6, stats.tolist(), torch.cuda.nccl.version()))
dist.barrier()
dist.destroy_process_group()
