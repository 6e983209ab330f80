"""RCCL on this box with a one-rank group: the collectives bench.py and sharded.py issue, with their dtypes and shapes."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier()
el = torch.tensor([1.25], device=dev, dtype=torch.float64); dist.all_reduce(el, op=dist.ReduceOp.MAX)
packed = torch.arange(10000 * 10, dtype=torch.int32, device=dev).view(10000, 10)
out = torch.empty((1 * 10000, 10), dtype=torch.int32, device=dev)
dist.all_gather_into_tensor(out, packed)
torch.cuda.synchronize()
assert torch.equal(out, packed) and float(el) == 1.25
dist.barrier(); dist.destroy_process_group()
print("rccl one-rank collectives ok")
