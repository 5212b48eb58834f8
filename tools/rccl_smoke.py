"""One-rank RCCL sanity check of the collectives bench.py uses for N > 1 (init with device_id, barrier,
all_reduce MAX, all_gather_into_tensor):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/rccl_smoke.py"""
import os
import torch
import torch.distributed as dist

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
dist.barrier()
t = torch.tensor([1.5 + rank], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
rec = torch.arange(256 * 7741, dtype=torch.float32, device="cuda").reshape(256, 7741) + rank
out = torch.empty((world * 256, 7741), dtype=torch.float32, device="cuda")
dist.all_gather_into_tensor(out, rec)
torch.cuda.synchronize()
assert torch.equal(out[:256], rec if rank == 0 else rec - rank)
print("rccl ok: world", world, "max", float(t.item()), "gathered", tuple(out.shape), flush=True)
dist.destroy_process_group()
