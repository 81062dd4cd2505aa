"""Debug: RCCL single-rank probe (process group init with device_id, barrier, all_gather_into_tensor, all_reduce) and the
bench under torchrun with one rank -- the parts of the multi-GPU path a one-GPU box can exercise."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(8, device=dev, dtype=torch.float32)
out = torch.empty(8, device=dev)
dist.all_gather_into_tensor(out, x)
dist.all_reduce(x)
dist.barrier()
torch.cuda.synchronize()
print("rccl ok", out.tolist()[:3], float(x.sum()))
dist.destroy_process_group()
