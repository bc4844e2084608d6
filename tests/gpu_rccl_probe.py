"""RCCL sanity on one GPU (not a test): a one-rank "nccl" process group, the collectives dist.py issues on device tensors."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
mine = torch.tensor([1, 2, 3], dtype=torch.int64, device=dev)
allv = [torch.zeros(3, dtype=torch.int64, device=dev)]
dist.all_gather(allv, mine)
flag = torch.tensor([1], dtype=torch.int64, device=dev); dist.all_reduce(flag, op=dist.ReduceOp.MAX)
el = torch.tensor([0.5], dtype=torch.float64, device=dev); dist.all_reduce(el, op=dist.ReduceOp.MAX)
dist.barrier()
print("rccl ok", allv[0].tolist(), int(flag.item()), float(el.item()), dist.get_backend())
dist.destroy_process_group()
