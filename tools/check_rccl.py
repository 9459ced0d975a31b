"""One-rank check that torch.distributed's "nccl" backend (RCCL) initialises on this box and that the collectives
bench.py uses (all_reduce SUM / MAX, barrier) run: `python tools/check_rccl.py`."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dev=torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x=torch.ones(2,80,device=dev); dist.all_reduce(x); dist.barrier()
el=torch.tensor([1.5],dtype=torch.float64,device=dev); dist.all_reduce(el, op=dist.ReduceOp.MAX)
print("nccl ok", float(x.sum()), float(el)); dist.destroy_process_group()
