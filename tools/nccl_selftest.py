"""Developer check: the dist.py helpers on a 1-rank RCCL group (API / device-placement errors show up without 8 GPUs)."""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd import dist as gd

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
# the helpers short-circuit at world 1, so call the collectives they wrap directly with the same arguments
buf = torch.arange(8, dtype=torch.float32, device=gd._comm_device(dev)); dist.broadcast(buf, src=0)
dist.barrier(device_ids=[torch.cuda.current_device()])
t = torch.tensor([3.5], dtype=torch.float64, device=gd._comm_device(dev)); dist.all_reduce(t, op=dist.ReduceOp.MAX)
rec = torch.ones((4, gd.RECORD_F64), dtype=torch.float64, device=dev)
parts = [torch.empty_like(rec)]; dist.all_gather(parts, rec)
torch.cuda.synchronize()
print("rccl 1-rank ok", float(t.item()), parts[0].sum().item(), dist.get_backend())
dist.destroy_process_group()
