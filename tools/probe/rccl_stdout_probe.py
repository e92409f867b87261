"""When does librccl write its "Librccl path : ..." line to stdout?  (bench.py's contract line must be the LAST stdout line.)"""
import ctypes
import datetime
import os
import sys

import torch
import torch.distributed as dist

libc = ctypes.CDLL(None)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
print("MARK0 before init", flush=True)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=60))
libc.fflush(None)
print("MARK1 after init", flush=True)
t = torch.ones(4, device=dev)
dist.all_reduce(t)
torch.cuda.synchronize()
libc.fflush(None)
print("MARK2 after first collective", flush=True)
dist.destroy_process_group()
libc.fflush(None)
print("MARK3 after destroy", flush=True)
