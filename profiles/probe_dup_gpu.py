"""Probe (run under torch.distributed.run --nproc-per-node 2): can two RCCL ranks share ONE device on this pool?"""
import os
import sys
import torch
import torch.distributed as dist

backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo")
    x = torch.full((4,), float(rank + 1), device="cuda:0", dtype=torch.float64)
    out = torch.empty((8,), device="cuda:0", dtype=torch.float64)
    dist.all_gather_into_tensor(out, x)
    torch.cuda.synchronize()
    print(f"rank {rank} {backend} all_gather_into_tensor on a shared device OK: {out.tolist()}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print(f"rank {rank} {backend} FAILED: {type(e).__name__}: {str(e)[:300]}", flush=True)
    sys.exit(3)
