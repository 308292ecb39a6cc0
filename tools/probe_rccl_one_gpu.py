"""Can two ranks share one GPU under RCCL on this pool?  (tools/ probe; decides whether the RCCL path of kbbq_amd/dist.py
can be rehearsed on a one-GPU box.)  Run: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/probe_rccl_one_gpu.py"""
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world)
    x = torch.full((1 << 20,), rank + 1, dtype=torch.uint8, device="cuda")
    out = torch.empty_like(x)
    dist.all_to_all_single(out, x)
    torch.cuda.synchronize()
    g = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(g, x)
    torch.cuda.synchronize()
    print("rank", rank, "ok", int(out[0]), int(out[-1]), [int(t[0]) for t in g], flush=True)
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print("rank", rank, "FAILED:", repr(e)[:400], flush=True)
    sys.exit(0)
