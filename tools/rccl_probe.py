"""Probe: can two ranks form an RCCL communicator through nvllm_ctx_create here?  (On a 1-GPU box both ranks
share device 0, which RCCL normally rejects; the point is to see bootstrap behaviour and the error path.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist

import nano_vllm_candle_amd as pkg

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
box = [pkg.Context.make_rccl_id() if rank == 0 else None]
dist.broadcast_object_list(box, src=0)
print(rank, "id ok", flush=True)
t = time.time()
dev = int(os.environ.get("PROBE_DEVICE", os.environ.get("LOCAL_RANK", 0)))
try:
    ctx = pkg.Context(dev, tp_rank=rank, tp_size=world, rccl_id=box[0])
    print(rank, "comm ok", time.time() - t, flush=True)
    import numpy as np
    d = ctx.to_device(np.full(8, rank + 1.0, np.float32))
    pkg._lib.check(pkg._lib.lib().nvllm_op_allreduce(ctx.h, d.ptr, 8), ctx.h)
    print(rank, "allreduce", d.numpy(), flush=True)
except Exception as e:
    print(rank, "ERR", repr(e)[:300], time.time() - t, flush=True)
dist.barrier()
