#!/usr/bin/env python3
"""RCCL smoke on whatever GPUs are visible (run under torch.distributed.run, or alone = world size 1): the path's one
collective, all_gather_into_tensor on bf16 pseudo-token rows, through backend 'nccl' (= RCCL on ROCm), plus the int64 form
used for VQ indices.  On a one-GPU box this still exercises communicator creation and the collective kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29541')
import torch
import torch.distributed as dist

rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
if world > torch.cuda.device_count():
    raise SystemExit(f'{world} ranks but {torch.cuda.device_count()} GPU(s): RCCL needs one GPU per rank')
dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
rows = 96 * 3
send = (torch.arange(rows * 4096, device=dev, dtype=torch.float32).reshape(rows, 4096) % 251 + rank).to(torch.bfloat16)
recv = torch.empty(world * rows, 4096, device=dev, dtype=torch.bfloat16)
work = dist.all_gather_into_tensor(recv, send, async_op=True)
work.wait()
idx = torch.full((rows,), rank, device=dev, dtype=torch.int64)
got = torch.empty(world * rows, device=dev, dtype=torch.int64)
dist.all_gather_into_tensor(got, idx)
torch.cuda.synchronize()
ok = all(torch.equal(recv[r * rows:(r + 1) * rows], (send.float() - rank + r).to(torch.bfloat16)) for r in range(world))
ok = ok and torch.equal(got, torch.arange(world, device=dev).repeat_interleave(rows))
if rank == 0:
    print('RCCL_CHECK', 'OK' if ok else 'MISMATCH', 'world', world, 'backend', dist.get_backend(), 'nccl/rccl version', torch.cuda.nccl.version(), flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
