"""Data-parallel replication of the GAN step: one process per GPU, RCCL over xGMI.

The reference's only parallelism is ``nn.DataParallel`` (processor_v2.py:167-172): single process, per-step
parameter broadcast, input scatter, output gather, gradients reduced onto GPU 0.  Here every rank owns a full
replica and its own B clips (weak scaling, per-replica BatchNorm statistics exactly as under DataParallel);
the only exchange is one SUM all-reduce per optimizer over the flat gradient arena (D: 313 k floats,
G: 13.2 M floats), after which Adam consumes grad/world_size.  With the gradient already contiguous there
is nothing to bucket: one collective moves the whole arena and RCCL pipelines it across the 7 xGMI links.
The frozen tri-modal baseline needs no communication.
"""
import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class DataParallelContext:
    rank: int = 0
    world_size: int = 1
    local_rank: int = 0

    @staticmethod
    def from_env(backend=None) -> 'DataParallelContext':
        world = int(os.environ.get('WORLD_SIZE', '1'))
        rank = int(os.environ.get('RANK', '0'))
        local = int(os.environ.get('LOCAL_RANK', '0'))
        force = os.environ.get('S2AG_FORCE_DIST', '0') == '1'     # world-size-1 process group: exercises the RCCL calls
        if (world > 1 or force) and not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29500')
            if backend is None:
                backend = 'nccl' if torch.cuda.is_available() else 'gloo'      # 'nccl' IS RCCL on ROCm
            if backend == 'nccl':
                torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        ctx = DataParallelContext(rank, world, local)
        ctx.forced = force and world == 1
        return ctx

    forced: bool = False

    @property
    def active(self) -> bool:
        return self.world_size > 1 or self.forced

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world_size

    def all_reduce_grads(self, arena) -> None:
        """SUM over ranks into ``arena.grad`` (scaled by 1/world inside the fused Adam)."""
        if self.active:
            dist.all_reduce(arena.grad, op=dist.ReduceOp.SUM)

    def broadcast_module(self, module, arena=None) -> None:
        """Rank 0's parameters (one flat broadcast when an arena exists) and buffers to everyone."""
        if not self.active:
            return
        if arena is not None:
            dist.broadcast(arena.data, src=0)
        else:
            for p in module.parameters():
                dist.broadcast(p.data, src=0)
        for b in module.buffers():
            dist.broadcast(b, src=0)

    def barrier(self) -> None:
        if self.active:
            dist.barrier()

    def max_over_ranks(self, value: float, device) -> float:
        if not self.active:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
