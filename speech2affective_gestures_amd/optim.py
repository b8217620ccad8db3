"""Flat parameter arenas and the fused Adam step.

MI355X-first memory layout: every trainable tensor of a network is a view into ONE contiguous fp32
arena, its gradient a view into a second one.  Consequences: ``zero_grad`` is one memset, Adam is one
kernel launch per optimizer (s2ag_adam_step), and the data-parallel gradient exchange is a single RCCL
all-reduce over the arena (parallel.py) instead of one collective per tensor.
"""
import ctypes as C
from typing import Iterable, List

import torch

from . import _lib as L


class ParamArena:
    def __init__(self, params: Iterable[torch.nn.Parameter], first: Iterable[torch.nn.Parameter] = ()):
        """``first``: parameters placed at the front of the arena (the rest keep their order) -- the data-parallel
        exchange wants the row-sparse word embedding in front of the dense buckets (parallel.GradExchange)."""
        seen, uniq = set(), []
        params = list(params)
        front = [p for p in first if any(p is q for q in params)]
        for p in front + params:
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                uniq.append(p)
        if not uniq:
            raise ValueError('no trainable parameters')
        dev = uniq[0].device
        L.require_gpu_device(dev, 'ParamArena (its parameters)')
        self.params: List[torch.nn.Parameter] = uniq
        self.epoch = 0          # bumped whenever the arena's values change behind torch's back (FusedAdam.step)
        sizes = [(p.numel() + 3) // 4 * 4 for p in uniq]          # keep every view 16-byte aligned
        self.numel = sum(sizes)
        self.data = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        off = 0
        self.offsets = []
        for p, n in zip(uniq, sizes):
            view = self.data[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[off:off + p.numel()].view(p.shape)
            p._s2ag_arena = self
            self.offsets.append(off)
            off += n

    def offset_of(self, p: torch.nn.Parameter) -> int:
        """Element offset of ``p``'s view inside ``data`` / ``grad``."""
        for q, off in zip(self.params, self.offsets):
            if q is p:
                return off
        raise KeyError('parameter is not in this arena')

    def zero_grad(self):
        self.grad.zero_()
        for p, off in zip(self.params, self.offsets):      # re-attach if autograd replaced a grad tensor
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + p.numel()].view(p.shape)


class FusedAdam:
    """torch.optim.Adam(lr, betas, eps=1e-8) semantics (processor_v2.py:215-220) as one launch."""

    def __init__(self, arena: ParamArena, lr: float, betas=(0.5, 0.999), eps: float = 1e-8):
        self.arena = arena
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.exp_avg = torch.zeros_like(arena.data)
        self.exp_avg_sq = torch.zeros_like(arena.data)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=arena.data.device)
        self.param_groups = [dict(lr=self.lr, betas=self.betas, eps=self.eps, params=arena.params)]

    def zero_grad(self, set_to_none: bool = False):
        self.arena.zero_grad()

    def step(self, grad_scale: float = 1.0):
        lib = L.load()
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        lr = float(self.param_groups[0]['lr'])
        L.check(lib.s2ag_counter_inc(C.c_void_p(self.step_count.data_ptr()), None, s), 'counter_inc')
        a = self.arena
        L.check(lib.s2ag_adam_step(C.c_void_p(a.data.data_ptr()), C.c_void_p(a.grad.data_ptr()),
                                   C.c_void_p(self.exp_avg.data_ptr()), C.c_void_p(self.exp_avg_sq.data_ptr()),
                                   a.numel, lr, self.betas[0], self.betas[1], self.eps,
                                   C.c_void_p(self.step_count.data_ptr()), float(grad_scale), s), 'adam_step')
        a.epoch += 1            # tensors derived from these parameters (ops._DerivedGroup) are stale now

    def state_dict(self):
        return dict(exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, step=self.step_count, lr=self.param_groups[0]['lr'])

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        self.step_count.copy_(sd['step'])
        self.param_groups[0]['lr'] = sd['lr']
