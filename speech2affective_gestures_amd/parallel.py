"""Data-parallel replication of the GAN step: one process per GPU, RCCL over xGMI.

The reference's only parallelism is ``nn.DataParallel`` (processor_v2.py:167-172): single process, per-step
parameter broadcast, input scatter, output gather, gradients reduced onto GPU 0.  Here every rank owns a full
replica and its own B clips (weak scaling, per-replica BatchNorm statistics exactly as under DataParallel); the
exchange per optimizer is a SUM over ranks of the flat gradient arena, after which Adam consumes grad/world_size.

``GradExchange`` is the schedule of the generator's exchange (the discriminator's 1.25 MB arena is one all-reduce):

  * the arena is laid out [word embedding | encoders | GRU decoder + out], i.e. in REVERSE order of the backward pass;
  * bucket A (GRU + out, 22.9 MB at the default config) is complete when the backward pass leaves the recurrent
    decoder: its all-reduce is launched right there, asynchronously on RCCL's stream, and runs beside the rest of the
    backward pass (encoders, TCN, embedding: ~0.7 ms of kernels against ~0.2-0.3 ms of collective);
  * bucket B (encoders, 6 MB) follows when backward ends;
  * the word embedding (24 MB dense at n_words = 20 000) never travels as a dense tensor: only the rows some replica's
    batch touched do (csrc/rows.hip: sorted unique ids, [id | row] records, ONE all-gather, merge in rank order --
    bit-identical on all ranks like an all-reduce).  The optimizer stays the reference's dense Adam.

xGMI is point to point (7 links per GPU), so a ring collective is bound per link: three mid-sized collectives per step
(22.9 MB hidden, 6 MB + ~1.4 MB x world exposed) instead of one exposed 53 MB all-reduce.  The frozen tri-modal
baseline needs no communication.  All collectives are issued from the host between / beside hipGraph segments
(processor_v2._GraphSegments); nothing here is captured into a graph.
"""
import os
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


@dataclass
class DataParallelContext:
    rank: int = 0
    world_size: int = 1
    local_rank: int = 0
    forced: bool = False
    n_collectives: int = 0          # issued so far (tests)

    @staticmethod
    def from_env(backend=None) -> 'DataParallelContext':
        world = int(os.environ.get('WORLD_SIZE', '1'))
        rank = int(os.environ.get('RANK', '0'))
        local = int(os.environ.get('LOCAL_RANK', '0'))
        from . import config
        force = bool(config.get('FORCE_DIST'))     # world-size-1 process group: exercises the RCCL calls
        if (world > 1 or force) and not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29500')
            if backend is None:
                # 'nccl' IS RCCL on ROCm.  S2AG_DIST_BACKEND=gloo lets several ranks share ONE GPU (RCCL refuses
                # duplicate devices): the two-rank test of the real step on a single MI355X (tests/test_gpu_step.py)
                backend = config.get('DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
            if backend == 'nccl':
                torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        return DataParallelContext(rank, world, local, force and world == 1)

    @property
    def active(self) -> bool:
        return self.world_size > 1 or self.forced

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world_size

    # ---- collectives --------------------------------------------------------------------------------
    def all_reduce(self, t: torch.Tensor, async_op: bool = False):
        """SUM over ranks, in place.  ``async_op``: returns the work handle; the caller's stream does not wait for the
        collective until ``handle.wait()`` (RCCL runs it on its own stream, ordered after what the caller has queued)."""
        if not self.active:
            return None
        self.n_collectives += 1
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)

    def all_reduce_max(self, t: torch.Tensor) -> None:
        """MAX over ranks, in place (stream ordered, no host wait)."""
        if self.world_size > 1:
            self.n_collectives += 1
            dist.all_reduce(t, op=dist.ReduceOp.MAX)

    def sync_error_flag(self, flag: Optional[torch.Tensor]) -> None:
        """The sticky error word of the step (ops.coop_error_flag: cooperative-GRU / one-launch-BatchNorm time-outs, int32)
        becomes the MAX over ranks, IN PLACE, stream ordered.  Issued right before every Adam launch: the device-side
        step guard (csrc/misc.hip adam_k) reads this very word, so either every replica applies the all-reduced gradient
        or none does -- a rank that timed out would otherwise skip its update while the others step with a gradient that
        already contains its invalid contribution, and the replicas would drift apart for good.  (RCCL has no bitwise OR;
        MAX keeps 'non-zero somewhere' and the largest bit pattern, which is all the guard and the trainer's read-back
        need.)"""
        if flag is None or self.world_size <= 1:
            return
        self.n_collectives += 1
        dist.all_reduce(flag.view(torch.int32), op=dist.ReduceOp.MAX)

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        """out (world * n,) <- inp (n,) of every rank, rank-major."""
        if not self.active:
            out.copy_(inp)
            return
        self.n_collectives += 1
        if dist.get_backend() == 'gloo':         # no all_gather_into_tensor for device tensors there
            dist.all_gather(list(out.view(self.world_size, -1).unbind(0)), inp)
        else:
            dist.all_gather_into_tensor(out, inp)

    def all_reduce_grads(self, arena) -> None:
        """SUM over ranks into ``arena.grad`` (scaled by 1/world inside the fused Adam)."""
        self.all_reduce(arena.grad)

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


@dataclass
class RowKernels:
    """The three device kernels of the touched-row exchange (csrc/rows.hip through ops.py in the product; the CPU tests
    of the schedule pass torch doubles).  All write into caller-owned static buffers (hipGraph friendly)."""
    unique: Callable      # (ids (n,) i64, n_entries, uids_out (cap,) i32) -> (1,) i32 tensor = number of distinct ids;
                          # uids_out <- the first cap of them, sorted, padded with n_entries
    pack: Callable        # (dense (n_entries, dim), uids, records_out (cap, dim + 1) f32)
    merge: Callable       # (gathered (world, cap, dim + 1), dense): dense rows overwritten with the rank-ordered sums
    unique_flagged: Optional[Callable] = None   # `unique` that also raises the sticky error word when the ids overflow the cap


@dataclass
class GradExchange:
    """Exchange schedule of one flat gradient arena.  ``split`` = element offset where bucket A begins (everything the
    backward pass finishes FIRST lives behind it); ``rows`` = (lo, hi, n_entries, dim) of a row-sparse tensor at the
    front of the arena, or None.  Per step, in this order:

        precheck(ids)     host, at the START of the step (eager, never captured): the batch's distinct ids are listed and
                          counted, MAX over ranks of the count travels to pinned host memory -- nobody waits
        launch_a()        host, when bucket A is complete on the current stream (async all-reduce, nobody waits yet)
        pack_rows(ids)    device kernels (capturable), when the whole gradient is complete
        exchange_rest()   host: all-reduce of bucket B, all-gather of the row records, then wait for bucket A
        merge_rows()      device kernel (capturable): touched rows <- sum over ranks, in rank order

    afterwards ``grad`` holds the SUM over ranks everywhere.  If ANY rank's batch holds more distinct ids than ``row_cap``
    (a too small ``args.max_words_per_clip``, data swapped under the trainer) the records would be truncated and the
    replicas would drift apart: the count that ``precheck`` sent ahead (it depends on the token ids only, so it is on the
    host long before the gradients exist when steps are synchronous; see exchange_rest for run-ahead loops) makes EVERY rank take the dense all-reduce of the
    row block for that step instead (``dense_fallbacks`` counts them) and the merge finds only sentinels."""
    dp: DataParallelContext
    grad: torch.Tensor
    split: int
    rows: Optional[Tuple[int, int, int, int]] = None
    row_cap: int = 0
    kernels: Optional[RowKernels] = None
    _work: list = field(default_factory=list)
    dense_fallbacks: int = 0

    def __post_init__(self):
        n = self.grad.numel()
        lo = self.rows[1] if self.rows else 0
        assert 0 <= lo <= self.split <= n, (lo, self.split, n)
        if self.rows:
            assert self.rows[0] == 0 and self.kernels is not None and self.row_cap > 0
            _, hi, n_entries, dim = self.rows
            assert hi == n_entries * dim
            dev = self.grad.device
            self.uids = torch.full((self.row_cap,), n_entries, dtype=torch.int32, device=dev)
            self.records = torch.zeros(self.row_cap, dim + 1, dtype=torch.float32, device=dev)
            self.gathered = torch.zeros(self.dp.world_size, self.row_cap, dim + 1, dtype=torch.float32, device=dev)
            self._cnt = torch.zeros(1, dtype=torch.int32, device=dev)
            self._cnt_host = torch.zeros(1, dtype=torch.int32).pin_memory() if dev.type == 'cuda' else None
            self._cnt_ev = torch.cuda.Event() if dev.type == 'cuda' else None
        self._pre = False

    @property
    def buckets(self) -> List[Tuple[str, int, int]]:
        lo = self.rows[1] if self.rows else 0
        return [('A', self.split, self.grad.numel()), ('B', lo, self.split)]

    def bytes_per_step(self) -> dict:
        """Payload each rank contributes per step (fp32 bytes) -- what the schedule moves instead of the dense arena."""
        out = {name: 4 * (hi - lo) for name, lo, hi in self.buckets}
        out['rows'] = 4 * self.records.numel() if self.rows else 0
        out['dense_arena'] = 4 * self.grad.numel()
        return out

    def launch_a(self) -> None:
        if self.split < self.grad.numel():
            w = self.dp.all_reduce(self.grad[self.split:], async_op=True)
            if w is not None:
                self._work.append(w)

    def precheck(self, ids: torch.Tensor) -> None:
        if self.rows is None:
            return
        _, hi, n_entries, dim = self.rows
        self._cnt.copy_(self.kernels.unique(ids.reshape(-1), n_entries, self.uids))
        self.dp.all_reduce_max(self._cnt)
        if self._cnt_host is not None:
            self._cnt_host.copy_(self._cnt, non_blocking=True)
            self._cnt_ev.record()
        self._pre = True

    def pack_rows(self, ids: torch.Tensor) -> None:
        if self.rows is None:
            return
        _, hi, n_entries, dim = self.rows
        if not self._pre:       # no precheck this step: list the ids here; an overflow cannot be repaired any more (no rank
            # knows the others' counts), so it must at least be REPORTED: unique_flagged raises the sticky error word
            (self.kernels.unique_flagged or self.kernels.unique)(ids.reshape(-1), n_entries, self.uids)
        self.kernels.pack(self.grad[:hi].view(n_entries, dim), self.uids, self.records)

    def exchange_rest(self) -> None:
        lo = self.rows[1] if self.rows else 0
        if lo < self.split:
            self.dp.all_reduce(self.grad[lo:self.split])
        if self.rows is not None:
            over = False
            if self._pre:
                if self._cnt_ev is not None and not self._cnt_ev.query():
                    # recorded at the START of the step, so normally long since complete; in sync=False / replayed-graph
                    # loops the host runs ahead of the device and this wait is what bounds the run-ahead to ~one step
                    # (the branch below must be taken with THIS step's count on every rank)
                    self._cnt_ev.synchronize()
                over = int((self._cnt_host if self._cnt_host is not None else self._cnt)[0]) > self.row_cap
                self._pre = False
            if over:
                self.dense_fallbacks += 1
                self.dp.all_reduce(self.grad[:self.rows[1]])
                self.gathered.view(torch.int32)[:, :, 0] = self.rows[2]   # sentinels: merge_rows leaves the summed rows alone
            else:
                self.dp.all_gather(self.gathered.view(-1), self.records.view(-1))
        for w in self._work:
            w.wait()
        self._work.clear()

    def merge_rows(self) -> None:
        if self.rows is None:
            return
        _, hi, n_entries, dim = self.rows
        self.kernels.merge(self.gathered, self.grad[:hi].view(n_entries, dim))
