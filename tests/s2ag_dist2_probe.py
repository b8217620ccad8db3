"""Helper processes of test_gpu_step.py::test_two_ranks_on_one_gpu_match_the_averaged_gradient_emulation.

    rank mode   RANK / WORLD_SIZE=2 / S2AG_DIST_BACKEND=gloo set by the test: one replica of the REAL data-parallel step
                (parallel.DataParallelContext + GradExchange inside Processor; gloo lets both replicas share cuda:0): its own
                batches and noise seed, three steps; saves the summed gradients and the weights after every step.
    emu mode    one plain process: per optimizer the gradients of the two replicas' batches are computed one after the other
                on the same weights (each with that replica's noise state), summed, and Adam consumes sum / 2 -- what the
                data-parallel step must equal (per-replica BatchNorm statistics, as under the reference's nn.DataParallel,
                processor_v2.py:167-172).

    python tests/s2ag_dist2_probe.py rank|emu OUT.pt [graph|eager] [overflow]
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
if os.environ.get('S2AG_EMU', '0') == '1':      # TEST INFRASTRUCTURE: both replicas on the CPU device model (tests/emu)
    sys.path.insert(0, os.path.join(HERE, 'emu'))
    import harness
    harness.install()

from oracle import s2ag_oracle as O  # noqa: E402  (input / weight recipes only)
from s2ag_testing import PASSES_PER_STEP, STEP_SEED, to_cuda  # noqa: E402
from test_gpu_step import make_processor  # noqa: E402

HIDDEN, N_WORDS, N_SPK, B, S0, STEPS, WORLD = 32, 64, 12, 6, 9500, 3, 2
PERM = [torch.tensor([3, 0, 5, 1, 2, 4]), torch.tensor([5, 4, 3, 2, 1, 0])]
_perm = [None]


def batch(rank, step):
    return to_cuda(O.recipe_inputs(B, 34, S0 + 100 + 10 * rank + step, N_WORDS, N_SPK))


def weights(pr):
    return dict(G={k: v.detach().cpu().clone() for k, v in pr.s2ag_generator.named_parameters()},
                D={k: v.detach().cpu().clone() for k, v in pr.s2ag_discriminator.named_parameters()})


def grads(pr):
    return dict(G=pr.gen_arena.grad.detach().cpu().clone(), D=pr.dis_arena.grad.detach().cpu().clone())


def run_rank(out_path, graph, overflow, flagged=False):
    from speech2affective_gestures_amd import noise, ops
    from speech2affective_gestures_amd import processor_v2 as P
    import torch.distributed as dist
    rank = int(os.environ['RANK'])
    P.torch.randperm = lambda n, *a, **k: _perm[0]
    _perm[0] = PERM[rank].cuda()
    extra = dict(max_words_per_clip=0) if overflow else {}      # capacity B * 1 rows: every batch overflows -> dense path
    pr, sds = make_processor(HIDDEN, N_WORDS, N_SPK, B, S0, 0.3, hip_graph=graph, **extra)
    assert pr.dp.active and pr.dp.world_size == WORLD and dist.get_backend() == 'gloo'
    # replicas start from rank 0's weights: knock rank 1's off and broadcast again (what Processor.__init__ does)
    if rank == 1:
        with torch.no_grad():
            pr.gen_arena.data.add_(0.05)
            pr.dis_arena.data.add_(0.05)
    pr.dp.broadcast_module(pr.s2ag_generator, pr.gen_arena)
    pr.dp.broadcast_module(pr.s2ag_discriminator, pr.dis_arena)
    start = weights(pr)
    noise.manual_seed(STEP_SEED + rank)
    res = dict(rank=rank, start=start, steps=[], ids=[])
    for s in range(STEPS):
        b = batch(rank, s)
        if graph:
            pr.train_step(b['in_text'], b['in_audio'], b['in_mfcc'], b['target'], b['vid'])
        else:
            pr.forward_pass_s2ag(b['in_text'], b['in_audio'], b['in_mfcc'], b['target'], b['vid'], True)
        torch.cuda.synchronize()
        res['steps'].append(dict(w=weights(pr), g=grads(pr), losses=dict(pr.last_losses)))
        res['ids'].append(sorted(set(b['in_text'].reshape(-1).tolist())))
    if flagged:
        # ONE rank's sticky error word goes up (as a cooperative-GRU / BatchNorm time-out would raise it): the step that
        # follows must leave the weights of EVERY rank alone (the word is MAX-reduced before each Adam) and raise everywhere
        before = weights(pr)
        if rank == 1:
            ops.coop_error_flag(pr.device).view(torch.int32).fill_(4)
        b = batch(rank, STEPS)
        raised = False
        try:
            pr.forward_pass_s2ag(b['in_text'], b['in_audio'], b['in_mfcc'], b['target'], b['vid'], True)
        except ops.CoopGruTimeout:
            raised = True
        torch.cuda.synchronize()
        res['flag_step'] = dict(raised=raised, before=before, after=weights(pr),
                                word=int(ops.coop_error_flag(pr.device).view(torch.int32)[0]))
    ex = pr._exchange()
    res.update(collectives=pr.dp.n_collectives, dense_fallbacks=ex.dense_fallbacks, row_cap=ex.row_cap,
               timeouts=ops.coop_gru_timeouts(), bytes=ex.bytes_per_step())
    torch.save(res, out_path)
    pr.dp.barrier()
    dist.destroy_process_group()


def run_emu(out_path, graph):
    from speech2affective_gestures_amd import noise, ops
    from speech2affective_gestures_amd import processor_v2 as P
    P.torch.randperm = lambda n, *a, **k: _perm[0]
    # every pass of the step on ONE stream, nothing hoisted or shared: the plainest form of the same arithmetic
    pr, sds = make_processor(HIDDEN, N_WORDS, N_SPK, B, S0, 0.3, hip_graph=False, overlap_passes=False,
                             share_encoders=False)
    assert not pr.dp.active
    dev = pr.device
    state = noise._dev_state(dev)

    def set_noise(seed, ctr):
        state.copy_(torch.tensor([seed, ctr], dtype=torch.int64))
    res = dict(start=weights(pr), steps=[])
    # graph mode: the first train_step call of a rank runs three eager warm-up steps on its batch before the replay
    seq = [0, 0, 0] + list(range(STEPS)) if graph else list(range(STEPS))
    for s, bi in enumerate(seq):
        bs = [batch(r, bi) for r in range(WORLD)]
        pres = [pr._make_pre_seq(b['target']) for b in bs]
        ops.begin_step()
        pr.s2ag_generator.share_passes = None
        g_sum = None
        for r, (b, pre) in enumerate(zip(bs, pres)):
            _perm[0] = PERM[r].cuda()
            set_noise(STEP_SEED + r, PASSES_PER_STEP * s)
            pr._dis_phase(b['in_text'], b['in_mfcc'], b['target'], b['vid'], pre, True, in_audio=b['in_audio'])
            g = pr.dis_arena.grad.clone()
            g_sum = g if g_sum is None else g_sum + g
        pr.dis_arena.grad.copy_(g_sum)
        pr.s2ag_dis_optimizer.step(1.0 / WORLD)
        h_sum = None
        for r, (b, pre) in enumerate(zip(bs, pres)):
            _perm[0] = PERM[r].cuda()
            set_noise(STEP_SEED + r, PASSES_PER_STEP * s + 3)        # the D phase drew three passes
            pr._gen_phase(b['in_text'], b['in_audio'], b['in_mfcc'], b['target'], b['vid'], pre, True)
            h = pr.gen_arena.grad.clone()
            h_sum = h if h_sum is None else h_sum + h
        pr.gen_arena.grad.copy_(h_sum)
        pr.s2ag_gen_optimizer.step(1.0 / WORLD)
        torch.cuda.synchronize()
        if s >= len(seq) - STEPS:
            res['steps'].append(dict(w=weights(pr), g=dict(G=h_sum.cpu(), D=g_sum.cpu())))
    res['timeouts'] = ops.coop_gru_timeouts()
    torch.save(res, out_path)


if __name__ == '__main__':
    mode, out_path = sys.argv[1], sys.argv[2]
    if mode == 'rank':
        run_rank(out_path, 'graph' in sys.argv[3:], 'overflow' in sys.argv[3:], 'flag' in sys.argv[3:])
    else:
        run_emu(out_path, 'graph' in sys.argv[3:])
