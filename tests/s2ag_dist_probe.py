"""Helper process of test_gpu_step.py::test_world_size_1_rccl_between_graph_segments: three hipGraph-replayed training
steps at a small width; prints one JSON line (per-step losses, per-module weight checksums).  With S2AG_FORCE_DIST=1 the
trainer opens a world-size-1 RCCL process group and issues its collectives between / beside the graph segments exactly
as it does on 8 GPUs (the numbers must not change: SUM over one rank, scale 1/1)."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
if os.environ.get('S2AG_EMU', '0') == '1':      # TEST INFRASTRUCTURE: on the CPU device model (tests/emu)
    sys.path.insert(0, os.path.join(HERE, 'emu'))
    import harness
    harness.install()

from oracle import s2ag_oracle as O  # noqa: E402
from s2ag_testing import STEP_SEED, to_cuda  # noqa: E402
from test_gpu_step import make_processor  # noqa: E402


def main():
    from speech2affective_gestures_amd import noise, ops
    from speech2affective_gestures_amd import processor_v2 as P
    hidden, n_words, n_spk, B, s0 = 32, 64, 12, 8, 9300
    perm = torch.arange(B - 1, -1, -1).cuda()
    P.torch.randperm = lambda n, *a, **k: perm
    pr, _ = make_processor(hidden, n_words, n_spk, B, s0, 0.3, hip_graph=True)
    batches = [to_cuda(O.recipe_inputs(B, 34, s0 + 100 + s, n_words, n_spk)) for s in range(3)]
    noise.manual_seed(STEP_SEED)
    steps = []
    for b in batches + batches:
        m = pr.train_step(b['in_text'], b['in_audio'], b['in_mfcc'], b['target'], b['vid'])
        steps.append(dict(metric=m, **pr.last_losses))
    # S2AG_PROBE_RUNAHEAD=n: n more replayed steps WITHOUT the per-step read-back (train_step(sync=False): what bench.py times).
    # Nothing then makes the host wait for the device except GradExchange.exchange_rest's wait for the id count it sent ahead
    # at the start of the step (parallel.py): when train_step(k) returns, every earlier step must be complete on the device.
    runahead = None
    n_ahead = int(os.environ.get('S2AG_PROBE_RUNAHEAD', '0'))
    if n_ahead:
        events, worst = [], 0
        for k in range(n_ahead):
            b = batches[k % len(batches)]
            pr.train_step(b['in_text'], b['in_audio'], b['in_mfcc'], b['target'], b['vid'], sync=False)
            worst = max(worst, sum(0 if e.query() else 1 for e in events))       # earlier steps still running on the device
            e = torch.cuda.Event()
            e.record()
            events.append(e)
        torch.cuda.synchronize()
        b = batches[0]
        m = pr.train_step(b['in_text'], b['in_audio'], b['in_mfcc'], b['target'], b['vid'])          # one read-back at the end
        ex = pr._exchange()
        runahead = dict(steps=n_ahead, max_unfinished_earlier_steps=worst, last=dict(metric=m, **pr.last_losses),
                        dense_fallbacks=None if ex is None else ex.dense_fallbacks)
    sums = {}
    for tag, mod in (('G', pr.s2ag_generator), ('D', pr.s2ag_discriminator)):
        for k, v in mod.state_dict().items():
            if v.dtype == torch.float32 and 'running' not in k:
                top = tag + '.' + k.split('.')[0]
                sums[top] = sums.get(top, 0.0) + float(v.double().abs().sum())
    import torch.distributed as dist
    print('PROBE ' + json.dumps(dict(steps=steps, sums=sums, dist=dist.is_initialized(), active=pr.dp.active,
                                     segments=len(pr._graphed['segs'].graphs), timeouts=ops.coop_gru_timeouts(),
                                     collectives=getattr(pr.dp, 'n_collectives', None), runahead=runahead)), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
