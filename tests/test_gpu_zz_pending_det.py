"""Deterministic mode through hipGraph replay (config switch DETERMINISTIC; csrc/s2ag_common.h det_enter / det_leave): the eager
form of this property -- two runs bit-identical -- is tests/test_gpu_step.py::test_deterministic_mode_two_runs_are_bit_identical,
which also runs on the CPU device model.  The REPLAYED form needs a real device (captured graphs are not modelled) and has
never run on one: it lives in a file that sorts last (see tests/test_gpu_zz_pending_wave32.py)."""
import copy

import pytest
import torch

from oracle import s2ag_oracle as O
from s2ag_testing import STEP_SEED, to_cuda
from test_gpu_step import make_processor

pytestmark = pytest.mark.gpu


def test_deterministic_graph_replay_equals_eager_bit_for_bit(monkeypatch):
    """In deterministic mode the three-segment hipGraph replay of the step and the eager step are the same sequence of launches
    on one stream with the same accumulation order: weights after three steps must be EQUAL, not 'Adam-close' (what
    test_hip_graph_replay_equals_eager has to accept in the default mode, where fp32 atomics arrive in another order)."""
    from speech2affective_gestures_amd import noise, ops
    from speech2affective_gestures_amd import processor_v2 as P
    hidden, n_words, n_spk, B, s0 = 32, 64, 12, 8, 9000
    perm = torch.arange(B - 1, -1, -1).cuda()
    monkeypatch.setattr(P.torch, 'randperm', lambda n, *a, **k: perm)
    batches = [to_cuda(O.recipe_inputs(B, 34, s0 + 100 + s, n_words, n_spk)) for s in range(3)]

    def run(graph):
        noise.reset_sites(100)
        pr, _ = make_processor(hidden, n_words, n_spk, B, s0, 0.3, hip_graph=graph, deterministic=True)
        if graph:       # capture (3 warm-up steps touch the state) ... then rewind everything to the start state
            state = dict(G=copy.deepcopy(pr.s2ag_generator.state_dict()), D=copy.deepcopy(pr.s2ag_discriminator.state_dict()),
                         T=copy.deepcopy(pr.trimodal_generator.state_dict()),
                         og=copy.deepcopy(pr.s2ag_gen_optimizer.state_dict()), od=copy.deepcopy(pr.s2ag_dis_optimizer.state_dict()))
            b = batches[0]
            pr._build_graphed(b['in_text'], b['in_audio'], b['in_mfcc'], b['target'], b['vid'])
            pr.s2ag_generator.load_state_dict(state['G'])
            pr.s2ag_discriminator.load_state_dict(state['D'])
            pr.trimodal_generator.load_state_dict(state['T'])
            pr.s2ag_gen_optimizer.load_state_dict(state['og'])
            pr.s2ag_dis_optimizer.load_state_dict(state['od'])
        noise.manual_seed(STEP_SEED)
        losses = []
        for b in batches:
            pr.train_step(b['in_text'], b['in_audio'], b['in_mfcc'], b['target'], b['vid'])
            losses.append(dict(pr.last_losses))
        return losses, {k: v.clone() for k, v in list(pr.s2ag_generator.state_dict().items()) +
                        [('D.' + k, v) for k, v in pr.s2ag_discriminator.state_dict().items()]}
    try:
        (le, sd_e), (lg, sd_g) = run(False), run(True)
    finally:
        ops.set_deterministic(False)
    assert le == lg
    differ = [k for k in sd_e if not torch.equal(sd_e[k], sd_g[k])]
    assert not differ, (len(differ), differ[:8])
