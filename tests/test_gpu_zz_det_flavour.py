"""Deterministic mode: a BUILD FLAVOUR of the library (libs2ag_hip_det.so, build.py --det, -DS2AG_DET=1; csrc/s2ag_common.h
det_enter / det_leave / S2AG_DET_WAVES_BEGIN..END), not a run-time word in the release kernels (as a run-time
word it had changed 39 default binaries).  The tests below need that flavour: in a process that loaded the release library they
skip, and `test_det_flavour_tests_in_a_process_of_their_own` runs this file again in a child process with S2AG_HIP_LIB pointing
at the det library.  On the CPU device model (tests/emu, always compiled with the mode available) they run in place.
The REPLAYED form (captured graphs) needs a real device and has not run on one yet.
The file name sorts LAST among the GPU tests on purpose: this is a debug flavour outside the hot path whose tests have never
seen hardware (GPU access closed since they were written) -- under `pytest -x` a surprise here must not cost the parity
suite's run."""
import copy
import os
import subprocess
import sys

import pytest
import torch

from oracle import s2ag_oracle as O
from s2ag_testing import STEP_SEED, to_cuda
from test_gpu_step import make_processor

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_flavour():
    from speech2affective_gestures_amd import ops
    if not ops.det_flavour():
        pytest.skip('release library loaded: runs in the child process of test_det_flavour_tests_in_a_process_of_their_own')


def test_det_flavour_tests_in_a_process_of_their_own():
    from speech2affective_gestures_amd import build, ops
    if ops.det_flavour():
        pytest.skip('this process already runs the det flavour')
    lib = build.FLAVOURS['det']['lib']
    assert os.path.exists(lib), f'{lib} missing: __graft_entry__.build() / build.py --det make it'
    env = dict(os.environ, S2AG_HIP_LIB=lib)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-m', 'gpu', '-x', '-k',
                        'not process_of_their_own and not release_library', '-p', 'no:cacheprovider'], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert ' passed' in r.stdout and 'skipped' not in r.stdout.splitlines()[-1], tail


def test_release_library_refuses_the_mode():
    """The release kernels carry no ordering code: asking for the mode must fail loudly, not silently run unordered."""
    from speech2affective_gestures_amd import ops
    if ops.det_flavour():
        pytest.skip('det flavour loaded')
    with pytest.raises(RuntimeError, match='build flavour'):
        ops.set_deterministic(True)
    assert not ops.deterministic()


def test_mode_is_set_by_every_trainer_and_side_streams_refuse_while_it_is_on():
    """the mode is process-wide library state -- a later Processor(deterministic=False) must switch it OFF (it
    forks passes onto side streams; two concurrent accumulating launches on one turn word would never end), and while it is
    on nothing may be marked as a side stream."""
    _need_flavour()
    from speech2affective_gestures_amd import ops
    try:
        pr, _ = make_processor(32, 64, 12, 4, 9100, 0.0, hip_graph=False, deterministic=True)
        assert ops.deterministic() and not ops.ASYNC_WGRAD
        with pytest.raises(RuntimeError, match='one stream'):
            ops.mark_side_stream(torch.cuda.Stream())
        ran = []
        ops.set_main_stream()
        ops.run_wgrad(lambda: ran.append(torch.cuda.current_stream()))
        assert ran == [torch.cuda.current_stream()]          # inline, not forked
        ops.join_side_streams()
        pr2, _ = make_processor(32, 64, 12, 4, 9100, 0.0, hip_graph=False)
        assert not pr2.deterministic and not ops.deterministic() and ops.ASYNC_WGRAD
        # after a turn time-out (sticky bit 3, turn word left anywhere) the mode can be armed again in-process:
        # switching it on starts from a zero turn word and a clear bit 3, the other error bits are left to the trainer
        key = torch.cuda.current_device()
        ops._DET_WORDS[key].fill_(7)
        ops._COOP_FLAG[key].fill_(8 | 16)
        ops.set_deterministic(True)
        assert int(ops._DET_WORDS[key]) == 0 and int(ops._COOP_FLAG[key]) == 16
        ops._COOP_FLAG[key].zero_()
    finally:
        ops.set_deterministic(False)


@pytest.mark.parametrize('hidden,B,mode', [(32, 6, 'fp32'), (300, 33, 'fp32'), (300, 6, 'bf16')])
def test_deterministic_mode_two_runs_are_bit_identical(monkeypatch, hidden, B, mode):
    """Deterministic mode (config switch DETERMINISTIC / Processor(deterministic=True); csrc/s2ag_common.h det_enter /
    det_leave / S2AG_DET_WAVES_BEGIN..END): two runs of the same two GAN steps from the same state leave EVERY weight, every
    gradient, every BatchNorm running statistic and every logged loss bit-identical -- where the default mode differs in
    the last bits of ~95 % of the tensors (fp32 atomics arrive in another order) and the replay test above has to allow for
    what Adam makes of that.  H = 300 goes through the cooperative GRU, the clip-resident TCN and the transpose-read weight
    gradients; 'bf16': the Conv1d path in bf16 mode (csrc/conv_bf16.hip, tcn_fused.hip, wgrad_tr.hip); on the CPU device model
    (tests/emu) the second run additionally uses another wavefront schedule."""
    _need_flavour()
    from speech2affective_gestures_amd import bf16, noise, ops
    from speech2affective_gestures_amd import processor_v2 as P
    n_words, n_spk, s0 = 64, 12, 9300
    perm = torch.arange(B - 1, -1, -1).cuda()
    monkeypatch.setattr(P.torch, 'randperm', lambda n, *a, **k: perm)
    batches = [to_cuda(O.recipe_inputs(B, 34, s0 + 100 + s, n_words, n_spk)) for s in range(2)]
    emu = None
    if os.environ.get('S2AG_EMU') == '1':
        import ctypes
        emu = ctypes.CDLL(os.environ['S2AG_HIP_LIB'])

    def run(sched):
        if emu is not None:
            emu.s2ag_emu_set_sched(sched, 11)
        noise.reset_sites(200)
        pr, _ = make_processor(hidden, n_words, n_spk, B, s0, 0.3, hip_graph=False, deterministic=True)
        assert pr.deterministic and not pr.overlap_passes and ops.deterministic()
        noise.manual_seed(STEP_SEED)
        losses = []
        with bf16.precision(mode):
            for b in batches:
                pr.forward_pass_s2ag(b['in_text'], b['in_audio'], b['in_mfcc'], b['target'], b['vid'], True)
                losses.append(dict(pr.last_losses))
        out = {}
        for tag, mod in (('G', pr.s2ag_generator), ('D', pr.s2ag_discriminator)):
            for k, p in mod.named_parameters():
                out[f'{tag}.{k}'] = p.detach().clone()
                if p.grad is not None:
                    out[f'{tag}.{k}.grad'] = p.grad.clone()
            for k, v in mod.state_dict().items():
                if 'running' in k:
                    out[f'{tag}.{k}'] = v.clone()
        return losses, out
    try:
        (l0, a), (l1, b) = run(0), run(2)
    finally:
        ops.set_deterministic(False)
        if emu is not None:
            emu.s2ag_emu_set_sched(int(os.environ.get('S2AG_EMU_SCHED', '0')), 1)
    assert l0 == l1
    differ = [k for k in a if not torch.equal(a[k], b[k])]
    assert not differ, (len(differ), len(a), differ[:8])
    assert ops.coop_gru_timeouts() == 0


def test_deterministic_graph_replay_equals_eager_bit_for_bit(monkeypatch):
    """In deterministic mode the three-segment hipGraph replay of the step and the eager step are the same sequence of launches
    on one stream with the same accumulation order: weights after three steps must be EQUAL, not 'Adam-close' (what
    test_hip_graph_replay_equals_eager has to accept in the default mode, where fp32 atomics arrive in another order)."""
    _need_flavour()
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
