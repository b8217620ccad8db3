"""CPU suite: a bounded part of the `-m gpu` parity tests on the CPU device model of tests/emu (README.md there) -- the
product's own kernel sources compiled for the host, driven through the same C ABI by the product's own Python host code.

GPU access was closed from outside for the second half of r03 and for r04: these runs are the evidence that the kernels
written or changed in that time (and the ones they sit next to) compute what the oracle / torch references say, lane
layouts, LDS images, barriers, tickets and cross-workgroup protocols included.  Each group is one pytest subprocess with
S2AG_EMU=1 (the harness makes 'cuda' mean 'cpu' in THAT process only).  The whole sweep, incl. the slow tests, is
tools/run_emu_suite.py -> profiles/r04_emu_suite.txt."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))


@pytest.fixture(scope='module')
def emu_lib():
    import build_emu
    return build_emu.build()


def test_device_model_library_exports_the_c_abi(emu_lib):
    import ctypes
    from speech2affective_gestures_amd import _lib
    lib = ctypes.CDLL(emu_lib)
    for name in _lib.SIGNATURES:
        assert hasattr(lib, name), name
    assert lib.s2ag_abi_version() == 2
    assert hasattr(lib, 's2ag_emu_set_sched') and hasattr(lib, 's2ag_emu_counters')


def test_product_does_not_know_the_device_model():
    """Nothing under speech2affective_gestures_amd/ may mention tests/emu or S2AG_EMU: the model is a checker."""
    pkg = os.path.join(ROOT, 'speech2affective_gestures_amd')
    for base, _, names in os.walk(pkg):
        if '_obj' in base or '__pycache__' in base:
            continue
        for n in names:
            if n.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(base, n)).read()
                assert 'S2AG_EMU' not in text and 'tests/emu' not in text and 'libs2ag_emu' not in text, n


# (files, -k expression, at least this many tests must pass, S2AG_EMU_SCHED)
GROUPS = {
    'wave head: statistics, forward, one-launch backward (csrc/wave12.hip, both modes)':
        (['tests/test_gpu_wave12.py'], 'not 36267', 20, 0),
    'wave head under the reversed wavefront order (a missing barrier changes results)':
        (['tests/test_gpu_wave12.py'], 'test_backward and 1000', 8, 2),
    'bf16 wave tail with folded BatchNorms (csrc/wave_fused.hip)':
        (['tests/test_gpu_wave_fused.py'], 'not 40', 10, 0),
    'embedding / rows / losses / Adam / BatchNorm one-launch grid wait (csrc/misc.hip, rows.hip, norm_elementwise.hip)':
        (['tests/test_gpu_ops.py'], 'embedding or rows or loss or adam or batch_norm or rng', 20, 0),
    'cooperative GRU: tagged-cell exchange between workgroups, three product modes (csrc/gru_coop.hip)':
        (['tests/test_gpu_ops.py'], 'test_gru_forward_backward and (9-6 or 17-1 or 33-2 or 16-3 or 5-7 or 3-5)', 6, 0),
    'deterministic mode: two GAN steps, two wavefront schedules, every weight / gradient / statistic bit-identical':
        (['tests/test_gpu_zz_det_flavour.py'], 'deterministic_mode and 32-6', 1, 0),
    'one whole GAN step strictly: branch decisions of all seven module passes replayed in the oracle (H = 300)':
        (['tests/test_gpu_step.py'], 'one_step_strictly and 300-6', 1, 0),
    'strict parity of both discriminators with the product\'s branch decisions (small batch)':
        (['tests/test_gpu_modules.py'], 'branch_decisions[5]', 1, 0),
    'r06 opt-in kernel variants against their default kernels (csrc/wgrad_tr32p.hip, tcn32p.hip, bn_foldapply.hip, emb_rows.hip), reversed wavefront order':
        (['tests/test_gpu_zy_variants.py'], '', 30, 2),
}


@pytest.mark.parametrize('group', list(GROUPS))
def test_gpu_parity_tests_on_the_cpu_device_model(emu_lib, group):
    files, expr, at_least, sched = GROUPS[group]
    env = dict(os.environ, S2AG_EMU='1', S2AG_EMU_SCHED=str(sched))
    env.pop('S2AG_HIP_LIB', None)
    cmd = [sys.executable, '-m', 'pytest', *files, '-m', 'gpu', '-q', '-p', 'no:cacheprovider'] + (['-k', expr] if expr else [])
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = p.stdout[-3000:]
    m = re.search(r'(\d+) passed', p.stdout)
    assert p.returncode == 0, tail
    assert m and int(m.group(1)) >= at_least, tail


def test_bench_launch_path_with_two_ranks(emu_lib):
    """`bench.py --gpus 2` launched exactly as the driver launches it -- python -m torch.distributed.run
    --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 --steps K --warmup W -- has never
    run anywhere (no multi-GPU node so far).  Here it runs with two gloo ranks on the CPU device model at a reduced width
    (tests/s2ag_emu_bench.py; bench.py --dry-width): rank environment, process group, the data-parallel step (four segments
    with the collectives between them), barrier + max-over-ranks timing, exactly ONE JSON line, from rank 0, with the
    contract's keys and the whole-job aggregate.  A launch-path test, not a measurement."""
    import json
    port = 29800 + os.getpid() % 150
    env = dict(os.environ)
    env.pop('S2AG_HIP_LIB', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', 's2ag_emu_bench.py'), '--gpus', '2', '--steps', '2',
           '--warmup', '1', '--batch', '4', '--dry-width', '32,64,12', '--no-extras', '--no-cpu-baseline']
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, lines                         # rank 0 prints; rank 1 prints nothing of the kind
    d = json.loads(lines[0])
    assert d['metric'].startswith('DRY_RUN_reduced_width_not_the_benchmark')        # cannot be mistaken for a measurement
    for k in ('value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config'):
        assert k in d, k
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['unit'] == 'clips/s'
    c = d['config']
    assert c['parallelism'] == 'dp2' and c['batch_per_gpu'] == 4 and c['global_batch'] == 8 and c['hidden_size'] == 32
    # whole-job aggregate: clips of BOTH ranks over the max-over-ranks time
    assert d['value'] == pytest.approx(8 * 2 / (d['ms_per_step'] * 2e-3), rel=1e-6)
    ex = c['gradient_exchange_bytes_per_rank']
    assert ex['A'] > 0 and ex['B'] > 0 and ex['rows'] > 0 and ex['A'] + ex['B'] + ex['rows'] < ex['dense_arena']
    assert all(v == v and abs(v) < 1e6 for v in c['last_step_losses'].values()), c['last_step_losses']


def test_bench_line_with_its_extras_assembles(emu_lib):
    """bench.py's default single-GPU line carries more than the two-rank dry run exercises: `alt_modes` and the top-level
    `value_fp32_equivalent` taken from it (r05), `cpu_baseline` + `gpu_over_cpu`, `long_context_run` on a second processor.
    None of that code can run without a device; here it runs on the model at a reduced width with the full-width sub-benches
    stubbed (tests/s2ag_emu_bench.py --stub-heavy-extras) -- a typo in the assembly would otherwise first show in the
    driver's BENCH run."""
    import json
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    env.pop('S2AG_HIP_LIB', None)
    # r06: the line also carries `opt_in_variants_all_on`, measured by a CHILD process with the variant switches in its
    # environment (isolated: the variants have never run on hardware).  Here the child is the same wrapper on the model.
    env['BENCH_PROBE_CMD'] = f"{sys.executable} {os.path.join(ROOT, 'tests', 's2ag_emu_bench.py')} --stub-heavy-extras"
    cmd = [sys.executable, os.path.join(ROOT, 'tests', 's2ag_emu_bench.py'), '--stub-heavy-extras', '--steps', '1', '--warmup', '1',
           '--batch', '4', '--dry-width', '32,64,12', '--no-graph']       # (eager steps: no capture warm-ups on the model)
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert set(d['alt_modes']) >= {'fp32_equivalent_3_bf16_pieces', 'f32_mfma_everywhere', 'bf16_conv_path', 'bf16_step'}
    assert d['value_fp32_equivalent'] == d['alt_modes']['fp32_equivalent_3_bf16_pieces']['clips_per_s']
    assert d['cpu_baseline']['kind'] == 'port' and d['cpu_baseline']['value'] > 0 and d['cpu_baseline']['cores'] >= 1
    assert '5 timed GAN steps' in d['cpu_baseline']['sample']
    assert d['gpu_over_cpu'] == pytest.approx(d['value'] / d['cpu_baseline']['value'])
    assert d['long_context_run']['clips_per_s'] > 0 and d['long_context_run']['steps'] == 10
    v = d['opt_in_variants_all_on']
    assert v['ok'], v
    assert {k: int(x) for k, x in v['switches'].items() if k in ('WGRAD32_PIPE', 'TCN32_PAIR', 'BN_FOLD_APPLY', 'EMB_BWD_ROWS')} == \
        {'WGRAD32_PIPE': 2, 'TCN32_PAIR': 1, 'BN_FOLD_APPLY': 1, 'EMB_BWD_ROWS': 1} and v['step_clips_per_s'] > 0
    assert v['step_vs_default'] == pytest.approx(v['step_clips_per_s'] / d['value'])
    assert not set(d['config']['non_default_switches']) & {'WGRAD32_PIPE', 'TCN32_PAIR', 'BN_FOLD_APPLY', 'EMB_BWD_ROWS'}      # the line itself: the default path
