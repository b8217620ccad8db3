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
    assert lib.s2ag_abi_version() == 1
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
        (['tests/test_gpu_det_flavour.py'], 'deterministic_mode and 32-6', 1, 0),
    'one whole GAN step strictly: branch decisions of all seven module passes replayed in the oracle (H = 300)':
        (['tests/test_gpu_step.py'], 'one_step_strictly and 300-6', 1, 0),
    'strict parity of both discriminators with the product\'s branch decisions (small batch)':
        (['tests/test_gpu_modules.py'], 'branch_decisions[5]', 1, 0),
}


@pytest.mark.parametrize('group', list(GROUPS))
def test_gpu_parity_tests_on_the_cpu_device_model(emu_lib, group):
    files, expr, at_least, sched = GROUPS[group]
    env = dict(os.environ, S2AG_EMU='1', S2AG_EMU_SCHED=str(sched))
    env.pop('S2AG_HIP_LIB', None)
    cmd = [sys.executable, '-m', 'pytest', *files, '-m', 'gpu', '-q', '-p', 'no:cacheprovider', '-k', expr]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = p.stdout[-3000:]
    m = re.search(r'(\d+) passed', p.stdout)
    assert p.returncode == 0, tail
    assert m and int(m.group(1)) >= at_least, tail
