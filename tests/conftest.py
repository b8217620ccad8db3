import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

if os.environ.get('S2AG_EMU', '0') == '1':
    # TEST INFRASTRUCTURE: the GPU parity tests on the CPU device model (tests/emu/README.md) -- the product's kernel sources
    # compiled for the host, called through the same C ABI; 'cuda' means 'cpu' in THIS test process only.
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
    import harness as _emu_harness
    _emu_harness.install()


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    if os.environ.get('S2AG_EMU', '0') != '1':
        return
    keep, gone = [], []
    for it in items:
        why = _emu_harness.deselected(it.nodeid)
        (gone if why else keep).append(it)
    if gone:
        config.hook.pytest_deselected(items=gone)
        items[:] = keep


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _same_noise_sites_whatever_ran_before(request):
    """Dropout / noise call sites are numbered as modules are constructed; without this a test's masks -- and with them
    WHICH activations happen to sit within rounding distance of a ReLU / LeakyReLU kink -- would depend on the tests
    that ran before it.  (At such an element product and oracle may legitimately take different sides; one flip moves a
    small-batch gradient by percents.  tools/diag_relu_flips.py, tools/diag_gru_in_grad.py)"""
    if 'gpu' in request.keywords:
        from speech2affective_gestures_amd import noise
        noise.reset_sites(0)
    yield


if os.environ.get('S2AG_POISON', '0') == '1':
    # Debug aid: every floating-point device buffer obtained through torch.empty / empty_like starts as NaN, so a kernel
    # that reads memory nobody wrote (results then depend on what the caching allocator hands out, i.e. on which tests
    # ran before) turns into NaNs instead of a state-dependent flake.
    import torch

    _empty, _empty_like = torch.empty, torch.empty_like

    def _poison(t):
        if t.is_cuda and t.is_floating_point() and t.numel():
            t.fill_(float('nan'))
        return t

    torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
    torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
