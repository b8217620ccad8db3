"""Embedding gather + dropout inside the fp32 clip-resident TCN forward launch (csrc/tcn_fused32.hip GATHER; ops.TCN32_GATHER):
TextEncoderTCN (net/multimodal_context_net_v2.py:61-91) with it against the same module with the gather as launches of its
own.  Same table rows, same dropout masks (site, index, per-pass snapshot) -> outputs bit-identical, single pass and three
lockstep passes; gradients: fp32 atomics / ordered folds either way (1e-5).

Opt-in path (written without access to a GPU): the tests skip unless S2AG_TCN32_GATHER=1."""
import os
import types

import pytest
import torch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get('S2AG_TCN32_GATHER', '0') != '1', reason='opt-in path: set S2AG_TCN32_GATHER=1')]


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(1e-9, float(b.abs().max())))


def _encoder():
    from speech2affective_gestures_amd import noise
    from speech2affective_gestures_amd.net.multimodal_context_net_v2 import TextEncoderTCN
    cfg = types.SimpleNamespace(hidden_size=300, n_layers=4, dropout_prob=0.3, freeze_wordembed=False)
    torch.manual_seed(3)
    noise.reset_sites(0)
    return TextEncoderTCN(cfg, 400, 300, dropout=0.3).cuda().train()


@pytest.mark.parametrize('B', [5, 64, 200])
def test_single_pass(B):
    from speech2affective_gestures_amd import noise, ops
    txt = _encoder()
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(0, 400, (B, 34), generator=g)
    ids[:, 20:] = 0
    dt = torch.randn(B, 34, 32, generator=g).cuda()
    res = {}
    prev = ops.TCN32_GATHER
    try:
        for gather in (False, True):
            ops.TCN32_GATHER = gather
            for p in txt.parameters():
                p.grad = None
            ops.begin_step()
            noise.manual_seed(5)
            t = txt(ids.cuda())[0]
            assert (type(t.grad_fn).__name__ != '') and txt.tcn.gather_capable(34, 300) == gather
            (t * dt).sum().backward()
            torch.cuda.synchronize()
            res[gather] = (t.detach().clone(), {k: p.grad.clone() for k, p in txt.named_parameters()})
    finally:
        ops.TCN32_GATHER = prev
    (t0, g0), (t1, g1) = res[False], res[True]
    assert torch.equal(t1, t0)
    for k in g0:
        assert rel(g1[k], g0[k]) < 1e-5, (k, rel(g1[k], g0[k]))


@pytest.mark.parametrize('B', [16, 128])
def test_three_passes_in_lockstep(B):
    from speech2affective_gestures_amd import noise, ops
    txt = _encoder()
    g = torch.Generator().manual_seed(6)
    ids = torch.randint(0, 400, (B, 34), generator=g).cuda()
    dt = torch.randn(B, 34, 32, generator=g).cuda()
    res = {}
    prev = ops.TCN32_GATHER
    try:
        for gather in (False, True):
            ops.TCN32_GATHER = gather
            for p in txt.parameters():
                p.grad = None
            ops.begin_step()
            noise.manual_seed(7)
            noises = noise.begin_passes(ids.device, 3)
            assert txt.lockstep_capable(ids)
            outs = txt.forward_passes(ids, noises)
            (outs[0] * dt).sum().backward()
            torch.cuda.synchronize()
            res[gather] = ([o.detach().clone() for o in outs], {k: p.grad.clone() for k, p in txt.named_parameters()})
    finally:
        ops.TCN32_GATHER = prev
    (o0, g0), (o1, g1) = res[False], res[True]
    for a, b in zip(o1, o0):
        assert torch.equal(a, b)
    for k in g0:
        assert rel(g1[k], g0[k]) < 1e-5, (k, rel(g1[k], g0[k]))
