"""Embedding gather + dropout inside the fp32 clip-resident TCN forward launch (csrc/tcn_fused32.hip GATHER; ops.TCN32_GATHER):
TextEncoderTCN (net/multimodal_context_net_v2.py:61-91) with it against the same module with the gather as launches of its
own.  Same table rows, same dropout masks (site, index, per-pass snapshot) -> outputs bit-identical, single pass and three
lockstep passes; gradients: fp32 atomics / ordered folds either way (1e-5).

Opt-in paths written without access to a GPU (config switches TCN_GATHER, TCN_RING_DEEP, EMB_FWD_ROWS): the tests arm them
themselves and live in a file that sorts LAST (see tests/test_gpu_zz_pending_wave32.py); they pass on the CPU device model."""
import os
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(1e-9, float(b.abs().max())))


def l2(a, b):
    a, b = a.detach().float().cpu().double(), b.detach().float().cpu().double()
    return float((a - b).norm() / max(1e-12, float(b.norm())))


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


@pytest.mark.parametrize('B', [5, 64, 200])
def test_embedding_gather_inside_the_clip_resident_tcn_launch(B):
    """csrc/tcn_fused.hip GATHER: TextEncoderTCN in bf16 mode with the embedding rows + dropout formed by the TCN forward launch's
    loader, against the same module with the gather as a launch of its own.  Same table rows, same dropout mask (site and
    index), same bf16 rounding -> the output is bit-identical; every gradient but the table's likewise (the table's is a sum
    of fp32 atomics either way: 1e-5)."""
    import types
    from speech2affective_gestures_amd import bf16, noise, ops
    from speech2affective_gestures_amd.net.multimodal_context_net_v2 import TextEncoderTCN
    cfg = types.SimpleNamespace(hidden_size=300, n_layers=4, dropout_prob=0.3, freeze_wordembed=False)
    torch.manual_seed(3)
    noise.reset_sites(0)
    txt = TextEncoderTCN(cfg, 400, 300, dropout=0.3).cuda().train()
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(0, 400, (B, 34), generator=g)
    ids[:, 20:] = 0
    dt = torch.randn(B, 34, 32, generator=g).cuda()
    res = {}
    prev = bf16.TCN_GATHER
    try:
        for gather in (False, True):
            bf16.TCN_GATHER = gather
            for p in txt.parameters():
                p.grad = None
            ops.begin_step()
            noise.manual_seed(5)
            with bf16.precision('bf16'):
                t = txt(ids.cuda())[0]
                (t * dt).sum().backward()
            torch.cuda.synchronize()
            res[gather] = (t.detach().clone(), {k: p.grad.clone() for k, p in txt.named_parameters()})
    finally:
        bf16.TCN_GATHER = prev
    (t0, g0), (t1, g1) = res[False], res[True]
    assert torch.equal(t1, t0)
    for k in g0:
        if k == 'embedding.weight':
            assert rel(g1[k], g0[k]) < 1e-5, (k, rel(g1[k], g0[k]))
        else:
            assert l2(g1[k], g0[k]) < 1e-5, (k, l2(g1[k], g0[k]))     # weight gradients leave through fp32 atomics / ordered folds


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
@pytest.mark.parametrize('B', [3, 70])
def test_deep_weight_rings_are_bit_identical(mode, B):
    """config switch TCN_RING_DEEP (csrc/tcn_fused.hip RING = 8, csrc/tcn_fused32.hip RING = 6): twice the weight fragments in
    flight, the same products in the same order -- TextEncoderTCN's output must not move by one bit; the gradients (sums of
    fp32 atomics whose order varies from run to run) within 1e-6.  The test flips the switch itself (s2ag_set_option through config.override)."""
    from speech2affective_gestures_amd import bf16, config, noise, ops
    txt = _encoder()
    g = torch.Generator().manual_seed(8)
    ids = torch.randint(0, 400, (B, 34), generator=g).cuda()
    dt = torch.randn(B, 34, 32, generator=g).cuda()
    res = {}
    for deep in (False, True):
        with config.override('TCN_RING_DEEP', deep), bf16.precision(mode):
            assert ops._lib().s2ag_get_option(b'TCN_RING_DEEP') == int(deep)
            for p in txt.parameters():
                p.grad = None
            ops.begin_step()
            noise.manual_seed(9)
            t = txt(ids)[0]
            (t * dt).sum().backward()
            torch.cuda.synchronize()
            res[deep] = (t.detach().clone(), {k: p.grad.clone() for k, p in txt.named_parameters()})
    (t0, g0), (t1, g1) = res[False], res[True]
    assert torch.equal(t1, t0)                       # the forward is a chain of identical products: not one bit moves
    for k in g0:                                     # gradients leave through fp32 atomics (order varies run to run)
        assert rel(g1[k], g0[k]) < 1e-6, (k, rel(g1[k], g0[k]))


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_embedding_forward_row_form_equals_element_form(mode):
    """config switch EMB_FWD_ROWS (csrc/misc.hip embedding_fwd_k / csrc/conv_bf16.hip embedding_fwd_bf16_k row forms): the
    same gather + counter dropout without a 64-bit division per element -- bit-identical rows (net/multimodal_context_net_v2.py:70-78)."""
    from speech2affective_gestures_amd import bf16, config, noise, ops
    g = torch.Generator().manual_seed(10)
    table = torch.randn(400, 300, generator=g).cuda()
    ids = torch.randint(0, 400, (37, 34), generator=g).cuda()
    ids[:, 25:] = 0
    st = torch.tensor([77, 3], dtype=torch.int64, device='cuda')
    outs = []
    for rows in (False, True):
        with config.override('EMB_FWD_ROWS', rows):
            y = (ops if mode == 'fp32' else bf16).embedding(ids, table, 0.1, st, 5)
            torch.cuda.synchronize()
            outs.append(y.detach().clone())
    assert outs[0].shape[:2] == (37, 34) and torch.equal(outs[0], outs[1])
    assert float(outs[0].float().abs().sum()) > 0


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
@pytest.mark.parametrize('B,T', [(1, 9), (3, 33), (2, 40), (7, 21)])
def test_gather_and_deep_rings_at_other_clip_lengths(mode, B, T):
    """The opt-in TCN paths away from the one geometry everything else uses (34 frames): 9 / 21 / 33 / 40 frames (40 = the
    clip-resident kernels' limit), single clips and ragged batches -- gather inside the launch + deep weight rings against the
    default launches: outputs bit-identical, gradients 1e-5."""
    from speech2affective_gestures_amd import bf16, config, noise, ops
    txt = _encoder()
    g = torch.Generator().manual_seed(12 + T)
    ids = torch.randint(0, 400, (B, T), generator=g).cuda()
    ids[:, T // 2:] = 0
    dt = torch.randn(B, T, 32, generator=g).cuda()
    res = {}
    prev = (ops.TCN32_GATHER, bf16.TCN_GATHER)
    try:
        for on in (False, True):
            ops.TCN32_GATHER = bf16.TCN_GATHER = on
            with config.override('TCN_RING_DEEP', on), bf16.precision(mode):
                for p in txt.parameters():
                    p.grad = None
                ops.begin_step()
                noise.manual_seed(13)
                t = txt(ids)[0]
                (t * dt).sum().backward()
                torch.cuda.synchronize()
                res[on] = (t.detach().clone(), {k: p.grad.clone() for k, p in txt.named_parameters()})
    finally:
        ops.TCN32_GATHER, bf16.TCN_GATHER = prev
    (t0, g0), (t1, g1) = res[False], res[True]
    assert t0.shape == (B, T, 32) and torch.equal(t1, t0)
    for k in g0:
        assert rel(g1[k], g0[k]) < 1e-5, (k, rel(g1[k], g0[k]))
