"""Head of the wave encoder without conv1's output in HBM (csrc/wave12.hip): Conv1d(1,16,15,s5,p1600) BatchNorm LeakyReLU(0.3)
Conv1d(16,32,15,s6) of net/multimodal_context_net_v2.py:18-21.  Every launch against torch on the CPU (float64, autograd through
F.conv1d / training-mode batch norm), in both modes: fp32 (forward on the f32 MFMA: 2e-6 of the largest element; backward products
from two bf16 pieces per operand, 16 mantissa bits: 5e-5) and bf16 (z1 / a1 / z2 rounded to bf16: tolerances of 8-mantissa-bit
storage, stated per assertion), then the WavEncoder module with and without it."""
import ctypes as C
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

GEOMS = [(2, 36267), (3, 3000), (5, 1000), (1, 517)]          # (clips, samples): the TED clip length, and ragged ones


def r16(t):
    return t.to(torch.bfloat16).to(t.dtype)


def ste16(t):
    """bf16 rounding with a straight-through gradient (the kernels round stored values; derivatives ignore the rounding)"""
    return t + (r16(t.detach()) - t.detach())


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / max(1e-9, float(b.abs().max())))


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _problem(N, Lin, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(N, Lin, generator=g) * 0.05).clamp(-1, 1)
    w1 = torch.randn(16, 1, 15, generator=g) / math.sqrt(15)
    b1 = torch.randn(16, generator=g) * 0.1
    g1, e1 = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.3
    w2 = torch.randn(32, 16, 15, generator=g) / math.sqrt(240)
    b2 = torch.randn(32, generator=g) * 0.1
    return x, w1, b1, g1, e1, w2, b2


def ref_head(x, w1, b1, g1, e1, w2, b2, bf, eps=1e-5, slope=0.3):
    """float64 reference; ``bf``: the roundings of bf16 mode (straight-through)."""
    z1 = F.conv1d(x.unsqueeze(1), w1, b1, stride=5, padding=1600)                 # (N, 16, L1)
    if bf:
        z1 = ste16(z1)
    mean = z1.mean(dim=(0, 2))
    var = z1.var(dim=(0, 2), unbiased=False)
    inv = (var + eps).rsqrt()
    t = (z1 - mean[None, :, None]) * (inv * g1)[None, :, None] + e1[None, :, None]
    a1 = torch.where(t > 0, t, slope * t)
    if bf:
        a1 = ste16(a1)
    z2 = F.conv1d(a1, r16(w2) if bf else w2, b2, stride=6)                       # (N, 32, L2)
    return z1, mean, var, inv, z2


class _BN:
    def __init__(self, C_, dev):
        self.running_mean = torch.zeros(C_, device=dev)
        self.running_var = torch.ones(C_, device=dev)
        self.num_batches_tracked = torch.zeros((), dtype=torch.int64, device=dev)
        self.eps, self.momentum = 1e-5, 0.1


@pytest.mark.parametrize('N,Lin', GEOMS)
@pytest.mark.parametrize('bf', [False, True])
def test_statistics_and_forward(N, Lin, bf):
    from speech2affective_gestures_amd import wave12
    x, w1, b1, g1, e1, w2, b2 = _problem(N, Lin, 11 + N)
    xd = x.double()
    z1, mean, var, inv, z2 = ref_head(xd, w1.double(), b1.double(), g1.double(), e1.double(), w2.double(), b2.double(), bf)
    dev = 'cuda'
    pk = wave12.packed_weights(w1.to(dev), w2.to(dev))
    bn1, bn2 = _BN(16, dev), _BN(32, dev)
    xc, b1c, g1c, e1c, b2c = x.to(dev), b1.to(dev), g1.to(dev), e1.to(dev), b2.to(dev)
    coef1 = wave12.stats(xc, pk, b1c, bn1, g1c, e1c, bf)
    torch.cuda.synchronize()
    # BatchNorm 1: mean / invstd / scale / shift and the running estimates (one update)
    n1 = z1.shape[0] * z1.shape[2]
    assert torch.allclose(coef1[2].cpu().double(), mean, rtol=1e-4, atol=2e-6)
    assert torch.allclose(coef1[3].cpu().double(), inv, rtol=1e-4)
    assert torch.allclose(coef1[0].cpu().double(), g1.double() * inv, rtol=1e-4)
    assert torch.allclose(coef1[1].cpu().double(), e1.double() - mean * g1.double() * inv, rtol=1e-4, atol=1e-5)
    assert torch.allclose(bn1.running_mean.cpu().double(), 0.1 * mean, rtol=1e-4, atol=1e-6)
    assert torch.allclose(bn1.running_var.cpu().double(), 0.9 + 0.1 * var * n1 / (n1 - 1), rtol=1e-4)
    assert int(bn1.num_batches_tracked) == 1
    g2, e2 = torch.rand(32) + 0.5, torch.randn(32)
    g2c, e2c = g2.to(dev), e2.to(dev)
    out, part, prow, coef2 = wave12.forward(xc, pk, b1c, coef1, 0.3, b2c, not bf, fold=(bn2, g2c, e2c))
    torch.cuda.synchronize()
    want = z2.transpose(1, 2).contiguous()
    assert out.dtype == (torch.bfloat16 if bf else torch.float32)
    # bf16 mode: a1 is rounded where the kernel's z1 differs from torch's in its last bits, then 240 bf16 products
    assert rel(out.float(), want) < (1.2e-2 if bf else 1e-5), rel(out.float(), want)
    # the column sums are those of the STORED tensor, exactly; their fold == training-mode batch-norm statistics of it
    yd = out.double().reshape(-1, 32).cpu()
    st = part[:2 * prow * 32].view(2, prow, 32).cpu()
    assert torch.allclose(st[0].sum(0), yd.sum(0), rtol=1e-12, atol=1e-9)
    assert torch.allclose(st[1].sum(0), (yd * yd).sum(0), rtol=1e-12, atol=1e-9)
    m2, v2 = yd.mean(0), yd.var(0, unbiased=False)
    assert torch.allclose(coef2[2].cpu().double(), m2, rtol=1e-5, atol=1e-7)
    assert torch.allclose(coef2[3].cpu().double(), (v2 + 1e-5).rsqrt(), rtol=1e-5)
    assert torch.allclose(coef2[0].cpu().double(), g2.double() * (v2 + 1e-5).rsqrt(), rtol=1e-5)
    assert int(bn2.num_batches_tracked) == 1


@pytest.mark.parametrize('N,Lin,cap', [g + (0,) for g in GEOMS] + [(3, 3000, 4), (5, 1000, 3), (4, 36267, 5)])
@pytest.mark.parametrize('bf', [False, True])
@pytest.mark.parametrize('slope', [1.0, 0.3])
def test_backward(N, Lin, cap, bf, slope):
    """s2ag_wave12_bwd: gradients of both conv weights and BatchNorm 1's gamma / beta from dy2, against autograd through the
    float64 reference; accumulating into non-zero slots.  slope = 1 (no kink) pins the linear algebra at the precision of the
    products; with the LeakyReLU's kink a pre-activation within rounding distance of zero takes the other branch in one of
    the two computations (~1e-5 of the elements in fp32 mode) and moves a column sum by 0.7 of ONE element's gradient: the
    tolerance there is that of a few such elements, not of the arithmetic.  ``cap`` > 0: that many workgroups walk all the steps
    (ring of prefetched steps, clip boundaries inside a workgroup's range -- what a full-size batch does on 512 of them)."""
    from speech2affective_gestures_amd import _lib as L
    from speech2affective_gestures_amd import wave12
    x, w1, b1, g1, e1, w2, b2 = _problem(N, Lin, 23 + N)
    g = torch.Generator().manual_seed(5 + N)
    leaves = [t.double().requires_grad_(True) for t in (w1, b1, g1, e1, w2, b2)]
    z1, mean, var, inv, z2 = ref_head(x.double(), *leaves, bf, slope=slope)
    L2 = z2.shape[2]
    dev = 'cuda'
    if bf:      # the operands of dy2 = ca dz + cc z2 + cb as wave_fused.hip's data gradient of conv3 leaves them
        dz = r16(torch.randn(N, L2, 32, generator=g))
        zq = r16(z2.detach().transpose(1, 2).float().contiguous())
        cabc = torch.stack([torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.05,
                            torch.randn(32, generator=g) * 0.1])
        dy2 = cabc[0] * dz + (cabc[2] * zq + cabc[1])
    else:
        dy2 = torch.randn(N, L2, 32, generator=g)
    (z2 * dy2.double().transpose(1, 2)).sum().backward()
    gw1, gb1, gg1, ge1, gw2, gb2 = [t.grad for t in leaves]
    assert float(gb1.abs().max()) < 1e-9 * float(dy2.abs().sum())            # identically zero behind a BatchNorm (not formed)

    pk = wave12.packed_weights(w1.to(dev), w2.to(dev))
    bn1 = _BN(16, dev)
    xc, b1c, g1c, e1c = x.to(dev), b1.to(dev), g1.to(dev), e1.to(dev)
    coef1 = wave12.stats(xc, pk, b1c, bn1, g1c, e1c, bf)
    init = {k: torch.randn(s, generator=g) for k, s in (('w1', (16, 1, 15)), ('g1', (16,)), ('e1', (16,)), ('w2', (32, 16, 15)))}
    slots = {k: v.to(dev) for k, v in init.items()}
    prev = L.load().s2ag_wave12_set_bwd_block_cap(cap)
    try:
        if bf:
            cabc1 = wave12.backward(xc, pk, b1c, coef1, g1c, slope, dz.to(torch.bfloat16).to(dev), zq.to(torch.bfloat16).to(dev),
                                    cabc.to(dev), slots)
        else:
            cabc1 = wave12.backward(xc, pk, b1c, coef1, g1c, slope, dy2.to(dev), None, None, slots)
        torch.cuda.synchronize()
    finally:
        L.load().s2ag_wave12_set_bwd_block_cap(prev)
    got = {k: slots[k].cpu() - init[k] for k in init}
    tol = 1e-2 if bf else (5e-5 if slope == 1.0 else 5e-3)
    errs = {'w2': rel(got['w2'], gw2), 'g1': rel(got['g1'], gg1), 'e1': rel(got['e1'], ge1), 'w1': rel(got['w1'], gw1)}
    print(f'[wave12 bwd N={N} Lin={Lin} cap={cap} bf={bf} slope={slope}]', {k: f'{v:.2e}' for k, v in errs.items()})
    for k, v in errs.items():
        assert v < tol, (k, v)
    # the coefficients of dz1 = ca du1 + cc z1 + cb: ca = gamma invstd
    assert torch.allclose(cabc1[0].cpu().double(), g1.double() * inv.detach(), rtol=1e-4)


@pytest.mark.parametrize('B', [3, 40])
def test_wave_encoder_fp32_with_and_without_the_fused_head(B):
    """WavEncoder (train mode, fp32): the fused head against the layer-by-layer kernels -- output, every parameter gradient,
    running statistics."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import s2ag_oracle as O
    from speech2affective_gestures_amd import ops, wave12
    from speech2affective_gestures_amd.net.multimodal_context_net_v2 import WavEncoder
    inp = O.recipe_inputs(B, 34, 77, 500, 12)
    res, state = {}, None
    prev = wave12.ENABLED
    try:
        for on in (False, True):
            torch.manual_seed(1)
            wav = WavEncoder().cuda().train()
            with torch.no_grad():
                for i in (1, 4, 7):
                    wav.feat_extractor[i].weight.uniform_(0.5, 1.5)
                    wav.feat_extractor[i].bias.normal_(0, 0.3)
            if state is None:
                state = {k: v.clone() for k, v in wav.state_dict().items()}
            wav.load_state_dict(state)
            wave12.ENABLED = on
            ops.begin_step()
            a = wav(inp['in_audio'].cuda())
            da = torch.randn(a.shape, generator=torch.Generator().manual_seed(2)).cuda()
            (a * da).sum().backward()
            torch.cuda.synchronize()
            res[on] = (a.detach().clone(), {k: p.grad.clone() for k, p in wav.named_parameters()},
                       {k: v.clone() for k, v in wav.state_dict().items() if 'running' in k or 'tracked' in k})
    finally:
        wave12.ENABLED = prev
    (a0, g0, s0), (a1, g1, s1) = res[False], res[True]
    assert rel(a1, a0) < 1e-4
    for k in s0:
        assert rel(s1[k].float(), s0[k].float()) < 1e-4, k
    dead = ('feat_extractor.0.bias', 'feat_extractor.3.bias', 'feat_extractor.6.bias')     # a BatchNorm cancels them
    for k in g0:
        if k in dead:
            assert float(g1[k].abs().max()) <= float(g0[k].abs().max()) + 1e-3 * float(g0['feat_extractor.9.bias'].abs().max()), k
        else:
            # LeakyReLU pre-activations within rounding distance of zero take the other branch in one of the two runs (see
            # test_backward; ~1e-5 of 1.7 M elements behind BatchNorm 2 at B = 40), each moving one row of a weight gradient
            # by ~0.3 % of its typical element: sparse, so the L2 criterion is the tight one (a lost step of 32 frames would
            # be 2.5 % in both)
            a, b = g1[k].double().cpu(), g0[k].double().cpu()
            l2 = float((a - b).norm() / b.norm())
            assert l2 < 4e-3 and rel(g1[k], g0[k]) < 2e-2, (k, l2, rel(g1[k], g0[k]))
