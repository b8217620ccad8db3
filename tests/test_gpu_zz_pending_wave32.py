"""fp32 tail of the wave encoder with BatchNorm folded into conv3 / conv4 (csrc/wave_fused.hip, the *32 entry points;
speech2affective_gestures_amd/wave32.py) -- feat_extractor.{4..9} of net/multimodal_context_net_v2.py:22-27.

The path is opt-in (config switch WAVE_TAIL32) until it has been run on a GPU.  The tests arm it themselves (wave32.ENABLED /
direct calls) and live in a file that sorts LAST, so that a hardware surprise in a never-run kernel cannot stop a `pytest -x`
run before the suites of the default path; on the CPU device model (tests/emu) they pass.  What they check: each launch against torch on the CPU in float64
(forward on the f32 matrix pipe: 1e-5 of the largest element; gradients from two bf16 pieces per operand, 16 mantissa bits:
1e-4 without a kink, 5e-3 with the LeakyReLU's kink -- see tests/test_gpu_wave12.py::test_backward for why), then the
WavEncoder module against the layer-by-layer kernels."""
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / max(1e-9, float(b.abs().max())))


class _BN:
    def __init__(self, C_, dev):
        self.running_mean = torch.zeros(C_, device=dev)
        self.running_var = torch.ones(C_, device=dev)
        self.num_batches_tracked = torch.zeros((), dtype=torch.int64, device=dev)
        self.eps, self.momentum = 1e-5, 0.1


def leaky(t, slope):
    return torch.where(t > 0, t, slope * t)


def _tail_problem(N, L2, seed):
    g = torch.Generator().manual_seed(seed)
    z2 = torch.randn(N, L2, 32, generator=g)
    coef2 = torch.stack([torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.3])   # scale, shift of BatchNorm 2
    w3 = torch.randn(64, 32, 15, generator=g) / math.sqrt(480)
    b3 = torch.randn(64, generator=g) * 0.1
    g3, e3 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    w4 = torch.randn(32, 64, 15, generator=g) / math.sqrt(960)
    b4 = torch.randn(32, generator=g) * 0.1
    return z2, coef2, w3, b3, g3, e3, w4, b4


def ref_tail(z2, coef2, w3, b3, g3, e3, w4, b4, slope, eps=1e-5):
    """float64: (N, L2, 32) raw conv2 output + BatchNorm 2's scale / shift -> z3, BatchNorm 3 statistics, out (channels-last)"""
    a2 = leaky(z2 * coef2[0] + coef2[1], slope).transpose(1, 2)
    z3 = F.conv1d(a2, w3, b3, stride=6)
    mean = z3.mean(dim=(0, 2))
    var = z3.var(dim=(0, 2), unbiased=False)
    inv = (var + eps).rsqrt()
    a3 = leaky((z3 - mean[None, :, None]) * (inv * g3)[None, :, None] + e3[None, :, None], slope)
    out = F.conv1d(a3, w4, b4, stride=6)
    return z3.transpose(1, 2), mean, var, inv, out.transpose(1, 2)


@pytest.mark.parametrize('N,L2', [(2, 1313), (3, 231), (5, 105), (1, 99), (40, 1313)])
def test_forward_of_conv3_and_conv4(N, L2):
    """s2ag_wave_conv_fwd32 twice: z3 + BatchNorm 3's fold (running estimates included), then out."""
    from speech2affective_gestures_amd import wave32
    z2, coef2, w3, b3, g3, e3, w4, b4 = _tail_problem(N, L2, 3 + N)
    d = [t.double() for t in (z2, coef2, w3, b3, g3, e3, w4, b4)]
    z3, mean, var, inv, out = ref_tail(*d, 0.3)
    dev = 'cuda'
    pk = wave32.packed_tail(w3.to(dev), w4.to(dev))
    (k3, _), (k4, _) = wave32.pack_views(pk)
    bn3 = _BN(64, dev)
    y3, coef3 = wave32.conv_fwd32(z2.to(dev), coef2.to(dev), 0.3, k3, b3.to(dev), 32, 64, fold=(bn3, g3.to(dev), e3.to(dev)))
    y4, _ = wave32.conv_fwd32(y3, coef3, 0.3, k4, b4.to(dev), 64, 32)
    torch.cuda.synchronize()
    assert rel(y3, z3) < 1e-5
    rows = z3.shape[0] * z3.shape[1]
    assert torch.allclose(coef3[2].cpu().double(), mean, atol=1e-6 * float(z3.abs().max()))
    assert torch.allclose(coef3[3].cpu().double(), inv, rtol=1e-5)
    assert torch.allclose(coef3[0].cpu().double(), g3.double() * inv, rtol=1e-5)
    assert torch.allclose(bn3.running_mean.cpu().double(), 0.1 * mean, atol=1e-6)
    assert torch.allclose(bn3.running_var.cpu().double(), 0.9 + 0.1 * var * rows / max(1, rows - 1), rtol=1e-5)
    assert int(bn3.num_batches_tracked) == 1
    assert rel(y4, out) < 2e-5


@pytest.mark.parametrize('N,L2', [(2, 1313), (3, 231), (5, 105), (1, 99), (24, 1313)])
@pytest.mark.parametrize('slope', [1.0, 0.3])
def test_backward_of_conv4_and_conv3(N, L2, slope):
    """s2ag_wave_conv_wgrad32 / _dgrad32 for both layers: gradients of w3, b3, gamma3, beta3, w4, b4 and of z2 (through
    BatchNorm 2's LeakyReLU: what the head's backward receives as the operands of dy2) against autograd in float64."""
    import ctypes as C
    from speech2affective_gestures_amd import _lib as L
    from speech2affective_gestures_amd import ops, wave32
    lib = L.load()
    z2, coef2, w3, b3, g3, e3, w4, b4 = _tail_problem(N, L2, 13 + N)
    gen = torch.Generator().manual_seed(7 + N)
    leaves = [t.double().requires_grad_(True) for t in (w3, b3, g3, e3, w4, b4)]
    a2pre = (z2.double() * coef2[0].double() + coef2[1].double()).requires_grad_(True)     # gradient w.r.t. it = dz2's definition
    a2 = leaky(a2pre, slope).transpose(1, 2)
    z3 = F.conv1d(a2, leaves[0], leaves[1], stride=6)
    mean = z3.mean(dim=(0, 2))
    var = z3.var(dim=(0, 2), unbiased=False)
    inv = (var + 1e-5).rsqrt()
    a3 = leaky((z3 - mean[None, :, None]) * (inv * leaves[2])[None, :, None] + leaves[3][None, :, None], slope)
    out = F.conv1d(a3, leaves[4], leaves[5], stride=6)
    gout = torch.randn(out.shape[0], out.shape[2], 32, generator=gen)
    (out * gout.double().transpose(1, 2)).sum().backward()
    gw3, gb3, gg3, ge3, gw4, gb4 = [t.grad for t in leaves]
    dz2_ref = a2pre.grad                                   # = da2 * leaky'(scale2 z2 + shift2)

    dev = 'cuda'
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    pk = wave32.packed_tail(w3.to(dev), w4.to(dev))
    (k3, p3), (k4, p4) = wave32.pack_views(pk)
    z2c, c2 = z2.to(dev), coef2.to(dev)
    # BatchNorm 2's mean / invstd only enter the sums of dz2 * xhat2 (gamma2's gradient, not checked here): any values do
    cp2 = torch.stack([c2[0], c2[1], torch.zeros(32, device=dev), torch.ones(32, device=dev)])
    bn3 = _BN(64, dev)
    y3, coef3 = wave32.conv_fwd32(z2c, c2, slope, k3, b3.to(dev), 32, 64, fold=(bn3, g3.to(dev), e3.to(dev)))
    init = {k: torch.randn(v.shape, generator=gen) for k, v in (('w3', w3), ('b3', b3), ('g3', g3), ('e3', e3), ('w4', w4), ('b4', b4))}
    slots = {k: v.to(dev) for k, v in init.items()}
    L3, L4 = y3.shape[1], gout.shape[1]
    g = gout.to(dev)

    def wgrad(dz, yy, cabc, yp, cp, Lin, Lout, ci, co, wk, bk):
        nb = lib.s2ag_wave_wgrad_blocks(N, Lout, ci, co)
        part = torch.empty(nb * co * 15 * ci + nb * co, device=dev)
        ca, cb, cc = (None, None, None) if cabc is None else (cabc[0], cabc[1], cabc[2])
        L.check(lib.s2ag_wave_conv_wgrad32(p(dz), p(yy), p(ca), p(cb), p(cc), int(cabc is None), p(yp), p(cp[0]), p(cp[1]), slope,
                                           p(part), p(part[nb * co * 15 * ci:]), p(slots[wk]), p(slots[bk]), N, Lin, Lout, ci, co, s),
                'wgrad32')
        return part

    def dgrad(dz, yy, cabc, wph, yp, cp, gamma, gk, ek, Lin, Lout, ci, co):
        prow = lib.s2ag_wave_dgrad_rows(N, Lin, ci)
        st = torch.empty(2 * (prow + (prow + 15) // 16) * ci, dtype=torch.float64, device=dev)
        dzp = torch.empty(N, Lin, ci, device=dev)
        out_c = torch.empty(3, ci, device=dev)
        ca, cb, cc = (None, None, None) if cabc is None else (cabc[0], cabc[1], cabc[2])
        L.check(lib.s2ag_wave_conv_dgrad32(p(dz), p(yy), p(ca), p(cb), p(cc), int(cabc is None), p(wph), p(yp), p(cp[0]), p(cp[1]),
                                           p(cp[2]), p(cp[3]), slope, p(dzp), p(st), ops._tickets(torch.device(dev), 1 + (prow + 15) // 16),
                                           p(gamma), p(slots[gk]) if gk else None, p(slots[ek]) if ek else None, p(out_c[0]),
                                           p(out_c[1]), p(out_c[2]), N, Lin, Lout, ci, co, s), 'dgrad32')
        return dzp, out_c, st
    keep = [wgrad(g, None, None, y3, coef3, L3, L4, 64, 32, 'w4', 'b4')]
    dz3, cabc3, st3 = dgrad(g, None, None, p4, y3, coef3, g3.to(dev), 'g3', 'e3', L3, L4, 64, 32)
    keep.append(wgrad(dz3, y3, cabc3, z2c, cp2, L2, L3, 32, 64, 'w3', 'b3'))
    gamma2 = torch.ones(32, device=dev)
    dz2, _, st2 = dgrad(dz3, y3, cabc3, p3, z2c, cp2, gamma2, None, None, L2, L3, 32, 64)
    torch.cuda.synchronize()
    got = {k: slots[k].cpu() - init[k] for k in init}
    tol = 1e-4 if slope == 1.0 else 5e-3
    errs = {'w4': rel(got['w4'], gw4), 'b4': rel(got['b4'], gb4), 'g3': rel(got['g3'], gg3), 'e3': rel(got['e3'], ge3),
            'w3': rel(got['w3'], gw3), 'dz2': rel(dz2, dz2_ref)}
    print(f'[wave32 bwd N={N} L2={L2} slope={slope}]', {k: f'{v:.2e}' for k, v in errs.items()})
    for k, v in errs.items():
        assert v < tol, (k, v)
    # conv3's bias feeds BatchNorm 3: its gradient is zero up to the rounding of ~N L3 terms
    assert float(got['b3'].abs().max()) < 1e-3 * float(gb4.abs().max()) + 1e-4
    assert float(gb3.abs().max()) < 1e-9 * float(gout.abs().sum())
    assert torch.allclose(cabc3[0].cpu().double(), g3.double() * inv.detach(), rtol=1e-4)


@pytest.mark.parametrize('B', [3, 40])
def test_wave_encoder_fp32_fully_folded_against_layer_by_layer(B):
    """WavEncoder (train mode, fp32): wave32.encoder_f32 against the default path (fused head + layer-by-layer tail) -- output,
    every parameter gradient, running statistics."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import s2ag_oracle as O
    from speech2affective_gestures_amd import ops, wave32
    from speech2affective_gestures_amd.net.multimodal_context_net_v2 import WavEncoder
    inp = O.recipe_inputs(B, 34, 77, 500, 12)
    res, state = {}, None
    prev = wave32.ENABLED
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
            wave32.ENABLED = on
            ops.begin_step()
            a = wav(inp['in_audio'].cuda())
            assert (type(a.grad_fn).__name__.startswith('_WaveFused32')) == on
            da = torch.randn(a.shape, generator=torch.Generator().manual_seed(2)).cuda()
            (a * da).sum().backward()
            torch.cuda.synchronize()
            res[on] = (a.detach().clone(), {k: p.grad.clone() for k, p in wav.named_parameters()},
                       {k: v.clone() for k, v in wav.state_dict().items() if 'running' in k or 'tracked' in k})
    finally:
        wave32.ENABLED = prev
    (a0, g0, s0), (a1, g1, s1) = res[False], res[True]
    assert rel(a1, a0) < 1e-4
    for k in s0:
        assert rel(s1[k].float(), s0[k].float()) < 1e-4, k
    dead = ('feat_extractor.0.bias', 'feat_extractor.3.bias', 'feat_extractor.6.bias')     # a BatchNorm cancels them
    for k in g0:
        if k in dead:
            assert float(g1[k].abs().max()) <= float(g0[k].abs().max()) + 1e-3 * float(g0['feat_extractor.9.bias'].abs().max()), k
        else:
            a, b = g1[k].double().cpu(), g0[k].double().cpu()       # kink flips: sparse, see tests/test_gpu_wave12.py
            l2 = float((a - b).norm() / b.norm())
            assert l2 < 4e-3 and rel(g1[k], g0[k]) < 2e-2, (k, l2, rel(g1[k], g0[k]))


@pytest.mark.parametrize('N,Lin', [(2, 36267), (3, 3000), (5, 1000)])
def test_pipelined_fp32_forward_is_bit_identical(N, Lin):
    """config switch W12_FWD_PIPE (csrc/wave12.hip wv12_fwd_k<2, true>): the head's fp32 forward with the next B operands
    requested before the current products -- same products, same order, same accumulators: z2 and the BatchNorm partial
    sums must be bit-identical to the default launch (net/multimodal_context_net_v2.py:18-21)."""
    from speech2affective_gestures_amd import config, wave12
    g = torch.Generator().manual_seed(21 + N)
    x = ((torch.randn(N, Lin, generator=g) * 0.05).clamp(-1, 1)).cuda()
    w1 = (torch.randn(16, 1, 15, generator=g) / math.sqrt(15)).cuda()
    b1 = (torch.randn(16, generator=g) * 0.1).cuda()
    g1, e1 = (torch.rand(16, generator=g) + 0.5).cuda(), (torch.randn(16, generator=g) * 0.3).cuda()
    w2 = (torch.randn(32, 16, 15, generator=g) / math.sqrt(240)).cuda()
    b2 = (torch.randn(32, generator=g) * 0.1).cuda()
    pk = wave12.packed_weights(w1, w2)
    coef1 = wave12.stats(x, pk, b1, _BN(16, 'cuda'), g1, e1, False)
    outs = []
    for pipe in (False, True):
        with config.override('W12_FWD_PIPE', pipe):
            z2, part, prow, _ = wave12.forward(x, pk, b1, coef1, 0.3, b2, True)
            torch.cuda.synchronize()
            outs.append((z2.clone(), part[:2 * prow * 32].clone()))
    assert torch.isfinite(outs[0][0]).all() and float(outs[0][0].abs().sum()) > 0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
