"""The wave encoder with BatchNorm folded into the neighbouring convs (csrc/wave_fused.hip): every kernel against torch fp32
on the SAME bf16-rounded operands and transform arithmetic, then the whole encoder (forward, every parameter gradient,
running statistics) against the oracle's WavEncoder (net/multimodal_context_net_v2.py:14-33 restated) -- tolerances are
those of 8-mantissa-bit storage, stated per assertion."""
import ctypes
import ctypes as C
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(16, 32), (32, 64), (64, 32)]


def r16(t):
    return t.to(torch.bfloat16).float()


def rel(a, b):
    a, b = a.detach().float().cpu().double(), b.detach().float().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / max(1e-6, float(b.abs().max())))


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pack_fwd(w, KP):
    """(Cout, Cin, 15) fp32 -> (Cout, KP) bf16, k = tap * Cin + ci (the layout bf16.WeightPack makes for these layers)"""
    Cout, Cin, ks = w.shape
    out = torch.zeros(Cout, KP)
    out[:, :ks * Cin] = w.permute(0, 2, 1).reshape(Cout, ks * Cin)
    return out.to(torch.bfloat16).cuda()


def lrelu(x, s):
    return torch.where(x > 0, x, x * s)


@pytest.mark.parametrize('Cin,Cout', SHAPES)
@pytest.mark.parametrize('N,Lout', [(3, 150), (2, 34), (5, 17)])
def test_fused_forward_conv(Cin, Cout, N, Lout):
    """s2ag_wave_conv_fwd: a = leaky(scale * x + shift) in the loader, flat-window conv on the bf16 pipe, raw output +
    fp64 column sums.  Against F.conv1d on the bf16-rounded a: bf16 outputs within 2^-8 of the largest element."""
    from speech2affective_gestures_amd import _lib as L
    lib = L.load()
    Lin = (Lout - 1) * 6 + 15 + (N % 3)          # a few unused trailing frames, as in the encoder
    g = torch.Generator().manual_seed(Cin * 100 + N)
    x = r16(torch.randn(N, Lin, Cin, generator=g))
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    w = torch.randn(Cout, Cin, 15, generator=g) / math.sqrt(Cin * 15)
    b = torch.randn(Cout, generator=g) * 0.1
    out_f32 = Cout == 32 and Cin == 64
    KP = (15 * Cin + 63) // 64 * 64
    a = r16(lrelu(torch.addcmul(sh, sc, x), 0.3))
    want = F.conv1d(a.transpose(1, 2), r16(w), b, stride=6).transpose(1, 2).contiguous()
    xd = x.to(torch.bfloat16).cuda()
    y = torch.empty(N, Lout, Cout, dtype=torch.float32 if out_f32 else torch.bfloat16, device='cuda')
    rows = lib.s2ag_wave_fwd_rows(N, Lout, Cin, Cout)
    ng = (rows + 15) // 16                       # group rows of the two-level fold behind the partial rows
    stats_buf = None if out_f32 else torch.full((2 * (rows + ng) * Cout,), float('nan'), dtype=torch.float64, device='cuda')
    stats = None if out_f32 else stats_buf[:2 * rows * Cout].view(2, rows, Cout)
    scd, shd, wd, bd = sc.cuda(), sh.cuda(), pack_fwd(w, KP), b.cuda()      # (keep the device tensors alive over the launch)
    # with a fold: the BatchNorm behind this conv gets its coefficients + running estimates from the last workgroup
    gamma, beta = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
    rm, rv, nbt = torch.zeros(Cout, device='cuda'), torch.ones(Cout, device='cuda'), torch.zeros((), dtype=torch.int64, device='cuda')
    coef, ticket = torch.full((4, Cout), float('nan'), device='cuda'), torch.zeros(1 + ng, dtype=torch.int32, device='cuda')
    fa = None if out_f32 else L.BnFoldArgs(_p(ticket), _p(gamma), _p(beta), _p(rm), _p(rv), _p(nbt), 1e-5, 0.1, 2, _p(coef[0]),
                                           _p(coef[1]), _p(coef[2]), _p(coef[3]))
    L.check(lib.s2ag_wave_conv_fwd(_p(xd), _p(scd), _p(shd), 0.3, _p(wd), KP, _p(bd), _p(y), int(out_f32), _p(stats_buf),
                                   C.byref(fa) if fa is not None else None, N, Lin, Lout, Cin, Cout, _stream()), 'wave_conv_fwd')
    torch.cuda.synchronize()
    assert rel(y, want) < (2e-5 if out_f32 else 6e-3)
    if stats is not None:       # column sums of the ROUNDED outputs, exactly
        yd = y.double().reshape(-1, Cout)
        assert torch.allclose(stats[0].sum(0), yd.sum(0), rtol=1e-12, atol=1e-9)
        assert torch.allclose(stats[1].sum(0), (yd * yd).sum(0), rtol=1e-12, atol=1e-9)
        # the fold == training-mode F.batch_norm statistics of the stored tensor, two running-estimate updates (repeat = 2)
        mean, var = yd.mean(0), yd.var(0, unbiased=False)
        invstd = (var + 1e-5).rsqrt()
        assert int(ticket.abs().sum()) == 0 and int(nbt) == 2
        assert torch.allclose(coef[2].double(), mean, rtol=1e-6, atol=1e-7) and torch.allclose(coef[3].double(), invstd, rtol=1e-6)
        assert torch.allclose(coef[0].double(), gamma.double() * invstd, rtol=1e-6)
        assert torch.allclose(coef[1].double(), beta.double() - mean * gamma.double() * invstd, rtol=1e-5, atol=1e-6)
        unb = yd.var(0, unbiased=True)
        assert torch.allclose(rm.double(), 0.19 * mean, rtol=1e-5, atol=1e-7)            # 0 -> 0.1 m -> 0.19 m
        assert torch.allclose(rv.double(), 0.81 + 0.19 * unb, rtol=1e-5)


def pack_phases(w, CPO):
    """(Cout, Cin, 15) -> (6, Cin, 3, CPO) bf16: [r][ci][i][co] = w[co][ci][r + 6 i] (bf16.WeightPack 'phases' layout)"""
    Cout, Cin, ks = w.shape
    out = torch.zeros(6, Cin, 3, CPO)
    for r in range(6):
        for i in range(3):
            if r + 6 * i < ks:
                out[r, :, i, :Cout] = w[:, :, r + 6 * i].t()
    return out.to(torch.bfloat16).cuda()


def _bwd_operands(g, N, Lout, Cout, g_f32):
    """(dy as the kernels form it, kernel arguments)"""
    if g_f32:
        gy = torch.randn(N, Lout, Cout, generator=g)
        return r16(gy), dict(dz=gy.cuda(), y=None, ca=None, cb=None, cc=None)
    dz, y = r16(torch.randn(N, Lout, Cout, generator=g)), r16(torch.randn(N, Lout, Cout, generator=g))
    ca, cb, cc = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1, torch.randn(Cout, generator=g) * 0.2
    dy = r16(ca * dz + (cc * y + cb))
    return dy, dict(dz=dz.to(torch.bfloat16).cuda(), y=y.to(torch.bfloat16).cuda(), ca=ca.cuda(), cb=cb.cuda(), cc=cc.cuda())


@pytest.mark.parametrize('Cin,Cout', SHAPES)
@pytest.mark.parametrize('N,Lout', [(3, 150), (2, 34), (5, 17)])
def test_fused_data_gradient(Cin, Cout, N, Lout):
    """s2ag_wave_conv_dgrad: dy = ca dz + cc y + cb in the loader, poly-phase transposed conv, dz_prev = da * leaky'(z) in
    the epilogue + fp64 column sums of dz_prev and dz_prev * xhat from the fp32 values.  Against F.conv_transpose1d."""
    from speech2affective_gestures_amd import _lib as L
    lib = L.load()
    Lin = (Lout - 1) * 6 + 15 + (N % 3)
    g = torch.Generator().manual_seed(Cin * 100 + N + 7)
    g_f32 = Cin == 64
    dy, args = _bwd_operands(g, N, Lout, Cout, g_f32)
    w = torch.randn(Cout, Cin, 15, generator=g) / math.sqrt(Cout * 3)
    yp = r16(torch.randn(N, Lin, Cin, generator=g))
    psc, psh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    pmean, pinv = torch.randn(Cin, generator=g) * 0.2, torch.rand(Cin, generator=g) + 0.5
    da = F.conv_transpose1d(dy.transpose(1, 2), r16(w), stride=6).transpose(1, 2)
    da = F.pad(da, (0, 0, 0, Lin - da.shape[1]))
    z = torch.addcmul(psh, psc, yp)
    want = torch.where(z > 0, da, da * 0.3)
    xhat = yp * pinv - pmean * pinv
    rows = lib.s2ag_wave_dgrad_rows(N, Lin, Cin)
    ng = (rows + 15) // 16
    stats_buf = torch.full((2 * (rows + ng) * Cin,), float('nan'), dtype=torch.float64, device='cuda')
    stats = stats_buf[:2 * rows * Cin].view(2, rows, Cin)
    out = torch.full((N, Lin, Cin), float('nan'), dtype=torch.bfloat16, device='cuda')
    CPO = (Cout + 31) // 32 * 32
    wd, ypd, dv = pack_phases(w, CPO), yp.to(torch.bfloat16).cuda(), [t.cuda() for t in (psc, psh, pmean, pinv)]
    # ... and, given a ticket word, the fold of those sums by the workgroup that finishes last
    gamma = torch.rand(Cin, generator=g) + 0.5
    gd, ticket, dgb, coef = gamma.cuda(), torch.zeros(1 + ng, dtype=torch.int32, device='cuda'), torch.zeros(2, Cin, device='cuda'), \
        torch.full((3, Cin), float('nan'), device='cuda')
    L.check(lib.s2ag_wave_conv_dgrad(_p(args['dz']), _p(args['y']), _p(args['ca']), _p(args['cb']), _p(args['cc']), int(g_f32),
                                     _p(wd), CPO, _p(ypd), _p(dv[0]), _p(dv[1]), _p(dv[2]), _p(dv[3]), 0.3, _p(out), _p(stats_buf),
                                     _p(ticket), _p(gd), _p(dgb[0]), _p(dgb[1]), _p(coef[0]), _p(coef[1]), _p(coef[2]),
                                     N, Lin, Lout, Cin, Cout, _stream()), 'wave_conv_dgrad')
    torch.cuda.synchronize()
    assert int(ticket.abs().sum()) == 0                                      # re-armed
    m1, m2 = stats[0].sum(0).cpu() / (N * Lin), stats[1].sum(0).cpu() / (N * Lin)
    gm, rr, mu = gamma.double(), pinv.double(), pmean.double()
    assert torch.allclose(dgb[0].cpu().double(), stats[1].sum(0).cpu(), rtol=1e-5, atol=1e-6)      # dgamma = sum dz xhat
    assert torch.allclose(dgb[1].cpu().double(), stats[0].sum(0).cpu(), rtol=1e-5, atol=1e-6)      # dbeta = sum dz
    assert torch.allclose(coef[0].cpu().double(), gm * rr, rtol=1e-6)
    assert torch.allclose(coef[2].cpu().double(), -gm * rr * rr * m2, rtol=1e-5, atol=1e-9)
    assert torch.allclose(coef[1].cpu().double(), gm * rr * (rr * mu * m2 - m1), rtol=1e-5, atol=1e-9)
    assert rel(out, want) < 6e-3
    s1, s2 = stats[0].sum(0).cpu(), stats[1].sum(0).cpu()
    w1, w2 = want.double().reshape(-1, Cin).sum(0), (want.double() * xhat.double()).reshape(-1, Cin).sum(0)
    scale = float(want.double().abs().reshape(-1, Cin).sum(0).max())
    assert float((s1 - w1).abs().max()) < 2e-4 * scale and float((s2 - w2).abs().max()) < 4e-4 * scale


@pytest.mark.parametrize('Cin,Cout', SHAPES)
@pytest.mark.parametrize('N,Lout', [(3, 150), (2, 34), (40, 70)])
def test_fused_weight_gradient(Cin, Cout, N, Lout):
    """s2ag_wave_conv_wgrad: dy formed and a_prev = leaky(psc y_prev + psh) recomputed in the loaders, contraction over
    the output frames through the LDS transpose read, partials + ordered reduce INTO dw / db (accumulating)."""
    from speech2affective_gestures_amd import _lib as L
    lib = L.load()
    Lin = (Lout - 1) * 6 + 15 + (N % 3)
    g = torch.Generator().manual_seed(Cin * 100 + N + 13)
    g_f32 = Cin == 64
    dy, args = _bwd_operands(g, N, Lout, Cout, g_f32)
    yp = r16(torch.randn(N, Lin, Cin, generator=g))
    psc, psh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    a = r16(lrelu(torch.addcmul(psh, psc, yp), 0.3))
    want = torch.nn.grad.conv1d_weight(a.transpose(1, 2), (Cout, Cin, 15), dy.transpose(1, 2), stride=6)
    want_b = dy.reshape(-1, Cout).sum(0)
    blocks = lib.s2ag_wave_wgrad_blocks(N, Lout, Cin, Cout)
    part = torch.full((blocks, Cout, 15, Cin), float('nan'), device='cuda')
    part_b = torch.full((blocks, Cout), float('nan'), device='cuda')
    dw0, db0 = torch.randn(Cout, Cin, 15, generator=g), torch.randn(Cout, generator=g)
    dw, db = dw0.cuda(), db0.cuda()
    ypd, pscd, pshd = yp.to(torch.bfloat16).cuda(), psc.cuda(), psh.cuda()
    L.check(lib.s2ag_wave_conv_wgrad(_p(args['dz']), _p(args['y']), _p(args['ca']), _p(args['cb']), _p(args['cc']), int(g_f32),
                                     _p(ypd), _p(pscd), _p(pshd), 0.3, _p(part), _p(part_b), _p(dw), _p(db), N, Lin, Lout, Cin,
                                     Cout, _stream()), 'wave_conv_wgrad')
    torch.cuda.synchronize()
    assert rel(dw.cpu() - dw0, want) < 2e-3 and rel(db.cpu() - db0, want_b) < 2e-3


def test_bn_backward_fold_and_conv1_weight_gradient():
    """s2ag_wave_bn_bwd_fold (gamma / beta gradients + the coefficients of dy = A dz + C y + B) against the closed form, then
    s2ag_wave_conv1_wgrad (conv1's weight gradient from dz_1, y_1 with that transform in its loader) against torch -- and
    the transform itself against autograd through F.batch_norm."""
    from speech2affective_gestures_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(5)
    N, Lin, C = 3, 3000, 16
    Lout = (Lin + 2 * 1600 - 15) // 5 + 1
    x = torch.randn(N, Lin, generator=g) * 0.1
    y = r16(torch.randn(N, Lout, C, generator=g) * 1.5 + 0.3)
    dz = r16(torch.randn(N, Lout, C, generator=g))
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    yf = y.reshape(-1, C).double()
    mean, var = yf.mean(0), yf.var(0, unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    xhat = (yf - mean) * invstd
    R = 7
    idx = torch.arange(yf.shape[0]) % R
    part = torch.zeros(2, R, C, dtype=torch.float64)
    part[0].index_add_(0, idx, dz.reshape(-1, C).double())
    part[1].index_add_(0, idx, dz.reshape(-1, C).double() * xhat)
    dgamma, dbeta = torch.ones(C).cuda(), torch.ones(C).cuda()
    coef = torch.empty(3, C, device='cuda')
    pd, gd, md, sd = part.cuda(), gamma.cuda(), mean.float().cuda(), invstd.float().cuda()
    L.check(lib.s2ag_wave_bn_bwd_fold(_p(pd), R, C, yf.shape[0], _p(gd), _p(md), _p(sd), _p(dgamma), _p(dbeta), _p(coef[0]),
                                      _p(coef[1]), _p(coef[2]), _stream()), 'wave_bn_bwd_fold')
    # autograd through batch_norm: dL/dy for L = sum(dz * bn(y))
    yl = y.reshape(-1, C).double().requires_grad_(True)
    gl, bl = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    (F.batch_norm(yl, None, None, gl, bl, True, 0.1, 1e-5) * dz.reshape(-1, C).double()).sum().backward()
    assert rel(dgamma.cpu() - 1, gl.grad.float()) < 1e-5 and rel(dbeta.cpu() - 1, bl.grad.float()) < 1e-5
    ca, cb, cc = coef.cpu().double()
    dy = ca * dz.reshape(-1, C).double() + cc * yf + cb
    assert rel(dy.float(), yl.grad.float()) < 1e-5
    geom = L.ConvGeom(N, Lin, Lout, 1, 16, 15, 5, 1600, 1, 1, 16, 0)
    dw, db = torch.zeros(16, 1, 15, device='cuda'), torch.zeros(16, device='cuda')
    dzd, yd, xd = dz.to(torch.bfloat16).cuda(), y.to(torch.bfloat16).cuda(), x.cuda()
    part1 = torch.full((lib.s2ag_wave_conv1_wgrad_blocks(ctypes.byref(geom)) * 256,), float('nan'), device='cuda')
    L.check(lib.s2ag_wave_conv1_wgrad(_p(dzd), _p(yd), _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(xd), _p(part1), _p(dw), _p(db),
                                      ctypes.byref(geom), _stream()), 'wave_conv1_wgrad')
    dyf = dy.float().reshape(N, Lout, C)
    want = torch.nn.grad.conv1d_weight(x.unsqueeze(1), (16, 1, 15), dyf.transpose(1, 2), stride=5, padding=1600)
    assert rel(dw.cpu(), want) < 1e-4
    assert float(db.abs().max().cpu()) < 1e-3 * float(dyf.abs().sum(dim=(0, 1)).max())      # sum dy = 0 behind a BatchNorm


@pytest.mark.parametrize('B', [6, 40])
def test_fused_wave_encoder_against_its_fp32_mode_and_the_layer_by_layer_bf16_path(B):
    """WavEncoder (train mode) in bf16 mode with the BatchNorms folded into the convs against (a) the same module in fp32
    mode and (b) the layer-by-layer bf16 path (S2AG_WAVE_FUSED=0): output, every parameter gradient, running statistics.
    The fused backward forms the BatchNorm sums from fp32 accumulators, so gamma / beta gradients -- sums of ~10^6 terms
    that cancel -- no longer carry the rounding of a bf16-stored gradient tensor."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import s2ag_oracle as O
    from speech2affective_gestures_amd import bf16, noise, ops
    from speech2affective_gestures_amd.net.multimodal_context_net_v2 import WavEncoder
    inp = O.recipe_inputs(B, 34, 77, 500, 12)
    res = {}
    prev = bf16.WAVE_FUSED
    state = None
    try:
        for mode, fused in (('fp32', False), ('bf16', False), ('bf16', True)):
            torch.manual_seed(1)
            wav = WavEncoder().cuda().train()
            with torch.no_grad():                                      # BatchNorm affine parameters away from (1, 0)
                for i in (1, 4, 7):
                    wav.feat_extractor[i].weight.uniform_(0.5, 1.5)
                    wav.feat_extractor[i].bias.normal_(0, 0.3)
            if state is None:
                state = {k: v.clone() for k, v in wav.state_dict().items()}
            wav.load_state_dict(state)
            bf16.WAVE_FUSED = fused
            ops.begin_step()
            with bf16.precision(mode):
                a = wav(inp['in_audio'].cuda())
            assert a.dtype == torch.float32 and a.shape == (B, 34, 32)
            da = torch.randn(a.shape, generator=torch.Generator().manual_seed(2)).cuda()
            (a * da).sum().backward()
            torch.cuda.synchronize()
            res[(mode, fused)] = (a.detach().clone(), {k: p.grad.clone() for k, p in wav.named_parameters()},
                                  {k: v.clone() for k, v in wav.state_dict().items() if 'running' in k or 'tracked' in k})
    finally:
        bf16.WAVE_FUSED = prev
    (a0, g0, s0), (a1, g1, s1), (a2, g2, s2) = res[('fp32', False)], res[('bf16', False)], res[('bf16', True)]

    def l2(a, b):
        a, b = a.double().cpu(), b.double().cpu()
        return float((a - b).norm() / max(1e-12, float(b.norm())))
    print(f'[fused wave encoder, B={B}] out vs fp32: fused {rel(a2, a0):.2e}, layer-by-layer {rel(a1, a0):.2e}; gradient L2 vs '
          'fp32 (fused / layer-by-layer): ' + ', '.join(f'{k[15:]} {l2(g2[k], g0[k]):.3f}/{l2(g1[k], g0[k]):.3f}' for k in g0 if k not in ('feat_extractor.0.bias', 'feat_extractor.3.bias', 'feat_extractor.6.bias')))
    assert rel(a2, a0) < 2e-2 and rel(a2, a1) < 2e-2
    for k in s0:
        assert rel(s2[k].float(), s0[k].float()) < 5e-3, k
        assert int(s2['feat_extractor.1.num_batches_tracked']) == 1
    dead = ('feat_extractor.0.bias', 'feat_extractor.3.bias', 'feat_extractor.6.bias')     # a BatchNorm cancels them
    for k in g0:
        if k in dead:       # true value 0: what is left is the rounding of bf16 dy terms, as in the layer-by-layer path
            assert float(g2[k].abs().max()) <= 3.0 * float(g1[k].abs().max()) + 1e-3, k
        else:
            assert l2(g2[k], g0[k]) < 0.15, (k, l2(g2[k], g0[k]))
