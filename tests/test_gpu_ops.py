"""Kernel-level parity: every C-ABI entry point (through ops.py) against a plain torch-CPU fp32 restatement
or the oracle.  Tolerance: 1e-4 relative-to-max (the north-star bar is 1e-3 rel fp32)."""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
import s2ag_rng  # noqa: E402
from oracle import s2ag_oracle as O  # noqa: E402

TOL = 1e-4


@pytest.fixture(scope='module')
def S():
    from speech2affective_gestures_amd import noise, ops, optim
    from speech2affective_gestures_amd import _lib
    return dict(ops=ops, noise=noise, optim=optim, lib=_lib)


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / max(1e-6, float(b.abs().max())))


def test_library_is_loaded_in_process(S):
    S['lib'].load()
    import os
    name = os.path.basename(os.environ.get('S2AG_HIP_LIB', 'libs2ag_hip.so'))      # a debug / asan flavour of the same ABI
    with open('/proc/self/maps') as f:
        assert name in f.read()


def test_cpu_tensor_raises(S):
    with pytest.raises(RuntimeError):
        S['ops'].linear(torch.randn(3, 4), torch.randn(5, 4), None)


def test_rng_matches_numpy_restatement(S):
    ops, noise = S['ops'], S['noise']
    st = torch.tensor([1234, 5], dtype=torch.int64, device='cuda')
    m = ops.dropout_mask(st, 77, 0.3, (1000,)).cpu().numpy()
    np.testing.assert_array_equal(m, s2ag_rng.keep_mask(1234, 5, 77, 0.3, 1000))
    e = ops.normal_noise(st, 9001, (4096,)).cpu().numpy()
    np.testing.assert_allclose(e, s2ag_rng.normal(1234, 5, 9001, 4096), atol=3e-6)
    big = ops.normal_noise(st, 3, (1 << 20,))
    assert abs(float(big.mean())) < 5e-3 and abs(float(big.std()) - 1.0) < 5e-3
    keep = (ops.dropout_mask(st, 4, 0.3, (1 << 20,)) > 0).float().mean()
    assert abs(float(keep) - 0.7) < 3e-3
    # passes advance the device counter
    noise.manual_seed(42)
    a, b = noise.begin_pass('cuda'), noise.begin_pass('cuda')
    assert a.tolist() == [42, 0] and b.tolist() == [42, 1]


# (N, Lin, Cin, Cout, k, stride, pad, dil, causal)
CONV_CASES = [
    (3, 700, 1, 16, 15, 5, 160, 1, False),     # WavEncoder conv1 shape family (Cin = 1, big padding)
    (3, 300, 16, 32, 15, 6, 0, 1, False),      # strided
    (3, 800, 16, 32, 15, 6, 0, 1, False),      # strided, flat-window forward kernel with reference-layout weights
    (2, 37, 71, 64, 5, 1, 2, 1, False),        # MFCCEncoder conv1 (odd channel count -> scalar loads)
    (4, 34, 300, 300, 2, 1, 4, 4, True),       # TCN block, dilation 4, causal + chomp
    (5, 34, 27, 16, 3, 1, 0, 1, False),        # ConvDiscriminator pre_conv
    (70, 1, 37, 32, 1, 1, 0, 1, False),        # Linear
    (1, 5, 8, 1, 1, 1, 0, 1, False),           # single output column
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv1d_nlc_forward_backward(S, case):
    ops = S['ops']
    N, Lin, Cin, Cout, k, stride, pad, dil, causal = case
    g = torch.Generator().manual_seed(sum(case[:8]))
    x = torch.randn(N, Lin, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) / math.sqrt(Cin * k)
    b = torch.randn(Cout, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    if causal:
        yr = F.conv1d(xr.transpose(1, 2), wr, br, padding=pad, dilation=dil)[:, :, :-pad].transpose(1, 2)
    else:
        yr = F.conv1d(xr.transpose(1, 2), wr, br, stride=stride, padding=pad, dilation=dil).transpose(1, 2)
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
    yg = ops.conv1d_nlc(xg, wg, bg, stride=stride, pad=pad, dil=dil, lout=Lin if causal else None)
    assert rel(yg, yr) < TOL
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    yg.backward(dy.cuda())
    assert rel(xg.grad, xr.grad) < TOL
    assert rel(wg.grad, wr.grad) < TOL
    assert rel(bg.grad, br.grad) < TOL


# (N, L, Cin, Cout, k, pad, dil, causal): stride-1 layers with TAP-MAJOR weights -> the straight-line kernels of gemm_lin.hip
# (forward, data gradient and weight gradient), with K / row / column tails and every padding style on the path
TM_CASES = [
    (128, 34, 300, 300, 2, 4, 4, True),       # TCN level 2 at full batch (64-row tiles, K = 600 = 18.75 tiles)
    (3, 34, 300, 300, 2, 8, 8, True),         # few rows: 32-row tiles, ragged last row block
    (5, 34, 144, 48, 3, 1, 1, False),         # folded ST-GCN conv: "same" padding, Cout < 64
    (2, 40, 36, 100, 9, 4, 1, False),         # 9 taps, channel count not a multiple of 32 (chunks straddle taps)
    (7, 33, 64, 64, 1, 0, 1, False),          # 1 tap through the tap-major entry (ks == 1)
    (1, 32, 32, 17, 3, 1, 1, False),          # smallest supported clip length / channel count, odd Cout
]


@pytest.mark.parametrize('split', [False, True])
@pytest.mark.parametrize('case', TM_CASES)
def test_tap_major_conv_straight_line_kernels(S, case, split, monkeypatch):
    ops = S['ops']
    # split: the bf16-pipe forward kernel (conv_sp_k: activations split by the loader) at every size; else the f32 kernels
    monkeypatch.setattr(ops, 'SPLIT_CONV', bool(split))
    monkeypatch.setattr(ops, 'SPLIT_CONV_MIN_FLOPS', 0.0)
    N, Ln, Cin, Cout, k, pad, dil, causal = case
    g = torch.Generator().manual_seed(sum(case[:7]))
    x = torch.randn(N, Ln, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) / math.sqrt(Cin * k)
    b = torch.randn(Cout, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    if causal:
        yr = F.conv1d(xr.transpose(1, 2), wr, br, padding=pad, dilation=dil)[:, :, :-pad].transpose(1, 2)
    else:
        yr = F.conv1d(xr.transpose(1, 2), wr, br, padding=pad, dilation=dil).transpose(1, 2)
    xg, bg = x.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    wtm = w.permute(0, 2, 1).contiguous().cuda().requires_grad_(True)            # (Cout, k, Cin)
    yg = ops.conv1d_nlc(xg, wtm, bg, pad=pad, dil=dil, lout=Ln, w_tap_major=True, bn_stats=True)
    assert rel(yg, yr) < TOL
    st = getattr(yg, '_s2ag_stats', None)       # column sums left by the epilogue (absent where the general kernel ran)
    if st is not None:
        part, rows = st
        part = part.view(2, rows, Cout).sum(1).cpu()
        flat = yr.detach().reshape(-1, Cout).double()
        assert rel(part[0], flat.sum(0)) < 1e-6 and rel(part[1], (flat * flat).sum(0)) < 1e-6
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    yg.backward(dy.cuda())
    assert rel(xg.grad, xr.grad) < TOL
    assert rel(wtm.grad.permute(0, 2, 1), wr.grad) < TOL
    assert rel(bg.grad, br.grad) < TOL


@pytest.mark.parametrize('case', TM_CASES + [(70, 1, 36, 20, 1, 0, 1, False), (9, 34, 64, 48, 1, 0, 1, False)])
def test_fused_data_and_weight_gradient_launch(S, case, monkeypatch):
    """With gradient slots in place (the trainer's arena / staging buffers) a stride-1 layer's data gradient and weight +
    bias gradient are ONE launch (bwd_pair_k); results as the separate kernels', accumulated onto what the slots held."""
    ops = S['ops']
    N, Ln, Cin, Cout, k, pad, dil, causal = case
    g = torch.Generator().manual_seed(sum(case[:7]) + 1)
    x = torch.randn(N, Ln, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) / math.sqrt(Cin * k)
    b = torch.randn(Cout, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    if causal:
        yr = F.conv1d(xr.transpose(1, 2), wr, br, padding=pad, dilation=dil)[:, :, :-pad].transpose(1, 2)
    else:
        yr = F.conv1d(xr.transpose(1, 2), wr, br, padding=pad, dilation=dil).transpose(1, 2)
    xg, bg = x.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    wtm = w.permute(0, 2, 1).contiguous().cuda().requires_grad_(True)            # (Cout, k, Cin)
    w0, b0 = torch.randn(wtm.shape, generator=g).cuda(), torch.randn(Cout, generator=g).cuda()
    wtm.grad, bg.grad = w0.clone(), b0.clone()
    taken = []
    real = ops.conv_bwd_pair_raw
    monkeypatch.setattr(ops, 'conv_bwd_pair_raw', lambda *a: (taken.append(real(*a)), taken[-1])[1])
    yg = ops.conv1d_nlc(xg, wtm, bg, pad=pad, dil=dil, lout=yr.shape[1], w_tap_major=True)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    yg.backward(dy.cuda())
    assert taken == [(Ln == 1 or Ln >= 32) and Cout % 4 == 0]      # else: the two separate launches
    assert rel(xg.grad, xr.grad) < TOL
    assert rel((wtm.grad - w0).permute(0, 2, 1), wr.grad) < TOL
    assert rel(bg.grad - b0, br.grad) < TOL


@pytest.mark.parametrize('N,Lin,stride,pad', [(3, 4000, 5, 100), (2, 700, 5, 1600), (5, 333, 1, 0), (1, 5000, 8, 7)])
def test_one_channel_wave_conv_direct_kernels(S, N, Lin, stride, pad):
    """nn.Conv1d(1, 16, 15, ...) -- the head of the wave encoder -- runs on direct kernels (conv_c1.hip): forward with
    the BatchNorm column sums riding along, weight + bias gradient; several runs per clip, ragged last run, padding
    larger than a run."""
    ops = S['ops']
    g = torch.Generator().manual_seed(N + Lin + stride + pad)
    x = torch.randn(N, Lin, 1, generator=g)
    w = torch.randn(16, 1, 15, generator=g) / 4
    b = torch.randn(16, generator=g)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv1d(x.transpose(1, 2), wr, br, stride=stride, padding=pad).transpose(1, 2)
    wg, bg = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    yg = ops.conv1d_nlc(x.cuda(), wg, bg, stride=stride, pad=pad, bn_stats=True)
    assert yg.shape == yr.shape and rel(yg, yr) < TOL
    part, rows = yg._s2ag_stats                  # this geometry always has the statistics epilogue
    part = part.view(2, rows, 16).sum(1).cpu()
    flat = yr.detach().reshape(-1, 16).double()
    assert rel(part[0], flat.sum(0)) < 1e-6 and rel(part[1], (flat * flat).sum(0)) < 1e-6
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    yg.backward(dy.cuda())
    assert rel(wg.grad, wr.grad) < TOL and rel(bg.grad, br.grad) < TOL


# (N, Lin, Cin, Cout, k, stride, pad): STRIDED convs with tap-major weights (the wave / MFCC encoders' derived copies):
# forward + weight gradient on the straight-line kernels, data gradient on the general residue kernel
TM_STRIDED = [
    (3, 300, 16, 32, 15, 6, 0),               # WavEncoder conv2 (Cin < 32: several taps per 32-wide K tile)
    (3, 1000, 16, 32, 15, 6, 0),              # ... long enough for the flat-window forward kernel (Lout = 165: ragged tail)
    (2, 7891, 16, 32, 15, 6, 0),              # ... at the encoder's own length (Lout = 1313, several chunks per clip)
    (70, 399, 16, 32, 15, 6, 0),              # ... Lout = 65: one chunk per clip, second wave round nearly empty
    (2, 260, 32, 64, 15, 6, 0),               # WavEncoder conv3
    (5, 250, 64, 32, 15, 6, 0),               # WavEncoder conv4: Lout = 40
    (4, 77, 20, 24, 5, 2, 3),                 # padding + stride, K = 100 (tail), Lout = 40
    (2, 40, 8, 9, 3, 3, 1),                   # Lout = 14 < 32: the weight gradient falls back to the general kernel
]


@pytest.mark.parametrize('case', TM_STRIDED)
def test_tap_major_strided_conv(S, case):
    ops = S['ops']
    N, Lin, Cin, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Lin, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) / math.sqrt(Cin * k)
    b = torch.randn(Cout, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv1d(xr.transpose(1, 2), wr, br, stride=stride, padding=pad).transpose(1, 2)
    xg, bg = x.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    wtm = w.permute(0, 2, 1).contiguous().cuda().requires_grad_(True)
    yg = ops.conv1d_nlc(xg, wtm, bg, stride=stride, pad=pad, w_tap_major=True, bn_stats=True)
    assert yg.shape == yr.shape and rel(yg, yr) < TOL
    st = getattr(yg, '_s2ag_stats', None)
    if st is not None:
        part, rows = st
        part = part.view(2, rows, Cout).sum(1).cpu()
        flat = yr.detach().reshape(-1, Cout).double()
        assert rel(part[0], flat.sum(0)) < 1e-6 and rel(part[1], (flat * flat).sum(0)) < 1e-6
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    yg.backward(dy.cuda())
    assert rel(xg.grad, xr.grad) < TOL
    assert rel(wtm.grad.permute(0, 2, 1), wr.grad) < TOL
    assert rel(bg.grad, br.grad) < TOL


@pytest.mark.parametrize('Cin,Cout', [(16, 32), (32, 64), (64, 32)])
@pytest.mark.parametrize('N,Lin', [(1, 15), (3, 16), (2, 97), (5, 1000), (2, 7891)])
@pytest.mark.parametrize('tap_major', [False, True])
def test_polyphase_strided_data_gradient(S, Cin, Cout, N, Lin, tap_major):
    """The poly-phase data-gradient kernel of the wave encoder's strided convs (csrc/conv_pp.hip; k = 15, stride 6) against
    torch's conv_transpose1d: one output frame, ragged tails (frames no window touches stay zero), several q chunks per
    clip, both weight layouts, a column-sliced gy and the accumulate flag."""
    ops = S['ops']
    k, stride = 15, 6
    Lout = (Lin - k) // stride + 1
    g = torch.Generator().manual_seed(Cin + Cout + N + Lin)
    w = torch.randn(Cout, Cin, k, generator=g) / math.sqrt(Cin * k)
    wide = torch.randn(N, Lout, Cout + 8, generator=g)
    gy = wide[..., 4:4 + Cout]
    ref = F.conv_transpose1d(gy.transpose(1, 2), w, stride=stride).transpose(1, 2)        # (N, 6 (Lout - 1) + 15, Cin)
    ref = F.pad(ref, (0, 0, 0, Lin - ref.shape[1]))
    wg = (w.permute(0, 2, 1).contiguous() if tap_major else w).cuda()
    gyg = wide.cuda()[..., 4:4 + Cout]
    dx = torch.full((N * Lin, Cin), float('nan'), device='cuda')
    ops.conv_bwd_data_raw(gyg.view(N * Lout, Cout), wg, dx, N, Lin, Lout, Cin, Cout, k, stride, 0, 1, False, int(tap_major))
    assert torch.isfinite(dx).all()
    assert rel(dx.view(N, Lin, Cin), ref) < TOL
    base = torch.randn(N * Lin, Cin, generator=g)
    dx2 = base.cuda()
    ops.conv_bwd_data_raw(gyg.view(N * Lout, Cout), wg, dx2, N, Lin, Lout, Cin, Cout, k, stride, 0, 1, True, int(tap_major))
    assert rel(dx2.view(N, Lin, Cin), ref + base.view(N, Lin, Cin)) < TOL


@pytest.mark.parametrize('M,K,N', [(4352, 600, 1800), (70, 36, 5), (33, 100, 64), (257, 88, 900), (64, 4, 16)])
def test_linear_straight_line_kernels_with_tails(S, M, K, N):
    """Linear forward / data gradient / weight gradient (+ bias gradient in the same launch) at shapes whose row, column
    and K extents are not multiples of the 64 x 64 x 32 tile; the GRU projection shape at full size."""
    ops = S['ops']
    g = torch.Generator().manual_seed(M + K + N)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.linear(xr, wr, br)
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
    yg = ops.linear(xg, wg, bg)
    assert rel(yg, yr) < TOL
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    yg.backward(dy.cuda())
    assert rel(xg.grad, xr.grad) < TOL and rel(wg.grad, wr.grad) < TOL and rel(bg.grad, br.grad) < TOL


def test_shifted_frame_weight_gradient(S):
    """dW_hh of the GRU is a 1-tap weight gradient against the state shifted by one frame inside each clip (pad = +-1):
    the straight-line kernel's frame-boundary test."""
    ops = S['ops']
    g = torch.Generator().manual_seed(77)
    B, T, H, H3 = 9, 34, 40, 120
    gy, h = torch.randn(B * T, H3, generator=g), torch.randn(B * T, H, generator=g)
    for pad in (1, -1):
        dw = torch.zeros(H3, H, device='cuda')
        db = torch.zeros(H3, device='cuda')
        ops.conv_bwd_weight_raw(gy.cuda(), h.cuda(), dw, B, T, T, H, H3, 1, 1, pad, 1, True, dbias=db)
        hp = torch.zeros(B, T, H)
        hv = h.view(B, T, H)
        if pad == 1:
            hp[:, 1:] = hv[:, :-1]
        else:
            hp[:, :-1] = hv[:, 1:]
        ref = gy.t() @ hp.reshape(B * T, H)
        assert rel(dw, ref) < TOL and rel(db, gy.sum(0)) < TOL


def test_conv_reads_and_writes_column_slices(S):
    ops = S['ops']
    g = torch.Generator().manual_seed(3)
    wide = torch.randn(6, 10, 40, generator=g)
    w = torch.randn(12, 16, 3, generator=g)
    ref = F.conv1d(wide[..., 8:24].transpose(1, 2), w, padding=1).transpose(1, 2)
    out = ops.conv1d_nlc(wide.cuda()[..., 8:24], w.cuda(), None, pad=1)
    assert rel(out, ref) < TOL


@pytest.mark.parametrize('act,slope', [('leaky', 0.3), ('leaky', 0.0), ('sigmoid', 0.0)])
def test_linear_fused_activation(S, act, slope):
    ops, lib = S['ops'], S['lib']
    g = torch.Generator().manual_seed(5)
    x, w, b = torch.randn(9, 34, generator=g), torch.randn(7, 34, generator=g), torch.randn(7, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    z = F.linear(xr, wr, b)
    yr = torch.sigmoid(z) if act == 'sigmoid' else F.leaky_relu(z, slope)
    xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    yg = ops.linear(xg, wg, b.cuda(), act=lib.ACT_SIGMOID if act == 'sigmoid' else lib.ACT_LEAKY, slope=slope)
    assert rel(yg, yr) < TOL
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    yg.backward(dy.cuda())
    assert rel(xg.grad, xr.grad) < TOL and rel(wg.grad, wr.grad) < TOL


def test_conv_epilogue_dropout_uses_materialised_mask(S):
    ops, noise, lib = S['ops'], S['noise'], S['lib']
    g = torch.Generator().manual_seed(7)
    x, w, b = torch.randn(3, 20, 24, generator=g), torch.randn(40, 24, 2, generator=g) / 7, torch.randn(40, generator=g)
    noise.manual_seed(11)
    nz = noise.begin_pass('cuda')
    xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    y = ops.conv1d_nlc(xg, wg, b.cuda(), pad=2, dil=2, lout=20, act=lib.ACT_LEAKY, slope=0.0, drop_p=0.3, noise=nz,
                       site=17)
    mask = ops.dropout_mask(nz, 17, 0.3, (3, 20, 40)).cpu()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.relu(F.conv1d(xr.transpose(1, 2), wr, b, padding=2, dilation=2)[:, :, :-2].transpose(1, 2)) * mask
    assert rel(y, yr) < TOL
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    y.backward(dy.cuda())
    assert rel(xg.grad, xr.grad) < TOL and rel(wg.grad, wr.grad) < TOL


@pytest.mark.parametrize('slope', [0.3, 1.0, 0.0])
def test_batch_norm_act_train_and_eval(S, slope):
    ops = S['ops']
    g = torch.Generator().manual_seed(9)
    x = torch.randn(6, 37, 48, generator=g) * 2 + 0.5
    bn_r = torch.nn.BatchNorm1d(48)
    with torch.no_grad():
        bn_r.weight.uniform_(0.5, 1.5, generator=g)
        bn_r.bias.uniform_(-0.5, 0.5, generator=g)
    bn_g = torch.nn.BatchNorm1d(48)
    bn_g.load_state_dict(bn_r.state_dict())
    bn_g.cuda()
    xr = x.clone().requires_grad_(True)
    yr = F.leaky_relu(bn_r(xr.transpose(1, 2)), slope).transpose(1, 2)
    xg = x.cuda().requires_grad_(True)
    yg = ops.batch_norm_act(xg, bn_g, slope=slope)
    assert rel(yg, yr) < TOL
    assert rel(bn_g.running_mean, bn_r.running_mean) < TOL and rel(bn_g.running_var, bn_r.running_var) < TOL
    assert int(bn_g.num_batches_tracked) == 1
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    yg.backward(dy.cuda())
    assert rel(xg.grad, xr.grad) < 5 * TOL
    assert rel(bn_g.weight.grad, bn_r.weight.grad) < TOL and rel(bn_g.bias.grad, bn_r.bias.grad) < TOL
    bn_r.eval(), bn_g.eval()
    assert rel(ops.batch_norm_act(x.cuda(), bn_g, slope=slope), F.leaky_relu(bn_r(x.transpose(1, 2)), slope).transpose(1, 2)) < TOL


def test_batch_norm_narrow_matrix_lane_dense_path(S):
    """16 columns x many rows (the wave encoder's first activation): the flat lane-dense reduction kernels."""
    ops = S['ops']
    g = torch.Generator().manual_seed(19)
    x = torch.randn(3, 5000, 16, generator=g) * 1.5 - 0.3
    bn_r, bn_g = torch.nn.BatchNorm1d(16), torch.nn.BatchNorm1d(16).cuda()
    xr = x.clone().requires_grad_(True)
    yr = F.leaky_relu(bn_r(xr.transpose(1, 2)), 0.3).transpose(1, 2)
    xg = x.cuda().requires_grad_(True)
    yg = ops.batch_norm_act(xg, bn_g, slope=0.3)
    assert rel(yg, yr) < TOL and rel(bn_g.running_var, bn_r.running_var) < TOL
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    yg.backward(dy.cuda())
    assert rel(xg.grad, xr.grad) < 5 * TOL and rel(bn_g.weight.grad, bn_r.weight.grad) < TOL
    assert rel(bn_g.bias.grad, bn_r.bias.grad) < TOL


@pytest.mark.parametrize('one_launch', [False, True])
@pytest.mark.parametrize('rows,cols,strided', [(4352, 900, False), (70000, 64, False), (300000, 8, False),
                                               (4352, 96, True), (33, 257, False), (1, 5, False)])
def test_batch_norm_fused_statistics_shapes_and_ticket_rearm(S, rows, cols, strided, one_launch, monkeypatch):
    """The one-launch statistics kernels (row-block partial sums, last block folds them): wide / tall / narrow / strided
    matrices, launched repeatedly on the same ticket words (each launch must leave its ticket at zero).  ``one_launch``:
    the form that also applies the coefficients in the same kernel behind a grid-wide wait (ops.BN_FUSED; its barrier words
    must be left zero too, and the sticky error word must stay clear)."""
    ops = S['ops']
    monkeypatch.setattr(ops, 'BN_FUSED', one_launch)
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols, generator=g) * 1.7 + 0.4
    bn_r, bn_g = torch.nn.BatchNorm1d(cols), torch.nn.BatchNorm1d(cols).cuda()
    dy = torch.randn(rows, cols, generator=g)
    if rows > 1:
        xr = x.clone().requires_grad_(True)
        yr = F.leaky_relu(bn_r(xr), 0.2)
        yr.backward(dy)
    for rep in range(3):
        bn_g.reset_running_stats()
        bn_g.weight.grad = bn_g.bias.grad = None
        if strided:
            big = torch.zeros(rows, cols + 32, device='cuda')
            big[:, 16:16 + cols] = x.cuda()
            xg = big.requires_grad_(True)
            yg = ops.batch_norm_act(xg[:, 16:16 + cols], bn_g, slope=0.2)
        else:
            xg = x.cuda().requires_grad_(True)
            yg = ops.batch_norm_act(xg, bn_g, slope=0.2)
        yg.backward(dy.cuda())
        if rows == 1:
            assert torch.isfinite(yg).all()
            continue
        gx = xg.grad[:, 16:16 + cols] if strided else xg.grad
        assert rel(yg, yr) < TOL, rep
        assert rel(bn_g.running_mean, bn_r.running_mean) < TOL and rel(bn_g.running_var, bn_r.running_var) < TOL
        assert rel(gx, xr.grad) < 5 * TOL
        assert rel(bn_g.weight.grad, bn_r.weight.grad) < 2 * TOL and rel(bn_g.bias.grad, bn_r.bias.grad) < 2 * TOL
    pool = ops._TICKETS[torch.cuda.current_device()][0]
    assert int(pool.abs().sum()) == 0
    assert int(ops._BARRIERS[torch.cuda.current_device()][0].abs().sum()) == 0
    assert int(ops._COOP_FLAG[torch.cuda.current_device()].item()) == 0


def test_batch_norm_channel_map_is_batchnorm2d(S):
    ops = S['ops']
    g = torch.Generator().manual_seed(10)
    N, T, V, Cc = 4, 11, 9, 16
    x = torch.randn(N, Cc, T, V, generator=g)
    bn_r, bn_g = torch.nn.BatchNorm2d(Cc), torch.nn.BatchNorm2d(Cc).cuda()
    cmap = torch.arange(V * Cc, dtype=torch.int32) % Cc          # columns ordered (v, c)
    xr = x.clone().requires_grad_(True)
    yr = F.relu(bn_r(xr))
    xg = x.permute(0, 2, 3, 1).reshape(N, T, V * Cc).cuda().requires_grad_(True)
    yg = ops.batch_norm_act(xg, bn_g, slope=0.0, chan_map=cmap.cuda())
    assert rel(yg.view(N, T, V, Cc).permute(0, 3, 1, 2), yr) < TOL
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    yg.backward(dy.permute(0, 2, 3, 1).reshape(N, T, V * Cc).cuda())
    assert rel(xg.grad.view(N, T, V, Cc).permute(0, 3, 1, 2), xr.grad) < 5 * TOL
    assert rel(bn_g.weight.grad, bn_r.weight.grad) < TOL
    assert rel(bn_g.running_var, bn_r.running_var) < TOL


def _gru_sd(I, H, L, seed):
    g = torch.Generator().manual_seed(seed)
    k = 1 / math.sqrt(H)
    return {n: (torch.rand(s, generator=g) * 2 - 1) * k for n, s in O._gru_shapes('gru.', I, H, L).items()}


def _flat(sd, L):
    out = []
    for l in range(L):
        for suf in ('', '_reverse'):
            out += [sd[f'gru.weight_ih_l{l}{suf}'], sd[f'gru.weight_hh_l{l}{suf}'], sd[f'gru.bias_ih_l{l}{suf}'],
                    sd[f'gru.bias_hh_l{l}{suf}']]
    return out


@pytest.mark.parametrize('B,T,I,H,L,sum_dirs,p', [(5, 7, 11, 32, 3, False, 0.0), (9, 6, 88, 300, 2, True, 0.0),
                                                   (10, 34, 8, 64, 4, True, 0.3), (3, 5, 20, 32, 2, False, 0.3),
                                                   (37, 34, 88, 300, 3, True, 0.3), (128, 34, 108, 300, 2, False, 0.3),
                                                   (17, 1, 88, 300, 2, False, 0.3), (33, 2, 20, 300, 1, True, 0.0),
                                                   (16, 3, 20, 300, 1, False, 0.0), (1, 34, 88, 300, 2, True, 0.3)])
def test_gru_forward_backward(S, B, T, I, H, L, sum_dirs, p):
    ops, noise = S['ops'], S['noise']
    sd = _gru_sd(I, H, L, B * 100 + H)
    g = torch.Generator().manual_seed(B + T)
    x = torch.randn(B, T, I, generator=g)
    noise.manual_seed(5)
    nz = noise.begin_pass('cuda')
    site0 = 300
    wg = [w.cuda().requires_grad_(True) for w in _flat(sd, L)]
    xg = x.cuda().requires_grad_(True)
    yg = ops.gru(xg, wg, H, L, True, p, nz, site0, sum_dirs)
    pinned = {f'gru.drop{l}': ops.dropout_mask(nz, site0 + l, p, (B, T, 2 * H)).cpu() for l in range(L - 1)} \
        if p > 0 else 'off'
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = O.gru(sdr, 'gru.', xr, True, p, O.Noise(pinned), 'gru')
    if sum_dirs:
        yr = yr[..., :H] + yr[..., H:]
    assert rel(yg, yr) < TOL
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    yg.backward(dy.cuda())
    assert rel(xg.grad, xr.grad) < 2 * TOL
    names = []
    for l in range(L):
        for suf in ('', '_reverse'):
            names += [f'gru.weight_ih_l{l}{suf}', f'gru.weight_hh_l{l}{suf}', f'gru.bias_ih_l{l}{suf}',
                      f'gru.bias_hh_l{l}{suf}']
    for n, w in zip(names, wg):
        assert rel(w.grad, sdr[n].grad) < 2 * TOL, n
    assert ops.coop_gru_timeouts() == 0          # cooperative (H = 300) launches never timed out on a peer


@pytest.mark.parametrize('B,I,H,L,sum_dirs,n_mates', [(128, 108, 300, 4, True, 2), (37, 88, 300, 2, False, 2),
                                                        (48, 88, 300, 2, True, 3), (128, 108, 300, 2, True, 1),
                                                        (9, 20, 32, 2, True, 2), (16, 88, 300, 2, True, 2)])
def test_lockstep_gru_passes_equal_separate_launches(S, B, I, H, L, sum_dirs, n_mates):
    """ops.gru(..., mates=[...]): further no-grad passes over the same weights run layer by layer in lockstep with the
    main pass -- at H = 300 their recurrences ride in ONE cooperative launch (s2ag_gru_coop_fwd_multi), each with its own
    noise snapshot.  Every pass must equal its stand-alone launch bit for bit, and the main pass's gradients too."""
    ops, noise = S['ops'], S['noise']
    T, p, site0 = 34, 0.3, 300
    sd = _gru_sd(I, H, L, B + H)
    g = torch.Generator().manual_seed(B + I)
    xs = [torch.randn(B, T, I, generator=g).cuda() for _ in range(1 + n_mates)]
    noise.manual_seed(11)
    nzs = [noise.begin_pass('cuda') for _ in xs]
    dy = torch.randn(B, T, H if sum_dirs else 2 * H, generator=g).cuda()

    def run(lockstep):
        wg = [w.cuda().requires_grad_(True) for w in _flat(sd, L)]
        x0 = xs[0].clone().requires_grad_(True)
        if lockstep:
            outs = ops.gru(x0, wg, H, L, True, p, nzs[0], site0, sum_dirs, mates=list(zip(xs[1:], nzs[1:])))
            assert not any(o.requires_grad for o in outs[1:])
        else:
            outs = [ops.gru(x0, wg, H, L, True, p, nzs[0], site0, sum_dirs)]
            with torch.no_grad():
                outs += [ops.gru(x, wg, H, L, True, p, nz, site0, sum_dirs) for x, nz in zip(xs[1:], nzs[1:])]
        outs[0].backward(dy)
        torch.cuda.synchronize()
        return [o.detach().clone() for o in outs], x0.grad.clone(), [w.grad.clone() for w in wg]
    o1, gx1, gw1 = run(True)
    o0, gx0, gw0 = run(False)
    assert len(o1) == 1 + n_mates
    for a, b in zip(o1, o0):
        assert torch.equal(a, b)
    assert not torch.equal(o1[0], o1[1])                 # different inputs and noise per pass
    assert torch.equal(gx1, gx0)
    for a, b in zip(gw1, gw0):
        assert rel(a, b) < 1e-6                         # (weight-gradient tiles are summed in launch-dependent order)
    assert ops.coop_gru_timeouts() == 0


@pytest.mark.parametrize('pieces,tol', [(0, 3e-6), (3, 3e-6), (2, 2e-5), (1, 2e-2)])
def test_coop_gru_products_on_the_bf16_pipe(S, pieces, tol):
    """H = 300 recurrence: the per-step products as exact bf16-piece splits of the fp32 operands (3 pieces = default,
    2 pieces) against the f32-MFMA kernels (0) -- all three checked against an fp64 torch GRU.  Three pieces must be as
    accurate as the f32 MFMA; two pieces carry 16 mantissa bits; one piece (bf16.precision('bf16_step'): a single bf16 product,
    8 mantissa bits per operand) is held to 2e-2 of the largest element over the two layers and 34 steps."""
    ops, noise, lib = S['ops'], S['noise'], S['ops']._lib()
    B, T, I, H, L_ = 21, 34, 88, 300, 2
    sd = _gru_sd(I, H, L_, 4242)
    g = torch.Generator().manual_seed(99)
    x = torch.randn(B, T, I, generator=g)
    ref = torch.nn.GRU(I, H, L_, batch_first=True, bidirectional=True).double()
    with torch.no_grad():
        for k, v in sd.items():
            getattr(ref, k[len('gru.'):]).copy_(v.double())
    xr = x.double().requires_grad_(True)
    yr, _ = ref(xr)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    prev = lib.s2ag_gru_coop_split_pieces()
    lib.s2ag_gru_coop_set_split_pieces(pieces)
    try:
        assert lib.s2ag_gru_coop_split_pieces() == pieces
        wg = [w.cuda().requires_grad_(True) for w in _flat(sd, L_)]
        xg = x.cuda().requires_grad_(True)
        yg = ops.gru(xg, wg, H, L_, True, 0.0, noise.begin_pass('cuda'), 300, False)
        yg.backward(dy.cuda())
        torch.cuda.synchronize()
    finally:
        lib.s2ag_gru_coop_set_split_pieces(-1)
    assert lib.s2ag_gru_coop_split_pieces() == prev
    err = lambda a, b: float((a.detach().double().cpu() - b.detach()).abs().max() / b.detach().abs().max())
    assert err(yg, yr) < tol and err(xg.grad, xr.grad) < tol
    assert err(wg[1].grad, ref.weight_hh_l0.grad) < tol and err(wg[0].grad, ref.weight_ih_l0.grad) < tol
    assert ops.coop_gru_timeouts() == 0


@pytest.mark.parametrize('M,K,N', [(4352, 600, 1800), (70, 36, 5), (33, 100, 64), (257, 88, 900), (64, 32, 64)])
def test_split_operand_gemm_accuracy(S, M, K, N):
    """y = a w^T + b on the bf16 matrix pipe from bf16-piece splits of the fp32 operands: the three planes reproduce the
    operands to 2^-24; with three pieces the GEMM is as accurate as an fp32 one, with the default two pieces its products
    carry 16 mantissa bits (checked against fp64); tails in M, N and K."""
    ops = S['ops']
    g = torch.Generator().manual_seed(M + K + N)
    a, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    ap, wp = ops.split_planes_raw(a.cuda()), ops.split_planes_raw(w.cuda())
    Kp = ap.shape[2]
    assert ap.shape == (3, M, Kp) and Kp % 32 == 0 and Kp - K < 32 and ap.dtype == torch.bfloat16
    rec = ap.float().sum(0)[:, :K].cpu()
    assert float((rec - a).abs().max()) <= 2.0 ** -23 * float(a.abs().max())
    assert float(ap.float()[:, :, K:].abs().max() if Kp > K else 0.0) == 0.0
    y = torch.empty(M, N, device='cuda')
    ops.gemm_split_raw(ap, wp, b.cuda(), y, K)
    ref = a.double() @ w.double().t() + b.double()
    err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    err32 = float(((a @ w.t() + b).double() - ref).abs().max() / ref.abs().max())
    if S['ops']._lib().s2ag_gru_coop_split_pieces() == 3:
        assert err < max(2e-6, 4 * err32), (err, err32)          # six products: fp32-equivalent
    else:
        assert err < 2e-5, (err, err32)                          # three products: 16 mantissa bits


@pytest.mark.parametrize('rows,L_,cn,ck,shift', [(4352, 34, 1800, 600, 0), (340, 34, 96, 40, -1), (340, 34, 96, 40, 1),
                                                    (77, 7, 33, 20, 0)])
def test_split_operand_weight_gradient(S, rows, L_, cn, ck, shift):
    """dW += gy^T x (and db += column sums of gy) through transposed bf16-piece planes and the split-K GEMM; with a frame
    shift inside the clips (the GRU's dW_hh pairs d(gh)_t with h_{t-1}); row counts that are not multiples of 32."""
    ops = S['ops']
    g = torch.Generator().manual_seed(rows + cn + ck + shift)
    gy, x = torch.randn(rows, cn, generator=g), torch.randn(rows, ck, generator=g)
    dw0, db0 = torch.randn(cn, ck, generator=g), torch.randn(cn, generator=g)
    dw, db = dw0.cuda(), db0.cuda()
    gT = ops.split_planes_t_raw(gy.cuda(), colsum=db)
    xT = ops.split_planes_t_raw(x.cuda(), shift=shift, L_=L_)
    assert gT.shape[1] == cn and gT.shape[2] % 32 == 0 and gT.shape[2] - rows < 32
    assert float(gT.float().sum(0)[:, rows:].abs().max() if gT.shape[2] > rows else 0.0) == 0.0
    ops.gemm_split_acc_raw(gT, xT, dw, rows)
    xs = torch.zeros_like(x).view(-1, L_, ck)
    xv = x.view(-1, L_, ck)
    if shift == 0:
        xs = xv.clone()
    elif shift < 0:
        xs[:, -shift:] = xv[:, :shift]           # row of frame t pairs with frame t + shift
    else:
        xs[:, :-shift] = xv[:, shift:]
    ref = dw0.double() + gy.double().t() @ xs.reshape(rows, ck).double()
    err = float((dw.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < (3e-6 if ops._lib().s2ag_gru_coop_split_pieces() == 3 else 3e-5), err
    assert rel(db, db0 + gy.sum(0)) < 1e-5


def test_embedding_dropout_and_dense_gradient(S):
    ops, noise = S['ops'], S['noise']
    g = torch.Generator().manual_seed(12)
    table = torch.randn(50, 300, generator=g)
    ids = torch.randint(0, 50, (4, 34), generator=g)
    ids[0, :5] = 3                                    # repeated rows exercise the atomics
    noise.manual_seed(2)
    nz = noise.begin_pass('cuda')
    tg = table.cuda().requires_grad_(True)
    out = ops.embedding(ids.cuda(), tg, 0.1, nz, 21)
    mask = ops.dropout_mask(nz, 21, 0.1, (4, 34, 300)).cpu()
    tr = table.clone().requires_grad_(True)
    ref = F.embedding(ids, tr) * mask
    assert rel(out, ref) < 1e-6
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    out.backward(dy.cuda())
    assert rel(tg.grad, tr.grad) < TOL


@pytest.mark.parametrize('B,dim,n_entries,pad_frac', [(40, 300, 500, 0.85), (9, 300, 12, 0.0), (33, 16, 1371, 0.0),
                                                      (3, 300, 400, 1.0), (17, 44, 7, 0.5)])
def test_embedding_gradient_with_pad_runs_and_duplicates(S, B, dim, n_entries, pad_frac):
    """embedding_bwd_k as rewritten in r03 (csrc/misc.hip: 256 rows x 64 columns per workgroup, the PAD id's rows summed in
    registers and through LDS, word rows as direct atomics) at the shapes it meets -- transcripts that are ~85 % PAD
    (utils/vocab.py PAD_token = 0), several row blocks, repeated words inside and across row blocks, the 16-wide speaker
    table, all-PAD and PAD-free batches, a width that is no multiple of the 64-column tile -- against F.embedding's dense
    gradient, with dropout (net/multimodal_context_net_v2.py:70-78)."""
    ops, noise = S['ops'], S['noise']
    g = torch.Generator().manual_seed(100 + B)
    table = torch.randn(n_entries, dim, generator=g)
    ids = torch.randint(1 if n_entries > 1 else 0, n_entries, (B, 34), generator=g)
    ids[torch.rand(B, 34, generator=g) < pad_frac] = 0
    ids[0, :7] = ids[0, 0]                                    # a run of one word inside a wave's rows
    ids[-1, -1] = ids[0, 0]                                   # ... and the same word again in the last row block
    noise.manual_seed(2)
    nz = noise.begin_pass('cuda')
    tg = table.cuda().requires_grad_(True)
    out = ops.embedding(ids.cuda(), tg, 0.1, nz, 21)
    mask = ops.dropout_mask(nz, 21, 0.1, (B, 34, dim)).cpu()
    tr = table.clone().requires_grad_(True)
    ref = F.embedding(ids, tr) * mask
    assert rel(out, ref) < 1e-6
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    out.backward(dy.cuda())
    assert rel(tg.grad, tr.grad) < TOL
    assert torch.equal(tg.grad.cpu() == 0, tr.grad == 0)      # untouched rows stay exactly zero


def test_weight_norm(S):
    ops = S['ops']
    g = torch.Generator().manual_seed(13)
    v, gg = torch.randn(300, 300, 2, generator=g), torch.rand(300, 1, 1, generator=g) + 0.5
    vr, gr = v.clone().requires_grad_(True), gg.clone().requires_grad_(True)
    wr = O.weight_norm_weight(gr, vr)
    vg, g2 = v.cuda().requires_grad_(True), gg.cuda().requires_grad_(True)
    wg = ops.weight_norm(vg, g2)
    assert rel(wg, wr) < TOL
    dw = torch.randn(wr.shape, generator=g)
    wr.backward(dw)
    wg.backward(dw.cuda())
    assert rel(vg.grad, vr.grad) < TOL and rel(g2.grad, gr.grad) < TOL


def test_csr_fold_and_its_transpose(S):
    import scipy.sparse as sp
    ops = S['ops']
    rs = np.random.RandomState(0)
    m = sp.random(700, 90, density=0.05, random_state=rs, format='csr', dtype=np.float64)
    csr = ops.CSR(m, 'cuda')
    w = torch.randn(90, dtype=torch.float32)
    wg = w.cuda().requires_grad_(True)
    y = ops.fold(wg, csr)
    dense = torch.from_numpy(m.toarray()).float()
    assert rel(y, dense @ w) < TOL
    dy = torch.randn(700)
    y.backward(dy.cuda())
    assert rel(wg.grad, dense.t() @ dy) < TOL


def test_derived_groups_cache_stage_and_flush(S):
    """ops.FoldGroup / ops.WeightNormGroup: one-launch recompute per parameter version, shared by several forward passes,
    gradients of all uses staged and flushed once at the end of backward == the plain autograd ops."""
    import scipy.sparse as sp
    ops = S['ops']
    rs = np.random.RandomState(3)
    g = torch.Generator().manual_seed(31)
    Cout, Cin, k = 24, 12, 3
    m_w = sp.random(Cout * k * Cin, 40, density=0.08, random_state=rs, format='csr', dtype=np.float64)
    m_b = sp.random(Cout, 7, density=0.5, random_state=rs, format='csr', dtype=np.float64)
    csr_w, csr_b = ops.CSR(m_w, 'cuda'), ops.CSR(m_b, 'cuda')
    pw, pb = torch.randn(40, generator=g), torch.randn(7, generator=g)
    v, gg = torch.randn(Cout, Cin, k, generator=g), torch.rand(Cout, 1, 1, generator=g) + 0.5
    x1, x2 = torch.randn(3, 11, Cin, generator=g).cuda(), torch.randn(3, 11, Cin, generator=g).cuda()

    def leaves():
        return [t.clone().cuda().requires_grad_(True) for t in (pw, pb, v, gg)]
    # reference: the autograd ops, two forward passes sharing the parameters
    a = leaves()
    wf = ops.fold(a[0], csr_w).view(Cout, k, Cin)
    bf = ops.fold(a[1], csr_b)
    wn = ops.weight_norm(a[2], a[3], tap_major=True)
    ref = sum((ops.conv1d_nlc(x, wf, bf, pad=1, w_tap_major=True) * ops.conv1d_nlc(x, wn, None, pad=1, w_tap_major=True)).sum()
              for x in (x1, x2))
    ref.backward()
    # derived groups
    b = leaves()
    fg = ops.FoldGroup([b[0], b[1]], [csr_w, csr_b], [(Cout, k, Cin), (Cout,)])
    wg = ops.WeightNormGroup([b[2]], [b[3]])
    outs = []
    for x in (x1, x2):
        f, (w2,) = fg.tensors(), wg.tensors()
        outs.append((ops.conv1d_nlc(x, f[0], f[1], pad=1, w_tap_major=True) * ops.conv1d_nlc(x, w2, None, pad=1, w_tap_major=True)).sum())
    f_again = fg.tensors()
    assert f_again[0] is f[0] and wg.tensors()[0] is w2              # cached: same tensors for both passes
    assert rel(f[0], wf) < 1e-6 and rel(f[1], bf) < 1e-6 and rel(w2, wn) < 1e-6
    tot = outs[0] + outs[1]
    assert rel(tot, ref) < 1e-5
    tot.backward()                                                   # flush runs as an autograd engine callback
    for i, name in enumerate(('fold w', 'fold b', 'weight_v', 'weight_g')):
        assert rel(b[i].grad, a[i].grad) < TOL, name
    assert float(f[0].grad.abs().sum()) == 0.0 and float(w2.grad.abs().sum()) == 0.0     # stages left zeroed
    # a changed source invalidates the cache (torch version counter); a second backward accumulates into .grad
    with torch.no_grad():
        b[0].mul_(2.0)
    f2 = fg.tensors()
    assert f2[0] is not f[0] and rel(f2[0], 2 * wf) < 1e-6
    g0 = b[1].grad.clone()
    ops.conv1d_nlc(x1, f2[0], f2[1], pad=1, w_tap_major=True).sum().backward()
    assert rel(b[1].grad - g0, torch.from_numpy(m_b.toarray().T).float().cuda() @ torch.full((Cout,), 33.0).cuda()) < TOL
    # frozen sources: the derived tensors do not ask for gradients
    for t in b:
        t.requires_grad_(False)
    assert not fg.tensors()[0].requires_grad and not wg.tensors()[0].requires_grad


def test_add_act_and_transposed_slices(S):
    ops = S['ops']
    g = torch.Generator().manual_seed(14)
    a, b = torch.randn(5, 9, 30, generator=g), torch.randn(5, 9, 30, generator=g)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.leaky_relu(ar + br, 0.01)
    ag, bg = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    yg = ops.add_act(ag, bg, 0.01)
    assert rel(yg, yr) < 1e-6
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    yg.backward(dy.cuda())
    assert rel(ag.grad, ar.grad) < 1e-6 and rel(bg.grad, br.grad) < 1e-6


def test_reparametrize(S):
    ops, noise = S['ops'], S['noise']
    g = torch.Generator().manual_seed(15)
    mu, lv = torch.randn(6, 16, generator=g), torch.randn(6, 16, generator=g) * 0.3
    noise.manual_seed(3)
    nz = noise.begin_pass('cuda')
    mg, lg = mu.cuda().requires_grad_(True), lv.cuda().requires_grad_(True)
    z = ops.reparametrize(mg, lg, nz, 33)
    eps = ops.normal_noise(nz, 33, (6, 16)).cpu()
    mr, lr_ = mu.clone().requires_grad_(True), lv.clone().requires_grad_(True)
    zr = O.re_parametrize(mr, lr_, O.Noise({'eps': eps}))
    assert rel(z, zr) < TOL
    dz = torch.randn(6, 16, generator=g)
    zr.backward(dz)
    z.backward(dz.cuda())
    assert rel(mg.grad, mr.grad) < TOL and rel(lg.grad, lr_.grad) < TOL


def test_losses_and_gradients(S):
    ops = S['ops']
    g = torch.Generator().manual_seed(16)
    B, T, P = 6, 34, 27
    dr, df = torch.rand(B, 1, generator=g) * 0.9 + 0.05, torch.rand(B, 1, generator=g) * 0.9 + 0.05
    drr, dfr = dr.clone().requires_grad_(True), df.clone().requires_grad_(True)
    lr_ = O.dis_loss(drr, dfr)
    drg, dfg = dr.cuda().requires_grad_(True), df.cuda().requires_grad_(True)
    lg = ops.dis_loss(drg, dfg)
    assert rel(lg, lr_) < TOL
    lr_.backward()
    lg.backward()
    assert rel(drg.grad, drr.grad) < TOL and rel(dfg.grad, dfr.grad) < TOL

    out, tgt = torch.randn(B, T, P, generator=g) * 0.3, torch.randn(B, T, P, generator=g) * 0.2
    out_rand = out + torch.randn(B, T, P, generator=g) * 0.02
    out_rand[0] = out[0] + 5.0                         # forces the clamp(-1000) branch for one clip
    out_tri = torch.randn(B, T, P, generator=g) * 0.2
    z, z_rand = torch.randn(B, 16, generator=g), torch.randn(B, 16, generator=g)
    z_rand[0] = z[0] + 1e-4
    mu, lv = torch.randn(B, 16, generator=g), torch.randn(B, 16, generator=g) * 0.3
    dis_out = torch.rand(B, 1, generator=g) * 0.9 + 0.05
    scfg = O.StepCfg()
    req = [t.clone().requires_grad_(True) for t in (out, dis_out, mu, lv)]
    loss_r, comp = O.gen_losses(scfg, req[0], tgt, req[1], out_rand, z, z_rand, req[2], req[3], True)
    loss_r.backward()
    gq = [t.cuda().requires_grad_(True) for t in (out, dis_out, mu, lv)]
    total, comps = ops.gen_loss(gq[0], gq[1], gq[2], gq[3], tgt.cuda(), out_tri.cuda(), out_rand.cuda(), z.cuda(),
                                z_rand.cuda(), (scfg.loss_regression_weight, scfg.loss_gan_weight,
                                                scfg.loss_reg_weight, scfg.loss_kld_weight))
    total.backward()
    c = comps.cpu()
    assert rel(total, loss_r) < TOL
    for i, k in ((1, 'huber'), (2, 'gen'), (3, 'div_reg'), (4, 'kld')):
        assert abs(float(c[i]) - float(comp[k])) < TOL * max(1.0, abs(float(comp[k]))), k
    assert abs(float(c[5]) - float(F.l1_loss(out, tgt))) < 1e-6
    assert abs(float(c[6]) - float(F.l1_loss(out_tri, tgt))) < 1e-6
    for a, b in zip(gq, req):
        assert rel(a.grad, b.grad) < TOL


def test_generator_loss_without_the_regulariser(S):
    """processor_v2.py:933-934 (z_type 'none' / loss_reg_weight 0): regression + GAN term alone.  The fused loss takes
    out_rand = None there and does not evaluate the divergence / KLD terms at all -- a log-variance whose exp() overflows
    leaves the loss finite, as upstream (with zero weights it was 0 * inf = NaN), and mu / log_var get no
    gradient from the loss."""
    ops = S['ops']
    g = torch.Generator().manual_seed(161)
    B, T, P = 5, 34, 27
    out, tgt = torch.randn(B, T, P, generator=g) * 0.3, torch.randn(B, T, P, generator=g) * 0.2
    out_tri = torch.randn(B, T, P, generator=g) * 0.2
    dis_out = torch.rand(B, 1, generator=g) * 0.9 + 0.05
    mu, lv = torch.randn(B, 16, generator=g), torch.full((B, 16), 200.0)        # exp(200) = inf in fp32
    scfg = O.StepCfg(loss_reg_weight=0.0)
    assert not O.uses_divergence_term(scfg)
    req = [t.clone().requires_grad_(True) for t in (out, dis_out)]
    loss_r, comp = O.gen_losses(scfg, req[0], tgt, req[1], None, None, None, mu, lv, True)
    loss_r.backward()
    gq = [t.cuda().requires_grad_(True) for t in (out, dis_out, mu, lv)]
    total, comps = ops.gen_loss(gq[0], gq[1], gq[2], gq[3], tgt.cuda(), out_tri.cuda(), None, None, None,
                                (scfg.loss_regression_weight, scfg.loss_gan_weight, 123.0, 456.0))   # [2:] ignored
    total.backward()
    c = comps.cpu()
    assert torch.isfinite(c).all() and rel(total, loss_r) < TOL
    assert abs(float(c[1]) - float(comp['huber'])) < TOL and abs(float(c[2]) - float(comp['gen'])) < TOL
    assert float(c[3]) == 0.0 and float(c[4]) == 0.0 and comp['div_reg'] is None and comp['kld'] is None
    assert abs(float(c[5]) - float(F.l1_loss(out, tgt))) < 1e-6 and abs(float(c[6]) - float(F.l1_loss(out_tri, tgt))) < 1e-6
    assert rel(gq[0].grad, req[0].grad) < TOL and rel(gq[1].grad, req[1].grad) < TOL
    assert gq[2].grad is None and gq[3].grad is None


def test_fused_adam_matches_torch_adam(S):
    optim = S['optim']
    g = torch.Generator().manual_seed(17)
    ps = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in ((7, 5), (13,), (3, 4, 2))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt_r = torch.optim.Adam(ref, lr=5e-4, betas=(0.5, 0.999))
    gp = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ps]
    arena = optim.ParamArena(gp)
    opt_g = optim.FusedAdam(arena, lr=5e-4, betas=(0.5, 0.999))
    for _ in range(5):
        opt_g.zero_grad()
        for p, q in zip(ref, gp):
            gr = torch.randn(p.shape, generator=g)
            p.grad = gr.clone()
            q.grad.add_(gr.cuda())
        opt_r.step()
        opt_g.step()
    for p, q in zip(ref, gp):
        assert rel(q, p) < 1e-5


@pytest.mark.parametrize('n_entries,dim,world,cap,n_tok', [(20000, 300, 8, 1152, 4352), (64, 5, 3, 40, 70), (1030, 300, 1, 64, 60)])
def test_touched_row_exchange_kernels(S, n_entries, dim, world, cap, n_tok):
    """csrc/rows.hip: sorted-unique row list (bit-exact vs torch.unique), [id | row] records, and the rank-ordered merge of
    `world` replicas' records (bit-exact vs a sequential sum in rank order; every replica touches the PAD row 0)."""
    ops = S['ops']
    ops.init_tickets(torch.device('cuda', 0))
    g = torch.Generator().manual_seed(n_entries + world)
    gathered = torch.zeros(world, cap, dim + 1, device='cuda')
    want = {}
    dense_r = []
    for r in range(world):
        ids = torch.randint(0, n_entries, (n_tok,), generator=g)
        ids[::3] = 0
        ids[1::7] = ids[2]                                           # repeats
        ids = ids[:max(1, n_tok // 12)].repeat(12)[:n_tok] if n_tok > cap else ids      # keep the unique count <= cap
        dense = torch.zeros(n_entries, dim)
        dense.index_add_(0, ids, torch.randn(ids.numel(), dim, generator=g))
        u = torch.unique(ids)
        assert u.numel() <= cap
        uids = torch.empty(cap, dtype=torch.int32, device='cuda')
        ops.rows_unique_raw(ids.cuda(), n_entries, uids)
        assert torch.equal(uids[:u.numel()].cpu().long(), u) and bool((uids[u.numel():] == n_entries).all())
        rec = torch.empty(cap, dim + 1, device='cuda')
        ops.rows_pack_raw(dense.cuda(), uids, rec)
        assert torch.equal(rec[:, 0].contiguous().view(torch.int32), uids)
        assert torch.equal(rec[:u.numel(), 1:].cpu(), dense[u]) and float(rec[u.numel():, 1:].abs().max() if u.numel() < cap else 0) == 0
        gathered[r] = rec
        for i in u.tolist():
            want[i] = dense[i].clone() if i not in want else want[i] + dense[i]
        dense_r.append(dense)
    out = dense_r[0].cuda().clone()                                   # replica 0's view: its own dense gradient
    ops.rows_merge_raw(gathered, out)
    out = out.cpu()
    touched = torch.zeros(n_entries, dtype=torch.bool)
    for i, v in want.items():
        assert torch.equal(out[i], v), i
        touched[i] = True
    assert torch.equal(out[~touched], dense_r[0][~touched])           # rows nobody listed are left alone
    assert ops.coop_gru_timeouts() == 0
    # capacity overflow is reported through the sticky error word (bit 1), loudly at the trainer's next read-back
    small = torch.empty(2, dtype=torch.int32, device='cuda')
    ops.rows_unique_raw(torch.arange(5, device='cuda'), n_entries, small)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match='row capacity'):
        ops.check_coop_flag(ops.coop_error_flag().cpu()[0].item())
    ops._COOP_FLAG[0].zero_()


@pytest.mark.parametrize('B,train,slots', [(5, True, False), (33, True, False), (33, True, True), (6, True, True), (4, False, False)])
def test_clip_resident_tcn_forward_fp32_equals_the_layer_by_layer_path(S, B, train, slots, monkeypatch):
    """csrc/tcn_fused32.hip (the four TemporalBlocks in one launch, fp32 rows in LDS, two-piece bf16 products) against the
    layer-by-layer kernels (conv_sp_k: the same two-piece products) on the same weights and noise stream: outputs and
    every parameter gradient (the backward pass is the layer-by-layer one on the tensors the fused forward leaves)."""
    import types
    ops = S['ops']
    if int(S['lib'].load().s2ag_gru_coop_split_pieces()) != 2:
        pytest.skip('the clip-resident fp32 TCN exists for the default two-piece products only')
    from speech2affective_gestures_amd import noise
    from speech2affective_gestures_amd.net.multimodal_context_net_v2 import TextEncoderTCN
    cfg = types.SimpleNamespace(hidden_size=300, n_layers=4, dropout_prob=0.3, freeze_wordembed=False)
    torch.manual_seed(3)
    noise.reset_sites(0)
    txt = TextEncoderTCN(cfg, 400, 300, dropout=0.3).cuda().train(train)
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(0, 400, (B, 34), generator=g)
    ids[:, 20:] = 0
    dt = torch.randn(B, 34, 32, generator=g).cuda()
    res = {}
    for fused in (False, True):
        monkeypatch.setattr(ops, 'TCN_FUSED32', fused)
        for p in txt.parameters():
            # gradient slots exist (as in the trainer's arenas): the kernels accumulate into them, which is also what
            # switches on the one-launch data-gradient chain + the one-launch weight gradients of the fused path
            p.grad = torch.zeros_like(p) if slots else None
        ops.begin_step()
        noise.manual_seed(5)
        assert ops.tcn_fused32_supported(34, 300, 2, 4) == fused
        t = txt(ids.cuda())[0]
        if train:
            (t * dt).sum().backward()
        torch.cuda.synchronize()
        res[fused] = (t.detach().clone(), {k: p.grad.clone() for k, p in txt.named_parameters()
                                            if '.net.' not in k and p.grad is not None})
    (t0, g0), (t1, g1) = res[False], res[True]
    assert rel(t1, t0) < 1e-4, rel(t1, t0)
    assert set(g0) == set(g1)
    for k in g0:
        # two summation orders of the same two-piece products differ by ~1e-6, so the few ReLU inputs that close to zero
        # take the other branch: single terms of single gradient entries come and go (one term of a bias-gradient entry is
        # ~1 % of the largest entry).  A flip in a late block changes what flows back through all earlier ones a little.  Kink-flip tolerant: relative L2 distance
        # (measured <= 1.1e-3) and a loose bound on the largest single entry.
        a, b = g1[k].double().flatten(), g0[k].double().flatten()
        l2 = float((a - b).norm() / b.norm().clamp_min(1e-12))
        assert l2 < 5e-3 and rel(g1[k], g0[k]) < 5e-2, (k, l2, rel(g1[k], g0[k]))


def test_step_helpers_pre_seq_snapshots_and_context_concat(S):
    """The small step-level kernels that replaced chains of ATen launches: s2ag_make_pre_seq, s2ag_rng_snapshots,
    s2ag_concat_cols (+ s2ag_sum_frames as the gradient of the broadcast speaker code) against their torch forms."""
    ops, noise = S['ops'], S['noise']
    g = torch.Generator().manual_seed(5)
    B, T, D, n_pre = 7, 34, 27, 4
    target = torch.randn(B, T, D, generator=g)
    ref = target.new_zeros(B, T, D + 1)
    ref[:, :n_pre, :-1] = target[:, :n_pre]
    ref[:, :n_pre, -1] = 1
    assert torch.equal(ops.make_pre_seq(target.cuda(), n_pre).cpu(), ref)
    # snapshots: n consecutive passes + derived offsets == n begin_pass calls (+ clone / add)
    noise.manual_seed(321)
    a = [noise.begin_pass('cuda').cpu() for _ in range(3)]
    noise.manual_seed(321)
    b = [t.cpu() for t in noise.begin_passes('cuda', 3, derived=(3, 6))]
    assert len(b) == 5 and all(torch.equal(x, y) for x, y in zip(a, b[:3]))
    assert b[3].tolist() == [a[0][0].item(), a[0][1].item() + 3] and b[4].tolist() == [a[0][0].item(), a[0][1].item() + 6]
    assert noise.begin_pass('cuda').cpu()[1].item() == a[0][1].item() + 3          # the counter advanced by n only
    # [pre | audio (a column slice of a wider tensor) | text | z over the frames]
    pre = torch.randn(B, T, 8, generator=g)
    wide = torch.randn(B, T, 40, generator=g)
    text = torch.randn(B, T, 32, generator=g)
    z = torch.randn(B, 16, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (pre, wide, text, z)]
    ref_out = torch.cat((leaves[0], leaves[1][..., 4:36], leaves[2], leaves[3].unsqueeze(1).expand(-1, T, -1)), dim=2)
    gl = [t.cuda().requires_grad_(True) for t in (pre, wide, text, z)]
    out = ops.context_cat((gl[0], gl[1][..., 4:36], gl[2]), gl[3])
    assert torch.equal(out.cpu(), ref_out.detach())
    dy = torch.randn(B, T, 88, generator=g)
    ref_out.backward(dy)
    out.backward(dy.cuda())
    for a_, b_ in zip(gl, leaves):
        assert rel(a_.grad, b_.grad) < 1e-6
    out2 = ops.context_cat((gl[0], gl[2]))                                          # no speaker code
    assert torch.equal(out2.cpu(), torch.cat((pre, text), dim=2))


def test_pose_metrics_kernel_matches_the_reference_golden(S, golden_dir):
    """s2ag_pose_metrics (L1, joint MAE behind the seed poses, acceleration difference in one launch, float64 like numpy
    upstream) against tests/golden/metrics.npz = the reference's own Processor.push_samples, and against the oracle on a
    full-size batch."""
    import sys
    sys.path.insert(0, golden_dir)
    from metrics_recipe import MEAN_DIR_VEC, N_BATCHES, N_PRE, metrics_inputs
    ops = S['ops']
    g = np.load(os.path.join(golden_dir, 'metrics.npz'))
    for b in range(N_BATCHES):
        out, tgt = metrics_inputs(b)
        got = ops.pose_metrics(torch.from_numpy(out).cuda(), torch.from_numpy(tgt).cuda(), MEAN_DIR_VEC, N_PRE).cpu().numpy()
        np.testing.assert_allclose(got, g['vals'][b], rtol=2e-6)       # the L1 term is a float32 mean upstream
    rs = np.random.RandomState(5)
    tgt = torch.from_numpy((rs.standard_normal((128, 136, 27)) * 0.2).astype(np.float32))
    out = tgt * 0.5 + torch.from_numpy((rs.standard_normal((128, 136, 27)) * 0.2).astype(np.float32))
    want = O.push_samples_metrics(out, tgt, MEAN_DIR_VEC, 4)
    got = ops.pose_metrics(out.cuda(), tgt.cuda(), MEAN_DIR_VEC, 4).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-6)
