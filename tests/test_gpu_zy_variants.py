"""Opt-in kernel VARIANTS (config.py: switches of the group 'opt-in kernel variants', default OFF; one csrc/*.hip file each so
that no default binary moves).  Written in r06 with GPU access closed: never timed -- tools/ab_variants.sh does that in one
process the day a GPU answers.  What is proven here (on the CPU device model now, on the MI355X with `-m gpu`): a variant
returns what the default kernel returns on the same inputs -- bit for bit where the arithmetic and its order are the same,
else to the tolerance the default kernel's own parity test uses -- and both agree with an fp64 / torch reference, so a
variant that wins its A/B can become the default without touching a single parity test.
The file name sorts behind the parity suite of the default path on purpose (and before the debug flavour's tests): under
`pytest -x` a surprise in a kernel that has never seen hardware must not cost the run of the tests that have."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def S():
    from speech2affective_gestures_amd import _lib as L
    from speech2affective_gestures_amd import config, ops
    return dict(L=L, lib=L.load(), config=config, ops=ops)


def _p(t):
    return C.c_void_p(t.data_ptr())


def _run_wgrad(S, jobs, nj, blocks):
    L, lib = S['L'], S['lib']
    need = int(lib.s2ag_f32_wgrad_tr_scratch_floats_n(jobs, nj, blocks))
    assert need > 0
    sc = torch.full((need,), float('nan'), device='cuda')          # every partial the reduce reads must have been written
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.s2ag_f32_wgrad_tr_n(jobs, nj, _p(sc), need, blocks, st), 'f32_wgrad_tr')
    torch.cuda.synchronize()


def _gru_jobs(S, B, T, H, In, seed):
    """The three weight-gradient jobs of one bidirectional GRU layer (ops.py _GruFn.backward): dW_ih of both directions
    (6H x In), dW_hh per direction with x = the layer's own output shifted by one frame (pos_off -1 / +1: the first / last
    frame of every clip multiplies zeros)."""
    L = S['L']
    g = torch.Generator().manual_seed(seed)
    H3 = 3 * H
    dgi = (torch.randn(B * T, 2 * H3, generator=g) * 0.3).cuda()
    dgh = (torch.randn(2, B * T, H3, generator=g) * 0.3).cuda()
    inp = torch.randn(B * T, In, generator=g).cuda()
    y = torch.randn(B * T, 2 * H, generator=g).cuda()

    def fresh():
        return dict(dw_ih=torch.zeros(2 * H3, In, device='cuda'), db_ih=torch.zeros(2 * H3, device='cuda'),
                    dw_hh=[torch.zeros(H3, H, device='cuda') for _ in range(2)],
                    db_hh=[torch.zeros(H3, device='cuda') for _ in range(2)])

    def jobs(o):
        j = (L.BF16Wgrad * 3)()
        j[0] = L.BF16Wgrad(_p(dgi), _p(inp), _p(o['dw_ih']), _p(o['db_ih']), B, T, T, T * In, In, 2 * H3, 1, 0, 0, 1, In, In,
                           2 * H3, In, In, 0, 1, 0, 1)
        for d in range(2):
            j[1 + d] = L.BF16Wgrad(C.c_void_p(dgh[d].data_ptr()), C.c_void_p(y.data_ptr() + 4 * d * H), _p(o['dw_hh'][d]),
                                   _p(o['db_hh'][d]), B, T, T, T * 2 * H, 2 * H, H3, 1, -1 if d == 0 else 1, 0, 1, H, H, H3, H, H,
                                   0, 1, 0, 1)
        return j

    def reference():
        r = dict(dw_ih=dgi.double().T @ inp.double(), db_ih=dgi.double().sum(0), dw_hh=[], db_hh=[])
        y3 = y.double().view(B, T, 2 * H)
        for d in range(2):
            prev = torch.zeros(B, T, H, dtype=torch.float64, device='cuda')
            if d == 0:
                prev[:, 1:] = y3[:, :-1, :H]
            else:
                prev[:, :-1] = y3[:, 1:, H:]
            r['dw_hh'].append(dgh[d].double().T @ prev.view(B * T, H))
            r['db_hh'].append(dgh[d].double().sum(0))
        return r
    return fresh, jobs, reference, (dgi, dgh, inp, y)


def _tcn_jobs(S, B, T, Cch, nb, seed):
    """The 2 * nb weight-gradient jobs of the clip-resident text TCN's backward (ops.py _Tcn32Fn.backward): two taps per
    conv, dilation 2^b (pos_off -d, pos_tap d: tap 0 reads frame t - d, zeros before the clip's first frame)."""
    L = S['L']
    g = torch.Generator().manual_seed(seed)
    rows = B * T
    gp = (torch.randn(2 * nb, rows, Cch, generator=g) * 0.2).cuda()
    xs = torch.randn(2 * nb, rows, Cch, generator=g).cuda()
    dils = [2 ** b for b in range(nb)]

    def fresh():
        return dict(dw=[torch.zeros(Cch, 2, Cch, device='cuda') for _ in range(2 * nb)],      # tap-major (Cout, ks, Cin)
                    db=[torch.zeros(Cch, device='cuda') for _ in range(2 * nb)])

    def jobs(o):
        j = (L.BF16Wgrad * (2 * nb))()
        for k in range(2 * nb):
            d = dils[k // 2]
            j[k] = L.BF16Wgrad(_p(gp[k]), _p(xs[k]), _p(o['dw'][k]), _p(o['db'][k]), B, T, T, T * Cch, Cch, Cch, 1, -d, d, 2,
                               Cch, Cch, Cch, Cch, 2 * Cch, Cch, 1, 0, 2)
        return j

    def reference():
        r = dict(dw=[], db=[])
        for k in range(2 * nb):
            d = dils[k // 2]
            x3, g3 = xs[k].double().view(B, T, Cch), gp[k].double().view(B, T, Cch)
            w = torch.zeros(Cch, 2, Cch, dtype=torch.float64, device='cuda')
            w[:, 1, :] = torch.einsum('bto,bti->oi', g3, x3)                       # tap 1: frame t
            w[:, 0, :] = torch.einsum('bto,bti->oi', g3[:, d:], x3[:, :T - d]) if d < T else 0.0     # tap 0: frame t - d
            r['dw'].append(w)
            r['db'].append(g3.sum((0, 1)))
        return r
    return fresh, jobs, reference, (gp, xs)


def _flat(o):
    out = []
    for k in sorted(o):
        v = o[k]
        out += [(f'{k}[{i}]', t) for i, t in enumerate(v)] if isinstance(v, list) else [(k, v)]
    return out


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / max(1e-30, float(b.abs().max())))


@pytest.mark.parametrize('ring', [1, 2])
@pytest.mark.parametrize('pieces', [2, 1])
@pytest.mark.parametrize('case', ['gru_h300_in600_b5', 'gru_h300_in32_b3', 'tcn_c300_b7', 'tcn_c300_T41_b4'])
def test_pipelined_weight_gradient_is_bit_identical(S, case, pieces, ring):
    """WGRAD32_PIPE (csrc/wgrad_tr32p.hip) against the default wgrad_tr32_k<160, 160> (csrc/wgrad_tr.hip) on the jobs the step
    really launches -- a GRU layer's three weight gradients (dW_hh with the one-frame shift, i.e. zero rows at clip edges) and
    the text TCN's dilated two-tap gradients (causal pad rows) -- with row counts that are NOT multiples of the kernels' 192-row
    period, channel counts that are not multiples of the 160-wide tiles (300, 600, 900, 1800: partial tiles, pad channels),
    several workgroups per tile (splits), and both product modes (2 bf16 pieces per operand / 1).  Same tiles, same pieces,
    same order of products and of steps: dw must be EQUAL, bit for bit.  The bias gradients are summed through LDS atomics
    whose order is not fixed on hardware: 1e-6.  Both are also held to the fp64 result at the default kernel's tolerance."""
    config, lib = S['config'], S['lib']
    if case.startswith('gru'):
        H, In, B = {'gru_h300_in600_b5': (300, 600, 5), 'gru_h300_in32_b3': (300, 32, 3)}[case]
        fresh, jobs, reference, keep = _gru_jobs(S, B, 34, H, In, 7100 + B)
        nj, blocks = 3, 0
    else:
        B, T = {'tcn_c300_b7': (7, 34), 'tcn_c300_T41_b4': (4, 41)}[case]
        fresh, jobs, reference, keep = _tcn_jobs(S, B, T, 300, 2, 7200 + B)
        nj, blocks = 4, 256                                    # as ops.TCN32_WGRAD_BLOCKS: the whole chip
    prev = lib.s2ag_gru_coop_split_pieces()
    with config.override('GRU_SPLIT', pieces):
        assert lib.s2ag_gru_coop_split_pieces() == pieces
        base = fresh()
        _run_wgrad(S, jobs(base), nj, blocks)
        with config.override('WGRAD32_PIPE', ring):
            assert lib.s2ag_get_option(b'WGRAD32_PIPE') == ring
            var = fresh()
            _run_wgrad(S, jobs(var), nj, blocks)
        assert lib.s2ag_get_option(b'WGRAD32_PIPE') == 0
    assert lib.s2ag_gru_coop_split_pieces() == prev
    ref = reference()
    tol = 2e-5 if pieces == 2 else 6e-3                        # 16 / 8 mantissa bits per operand, fp32 accumulation
    for (name, a), (_, b), (_, r) in zip(_flat(base), _flat(var), _flat(ref)):
        assert torch.isfinite(a).all() and torch.isfinite(b).all(), name
        if name.startswith('dw'):
            assert torch.equal(a, b), (name, _rel(b, a))
            assert _rel(b, r) < tol, (name, _rel(b, r))
        else:
            assert _rel(b, a) < 1e-6 and _rel(b, r) < 1e-5, (name, _rel(b, a), _rel(b, r))


def test_pipelined_weight_gradient_refuses_what_32_bit_offsets_cannot_address(S):
    """The variant addresses its operands with 32-bit byte offsets whose top bit means 'reads as zeros', and relies on a row
    past the last clip lying past num_records.  A job it cannot address that way (an operand of 2 GiB or more; a clip pitch
    smaller than the rows a clip may read, i.e. overlapping clips -- the case built here, no 2 GiB tensor is allocated) is
    run by the default kernel inside the library: the call succeeds and the result is the default's."""
    L, lib, config = S['L'], S['lib'], S['config']
    B, T, Cch = 3, 40, 320
    g = torch.Generator().manual_seed(99)
    gy = torch.randn(B * T, Cch, generator=g).cuda()
    x = torch.randn(B * T + 64, Cch, generator=g).cuda()
    outs = []
    for pipe in (0, 1):
        dw = torch.zeros(Cch, Cch, 1, device='cuda')
        j = (L.BF16Wgrad * 1)()
        # clip pitch T * Cch but Lin = T + 8 rows per clip are addressable (overlap): x_clip < Lin * ldx
        j[0] = L.BF16Wgrad(_p(gy), _p(x), _p(dw), None, B, T, T + 8, T * Cch, Cch, Cch, 1, 4, 0, 1, Cch, Cch, Cch, Cch, Cch, 1, 1, 0, 1)
        with config.override('WGRAD32_PIPE', pipe):
            _run_wgrad(S, j, 1, 0)
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])
    x3 = torch.stack([x[b * T + 4:b * T + 4 + T] for b in range(B)]).double()
    ref = torch.einsum('bto,bti->oi', gy.double().view(B, T, Cch), x3)
    assert _rel(outs[1][:, :, 0], ref) < 2e-5


# ---- TCN32_PAIR: two clips per workgroup in the clip-resident text TCN of the fp32 step (csrc/tcn32p.hip) -----------------------
def _tcn32_run(S, x, ws, bs, dils, drop_p, n_passes, save_clips, gy, seed=5):
    """One forward (s2ag_tcn32_fwd / _fwd_passes) + one data-gradient chain (s2ag_tcn32_bwd) through the C ABI, as
    ops._TcnFused32 drives them.  Returns every tensor the launches write."""
    L, lib = S['L'], S['lib']
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nb = len(dils)
    n_clips, T, Cch = x.shape
    B = n_clips // n_passes
    frag = torch.empty(int(lib.s2ag_tcn32_pack_elems(2 * nb)), dtype=torch.bfloat16, device='cuda')
    ptrs = (C.c_void_p * (2 * nb))(*[w.data_ptr() for w in ws])
    L.check(lib.s2ag_tcn32_pack(ptrs, 2 * nb, Cch, _p(frag), st), 'tcn32_pack')
    rows = save_clips * T
    nan = float('nan')
    saved = torch.full((3 * nb - 1, rows, Cch), nan, device='cuda')
    y_last = torch.full((n_clips * T, Cch), nan, device='cuda')
    a = L.Tcn32()
    a.x, a.wfrag = x.data_ptr(), frag.data_ptr()
    for b in range(nb):
        a.h1[b], a.h2[b] = saved[3 * b].data_ptr(), saved[3 * b + 1].data_ptr()
        a.y[b] = saved[3 * b + 2].data_ptr() if b < nb - 1 else y_last.data_ptr()
        a.dil[b] = int(dils[b])
        for j in range(2):
            a.bias[2 * b + j] = bs[2 * b + j].data_ptr()
            a.site[2 * b + j] = 40 + 2 * b + j
    a.n_blocks, a.n_clips, a.T, a.C = nb, n_clips, T, Cch
    a.drop_p = float(drop_p)
    noises = [torch.tensor([seed, 10 + k], dtype=torch.int64, device='cuda') for k in range(n_passes)]
    keep = torch.zeros(max(16, int(lib.s2ag_tcn32_keep_bytes(n_clips, nb))), dtype=torch.uint8, device='cuda')
    a.rng, a.keep = noises[0].data_ptr(), keep.data_ptr()
    if n_passes == 1:
        L.check(lib.s2ag_tcn32_fwd(C.byref(a), st), 'tcn32_fwd')
    else:
        rngs = (C.c_void_p * n_passes)(*[nz.data_ptr() for nz in noises])
        L.check(lib.s2ag_tcn32_fwd_passes(C.byref(a), n_passes, rngs, save_clips, st), 'tcn32_fwd_passes')
    torch.cuda.synchronize()
    out = dict(saved=saved.clone(), y_last=y_last.clone())
    # backward over the clips that kept their activations
    gx = torch.full((rows, Cch), nan, device='cuda')
    gp = torch.full((2 * nb, rows, Cch), nan, device='cuda')
    b_ = L.Tcn32()
    b_.wfrag = frag.data_ptr()
    for b in range(nb):
        b_.h1[b], b_.h2[b] = saved[3 * b].data_ptr(), saved[3 * b + 1].data_ptr()
        b_.y[b] = saved[3 * b + 2].data_ptr() if b < nb - 1 else y_last.data_ptr()
        b_.gp1[b], b_.gp2[b] = gp[2 * b].data_ptr(), gp[2 * b + 1].data_ptr()
        b_.dil[b] = int(dils[b])
    b_.n_blocks, b_.n_clips, b_.T, b_.C = nb, save_clips, T, Cch
    b_.drop_p = float(drop_p)
    b_.gy, b_.gx = gy.data_ptr(), gx.data_ptr()
    L.check(lib.s2ag_tcn32_bwd(C.byref(b_), st), 'tcn32_bwd')
    torch.cuda.synchronize()
    out.update(gx=gx, gp=gp)
    return out


@pytest.mark.parametrize('ncl_switch', [1, 2])
@pytest.mark.parametrize('case', ['B4_T34', 'B5_T34_odd', 'B2_T40_nodrop', 'B3_T34_two_blocks', 'lockstep_3x4', 'lockstep_3x3'])
def test_pair_tcn_is_bit_identical(S, case, ncl_switch):
    """TCN32_PAIR (csrc/tcn32p.hip: two clips per workgroup, activations in LDS as bf16 hi / lo planes split once by their
    producer, residual and running gradient in registers) against the default clip-resident kernels (csrc/tcn_fused32.hip)
    through the same C entry points: everything the forward leaves (h1, h2, y of every block, the last y of every lockstep
    pass) and everything the data-gradient chain leaves (gp1 / gp2 of every conv = the weight gradients' operands, gx) must be
    EQUAL bit for bit -- same pieces, same K order, same fp32 epilogue, same keep bits.  Covers an odd batch (a workgroup
    with a single clip), T = 40 (the kernels' limit), no dropout, two blocks, and the trainer's lockstep batch of three passes
    with only the first pass saving.  `ncl_switch` 2 = the same kernel with ONE clip per workgroup (TCN32_PAIR=2); `lockstep_3x3`
    (an odd number of clips per pass) is taken by the one-clip form and falls back to the default kernel in the pair form."""
    config, lib = S['config'], S['lib']
    B, T, nb, drop, nP = {'B4_T34': (4, 34, 4, 0.3, 1), 'B5_T34_odd': (5, 34, 4, 0.3, 1), 'B2_T40_nodrop': (2, 40, 4, 0.0, 1),
                          'B3_T34_two_blocks': (3, 34, 2, 0.3, 1), 'lockstep_3x4': (4, 34, 4, 0.3, 3),
                          'lockstep_3x3': (3, 34, 4, 0.3, 3)}[case]
    Cch = 300
    assert lib.s2ag_tcn32_supported(T, Cch, 2)
    g = torch.Generator().manual_seed(8100 + B + T)
    x = torch.randn(nP * B, T, Cch, generator=g).cuda()
    ws = [(torch.randn(Cch, 2, Cch, generator=g) * (0.6 / (2 * Cch) ** 0.5)).cuda() for _ in range(2 * nb)]
    bs = [(torch.randn(Cch, generator=g) * 0.1).cuda() for _ in range(2 * nb)]
    gy = (torch.randn(B * T, Cch, generator=g) * 0.1).cuda()
    dils = [2 ** b for b in range(nb)]
    base = _tcn32_run(S, x, ws, bs, dils, drop, nP, B, gy)
    with config.override('TCN32_PAIR', ncl_switch):
        assert lib.s2ag_get_option(b'TCN32_PAIR') == ncl_switch
        var = _tcn32_run(S, x, ws, bs, dils, drop, nP, B, gy)
    assert lib.s2ag_get_option(b'TCN32_PAIR') == 0
    for k in base:
        assert torch.isfinite(base[k]).all(), k           # (every element written by the default ...)
        assert torch.equal(base[k], var[k]), (k, _rel(var[k], base[k]))      # ... and the same bits by the variant
    # and the forward is the TemporalConvNet it claims to be: fp64 reference with the kernels' own keep bits (drop 0: exact path)
    if drop == 0.0:
        xr = x.double()
        for b in range(nb):
            d = dils[b]
            def conv(inp, w, bias):
                prev = torch.zeros_like(inp)
                prev[:, d:] = inp[:, :T - d]
                return torch.einsum('bti,oi->bto', prev, w[:, 0].double()) + torch.einsum('bti,oi->bto', inp, w[:, 1].double()) + bias.double()
            h1 = torch.relu(conv(xr, ws[2 * b], bs[2 * b]))
            h2 = torch.relu(conv(h1, ws[2 * b + 1], bs[2 * b + 1]))
            xr = torch.relu(h2 + xr)
        assert _rel(var['y_last'].view(nP * B, T, Cch), xr) < 1e-4


# ---- BN_FOLD_APPLY: statistics fold + normalise + LeakyReLU in one launch (csrc/bn_foldapply.hip) ----------------------------
@pytest.mark.parametrize('case', ['mfcc_64ch', 'mfcc_48ch_k3', 'time_steps_34ch', 'repeat3'])
def test_bn_fold_apply_in_one_launch(S, case):
    """BN_FOLD_APPLY against the default pair bn_fold_k + bn_apply_k on the real call chain -- ops.conv1d_nlc(bn_stats=True)
    leaves its column sums, ops.batch_norm_act consumes them -- at MFCCEncoder's shapes (net/multimodal_context_net_v2.py:39-48):
    output, running estimates, batch counter, and every gradient of a loss through the layer (the backward pass reads the
    coefficient vectors the forward stored) against the default path (1e-6: the default sums the partial rows through LDS
    atomics in arrival order, the variant in a fixed order) and against torch's own batch_norm (1e-5).  `repeat3`: the
    shared-encoder case, three passes of a step advance the running estimates in one launch (ops.bn_repeat)."""
    ops, config = S['ops'], S['config']
    import torch.nn.functional as F
    B, Lpos, Cin, Cout, ks, rep = {'mfcc_64ch': (6, 37, 64, 64, 5, 1), 'mfcc_48ch_k3': (5, 37, 64, 48, 3, 1),
                                   'time_steps_34ch': (7, 37, 48, 34, 3, 1), 'repeat3': (4, 37, 64, 64, 5, 3)}[case]
    g = torch.Generator().manual_seed(9100 + Cout + B)
    x = torch.randn(B, Lpos, Cin, generator=g)
    w = torch.randn(Cout, Cin, ks, generator=g) * (1.0 / (Cin * ks) ** 0.5)
    b = torch.randn(Cout, generator=g) * 0.1
    gw, gb = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2
    dy = torch.randn(B, Lpos, Cout, generator=g)

    def run(on):
        bn = torch.nn.BatchNorm1d(Cout).cuda().train()
        with torch.no_grad():
            bn.weight.copy_(gw)
            bn.bias.copy_(gb)
        xg, wg, bg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        with config.override('BN_FOLD_APPLY', on), ops.bn_repeat(rep):
            assert ops.BN_FOLD_APPLY == bool(on)
            z = ops.conv1d_nlc(xg, wg, bg, pad=ks // 2, bn_stats=True, tm_copy=True)      # as MFCCEncoder.forward calls it
            assert getattr(z, '_s2ag_stats', None) is not None       # the path under test: sums left by the conv
            y = ops.batch_norm_act(z, bn, slope=0.3)
        y.backward(dy.cuda())
        torch.cuda.synchronize()
        return dict(y=y.detach(), rm=bn.running_mean.clone(), rv=bn.running_var.clone(), nbt=int(bn.num_batches_tracked),
                    dx=xg.grad, dw=wg.grad, db=bg.grad, dgamma=bn.weight.grad, dbeta=bn.bias.grad)
    base, var = run(False), run(True)
    assert base['nbt'] == var['nbt'] == rep
    for k in base:
        if k == 'db':        # a bias in front of a BatchNorm has NO gradient (the mean is subtracted): rounding residue of ~1e-6 x |dy| sums
            assert float(var[k].abs().max()) < 1e-4 and float(base[k].abs().max()) < 1e-4
        elif k != 'nbt':
            # forward quantities: 1e-6 (fp64 sums in a different order, rounded to fp32).  Gradients: both runs accumulate
            # weight / gamma / beta gradients through fp32 atomics whose order is not fixed on hardware (run-to-run ~3e-7 of the
            # largest element): 5e-6
            tol_k = 1e-6 if k in ('y', 'rm', 'rv') else 5e-6
            assert torch.isfinite(var[k]).all() and _rel(var[k], base[k]) < tol_k, (k, _rel(var[k], base[k]))
    # torch reference (channels-first), one pass
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    bnr = torch.nn.BatchNorm1d(Cout).double().train()
    with torch.no_grad():
        bnr.weight.copy_(gw)
        bnr.bias.copy_(gb)
    zr = F.conv1d(xr.transpose(1, 2), wr, br, padding=ks // 2)
    yr = F.leaky_relu(bnr(zr), 0.3).transpose(1, 2)
    for _ in range(rep - 1):
        bnr(zr.detach())
    yr.backward(dy.double())
    # (the large products of the step carry 16 mantissa bits: the same tolerance as the default path's own test)
    tol = 3e-4
    assert _rel(var['y'], yr.detach()) < tol and _rel(var['rm'], bnr.running_mean) < tol and _rel(var['rv'], bnr.running_var) < tol
    for k, r in (('dx', xr.grad), ('dw', wr.grad), ('dgamma', bnr.weight.grad), ('dbeta', bnr.bias.grad)):
        assert _rel(var[k], r) < 5 * tol, (k, _rel(var[k], r))


# ---- EMB_BWD_ROWS: 256-row blocks, the PAD row summed in registers (csrc/emb_rows.hip) -----------------------------------------
@pytest.mark.parametrize('case', ['transcripts_B9', 'all_pad', 'all_distinct', 'dropout_B5', 'strided_rows'])
def test_row_block_embedding_backward(S, case):
    """EMB_BWD_ROWS against the default embedding_bwd_k through ops.embedding's own backward (gradient scattered into a dense
    (n_words, 300) table; net/multimodal_context_net_v2.py:84) and against index_add_ in fp64: TED-shaped transcripts (85 % PAD =
    id 0, a few words per clip, some repeated), the all-PAD and all-distinct extremes, the dropout-scaled form (same keep bits
    as the forward), a row count that is not a multiple of the 256-row block, and a gradient with a row pitch (a column slice
    of a wider matrix).  Both kernels accumulate through fp32 atomics in different groupings: 2e-6 of the largest element."""
    ops, config, lib = S['ops'], S['config'], S['lib']
    B, T, n_words, dim, p = {'transcripts_B9': (9, 34, 500, 300, 0.0), 'all_pad': (4, 34, 50, 300, 0.0),
                             'all_distinct': (3, 34, 400, 300, 0.0), 'dropout_B5': (5, 34, 300, 300, 0.1),
                             'strided_rows': (8, 33, 200, 300, 0.0)}[case]
    g = torch.Generator().manual_seed(9500 + B)
    if case == 'all_pad':
        ids = torch.zeros(B, T, dtype=torch.int64)
    elif case == 'all_distinct':
        ids = torch.randperm(n_words, generator=g)[:B * T].view(B, T)
    else:
        ids = torch.randint(1, n_words, (B, T), generator=g)
        ids[torch.rand(B, T, generator=g) < 0.85] = 0
        ids[:, 3] = 7                                            # one word in every clip: the same row from many workgroups
    table = torch.randn(n_words, dim, generator=g)
    wide = torch.randn(B * T, dim + 20, generator=g)
    dy = wide[:, 4:4 + dim] if case == 'strided_rows' else wide[:, :dim].contiguous()
    noise = torch.tensor([77, 5], dtype=torch.int64, device='cuda')

    def run(on):
        tg = table.cuda().requires_grad_(True)
        with config.override('EMB_BWD_ROWS', on):
            assert lib.s2ag_get_option(b'EMB_BWD_ROWS') == on
            y = ops.embedding(ids.cuda(), tg, p, noise, 31)
            wc = wide.cuda()                                    # (sliced ON the device: a host-side slice would arrive contiguous)
            dyc = wc[:, 4:4 + dim] if case == 'strided_rows' else wc[:, :dim].contiguous()
            assert dyc.is_contiguous() == (case != 'strided_rows')
            y.backward(dyc.view(B, T, dim))
        torch.cuda.synchronize()
        return y.detach(), tg.grad
    (y0, g0), (y1, g1) = run(0), run(1)
    assert torch.equal(y0, y1)
    mask = torch.ones(B * T, dim, dtype=torch.float64)
    if p > 0:
        mask = ops.dropout_mask(noise, 31, p, (B * T, dim)).cpu().double()       # the keep / scale factors the kernels draw
    ref = torch.zeros(n_words, dim, dtype=torch.float64).index_add_(0, ids.view(-1), dy.double() * mask)
    assert _rel(g1, g0) < 2e-6 and _rel(g1, ref) < 2e-6 and _rel(g0, ref) < 2e-6, (_rel(g1, g0), _rel(g1, ref))
    touched = torch.zeros(n_words, dtype=torch.bool)
    touched[ids.view(-1)] = True
    assert float(g1[~touched.cuda()].abs().max() if (~touched).any() else 0.0) == 0.0      # untouched rows stay exactly zero


@pytest.mark.parametrize('cols,nchan,prow', [(48, 16, 7), (480, 48, 5), (300, 300, 9)])
def test_bn_fold_apply_with_a_column_to_channel_map(S, cols, nchan, prow):
    """The ST-GCN blocks' BatchNorm2d sees (N * T, V * C) rows: several COLUMNS belong to one channel (ops.batch_norm_act's
    chan_map).  The folded convs in front of them leave column sums like any other conv; the one-launch fold + apply must sum
    them per channel (columns of a channel in ascending order) exactly like bn_fold_k.  Partials are built here from the input
    itself (row blocks of uneven length), both paths consume the same ones; reference: torch batch_norm over the channel axis."""
    ops, config = S['ops'], S['config']
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(9700 + cols)
    rows = 204
    x = torch.randn(rows, cols, generator=g) * 1.5 + 0.3
    per = cols // nchan
    cmap = (torch.arange(cols) % nchan).to(torch.int32) if per > 1 else None        # column c -> channel c mod nchan (interleaved)
    cuts = sorted(torch.randperm(rows - 1, generator=g)[:prow - 1].add(1).tolist())
    blocks = [x[a:b].double() for a, b in zip([0] + cuts, cuts + [rows])]
    part = torch.stack([torch.stack([b.sum(0) for b in blocks]), torch.stack([(b * b).sum(0) for b in blocks])]).contiguous()
    gw, gb = torch.rand(nchan, generator=g) + 0.5, torch.randn(nchan, generator=g) * 0.2
    dy = torch.randn(rows, cols, generator=g)

    def run(on):
        bn = torch.nn.BatchNorm1d(nchan).cuda().train()
        with torch.no_grad():
            bn.weight.copy_(gw)
            bn.bias.copy_(gb)
        xg = x.cuda().requires_grad_(True)
        xin = xg * 1.0                                            # a non-leaf tensor carries the attribute, as a conv output does
        xin._s2ag_stats = (part.cuda().view(-1).clone(), prow)
        with config.override('BN_FOLD_APPLY', on):
            y = ops.batch_norm_act(xin, bn, slope=0.3, chan_map=None if cmap is None else cmap.cuda())
        y.backward(dy.cuda())
        torch.cuda.synchronize()
        return dict(y=y.detach(), rm=bn.running_mean.clone(), rv=bn.running_var.clone(), dx=xg.grad, dgamma=bn.weight.grad,
                    dbeta=bn.bias.grad, nbt=int(bn.num_batches_tracked))
    base, var = run(False), run(True)
    assert base['nbt'] == var['nbt'] == 1
    for k in base:
        if k != 'nbt':
            assert _rel(var[k], base[k]) < (1e-6 if k in ('y', 'rm', 'rv') else 5e-6), (k, _rel(var[k], base[k]))
    # torch: the columns of a channel are extra 'spatial' positions of that channel
    xr = x.double().requires_grad_(True)
    xc = xr.view(rows, per, nchan).permute(0, 2, 1) if per > 1 else xr.view(rows, nchan, 1)
    bnr = torch.nn.BatchNorm1d(nchan).double().train()
    with torch.no_grad():
        bnr.weight.copy_(gw)
        bnr.bias.copy_(gb)
    yr = F.leaky_relu(bnr(xc), 0.3)
    yr = yr.permute(0, 2, 1).reshape(rows, cols) if per > 1 else yr.view(rows, cols)
    yr.backward(dy.double())
    assert _rel(var['y'], yr.detach()) < 1e-5 and _rel(var['rm'], bnr.running_mean) < 1e-5 and _rel(var['rv'], bnr.running_var) < 1e-5
    assert _rel(var['dx'], xr.grad) < 1e-4 and _rel(var['dgamma'], bnr.weight.grad) < 1e-4 and _rel(var['dbeta'], bnr.bias.grad) < 1e-4
