"""Opt-in kernel VARIANTS (config.py: switches of the group 'opt-in kernel variants', default OFF; one csrc/*.hip file each so
that no default binary moves).  Written in r06 with GPU access closed: never timed -- tools/ab_variants.sh does that in one
process the day a GPU answers.  What is proven here (on the CPU device model now, on the MI355X with `-m gpu`): a variant
returns what the default kernel returns on the same inputs -- bit for bit where the arithmetic and its order are the same,
else to the tolerance the default kernel's own parity test uses -- and both agree with an fp64 / torch reference, so a
variant that wins its A/B can become the default without touching a single parity test."""
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
