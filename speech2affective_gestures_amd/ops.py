"""Autograd-level operators of the S2AG step, each a thin ``torch.autograd.Function`` whose forward and
backward call the C ABI of ``libs2ag_hip.so`` (include/s2ag_hip.h) on torch's current HIP stream.

PyTorch is plumbing here: it owns device memory (caching allocator, so everything is hipGraph-capturable),
streams and the autograd tape.  Every arithmetic operation is one of our gfx950 kernels.  There is no CPU
path: a non-CUDA tensor raises.

Layout: all activations are fp32 channels-last, i.e. a (clips, frames, channels) tensor is treated as a
row matrix; column slices of wider matrices are passed by pointer + row pitch, never copied.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from . import config

Tensor = torch.Tensor


def _lib():
    return L.load()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('S2AG ops run on MI355X only (HIP kernels); got a CPU tensor and there is no CPU '
                               'fallback by design')


def as_rows(t: Tensor) -> Tuple[Tensor, int, int, int]:
    """(tensor, rows, cols, ld): view ``t`` as a row matrix with unit column stride and uniform row pitch.
    Copies only if the leading dims do not collapse (never for the slices this package creates)."""
    if t.dtype != torch.float32:
        raise TypeError(f'expected float32, got {t.dtype}')
    if t.dim() == 0:
        t = t.reshape(1, 1)
    if t.dim() == 1:
        t = t.unsqueeze(0)
    cols = t.shape[-1]
    rows = t.numel() // max(cols, 1)
    ok = (t.stride(-1) == 1 or cols == 1) and t.numel() > 0
    ld = cols
    if ok:
        ld = t.stride(-2) if t.shape[-2] > 1 else cols
        exp = ld
        for d in range(t.dim() - 2, -1, -1):
            if t.shape[d] > 1 and t.stride(d) != exp:
                ok = False
                break
            exp *= t.shape[d]
        if ld < cols:
            ok = False
    if not ok:
        t = t.contiguous()
        ld = cols
    return t, rows, cols, ld


def _epi(act=L.ACT_NONE, slope=1.0, drop_p=0.0, noise: Optional[Tensor] = None, site=0):
    return L.Epilogue(act, float(slope), float(drop_p), _p(noise) if drop_p > 0 else None, int(site))


def _geom(N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, ldx, ldy, wtm=0):
    return L.ConvGeom(N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, ldx, ldy, int(wtm))


# ----------------------------------------------------------------------------------------------------
# raw (non-autograd) launch helpers
# ----------------------------------------------------------------------------------------------------
def conv_fwd_raw(x: Tensor, w: Tensor, bias: Optional[Tensor], y: Tensor, N, Lin, Lout, Cin, Cout, ks, stride, pad,
                 dil, act=L.ACT_NONE, slope=1.0, drop_p=0.0, noise=None, site=0, wtm=0, stats_out=None):
    x, xr, xc, ldx = as_rows(x)
    _, yr, yc, ldy = as_rows(y)
    assert xr == N * Lin and xc == Cin, (xr, xc, N, Lin, Cin)
    assert yr == N * Lout and yc == Cout, (yr, yc, N, Lout, Cout)
    assert w.is_contiguous() and w.numel() == Cout * Cin * ks
    g = _geom(N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, ldx, ldy, wtm)
    e = _epi(act, slope, drop_p, noise, site)
    lib = _lib()
    split = (SPLIT_CONV and (wtm or ks == 1) and (ks > 1 or SPLIT_LINEAR) and stride == 1 and Lin == Lout and Cin % 4 == 0
             and 2.0 * N * Lout * Cout * Cin * ks >= SPLIT_CONV_MIN_FLOPS and lib.s2ag_gru_coop_split_pieces() != 0)
    wp = None
    if split:
        # big stride-1 tap-major convs (the TCN, the folded ST-GCN convs): bf16-pipe kernel, weight planes cached on the
        # (derived) weight tensor
        wp = getattr(w, '_s2ag_wp', None)
        if wp is None or wp[0] != w._version:
            wp = (w._version, split_planes_raw(w.detach().view(Cout * ks, Cin)))
            w._s2ag_wp = wp
    if stats_out is None:
        if split:
            rc = lib.s2ag_conv1d_nlc_fwd_split(_p(x), _p(wp[1]), _p(bias), _p(y), C.byref(g), C.byref(e), None, None,
                                               _stream())
            if rc != L.E_UNSUPPORTED:
                L.check(rc, 'conv_fwd_split')
                return None
        L.check(lib.s2ag_conv1d_nlc_fwd(_p(x), _p(w), _p(bias), _p(y), C.byref(g), C.byref(e), _stream()), 'conv_fwd')
        return None
    # the layer feeds a training-mode BatchNorm: let the kernel leave per-row-block column sums behind
    rows = lib.s2ag_conv_stats_rows(C.byref(g))
    part = torch.empty(2 * rows * Cout, dtype=torch.float64, device=y.device)
    got = C.c_int(0)
    if split:
        rc = lib.s2ag_conv1d_nlc_fwd_split(_p(x), _p(wp[1]), _p(bias), _p(y), C.byref(g), C.byref(e), _p(part),
                                           C.byref(got), _stream())
        if rc != L.E_UNSUPPORTED:
            L.check(rc, 'conv_fwd_split')
            return (part[:2 * got.value * Cout], got.value) if got.value > 0 else None
    L.check(lib.s2ag_conv1d_nlc_fwd_stats(_p(x), _p(w), _p(bias), _p(y), C.byref(g), C.byref(e), _p(part),
                                          C.byref(got), _stream()), 'conv_fwd_stats')
    return (part[:2 * got.value * Cout], got.value) if got.value > 0 else None


def conv_bwd_data_raw(gy: Tensor, w: Tensor, dx: Tensor, N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, accumulate,
                      wtm=0):
    gy, gr, gc, ldg = as_rows(gy)
    _, xr, xc, ldx = as_rows(dx)
    assert gr == N * Lout and gc == Cout and xr == N * Lin and xc == Cin
    g = _geom(N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, ldx, ldg, wtm)
    L.check(_lib().s2ag_conv1d_nlc_bwd_data(_p(gy), _p(w), _p(dx), C.byref(g), int(accumulate), _stream()),
            'conv_bwd_data')


def conv_bwd_weight_raw(gy: Tensor, x: Tensor, dw: Tensor, N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, accumulate,
                        wtm=0, dbias: Optional[Tensor] = None):
    """dw (+)= gy^T x_windows; ``dbias`` (+)= column sums of gy in the same launch (the gy tiles pass through the
    kernel anyway -- a separate bias-gradient reduction was 69 launches per step)."""
    gy, gr, gc, ldg = as_rows(gy)
    x, xr, xc, ldx = as_rows(x)
    assert gr == N * Lout and gc == Cout and xr == N * Lin and xc == Cin, (gr, gc, xr, xc)
    assert dw.is_contiguous() and dw.numel() == Cout * Cin * ks
    assert dbias is None or (dbias.is_contiguous() and dbias.numel() == Cout)
    g = _geom(N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, ldx, ldg, wtm)
    L.check(_lib().s2ag_conv1d_nlc_bwd_weight(_p(gy), _p(x), _p(dw), _p(dbias), C.byref(g), int(accumulate),
                                              _stream()), 'conv_bwd_weight')


def conv_bwd_weight_multi_raw(jobs) -> bool:
    """jobs: list of (gy, x, dw, dbias, (N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, wtm)) -- accumulating weight
    (+ bias) gradients, all in ONE launch.  False (nothing launched) if a job is outside the straight-line kernel."""
    arr = (L.WgradJob * len(jobs))()
    keep = []
    for k, (gy, x, dw, dbias, (N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, wtm)) in enumerate(jobs):
        gy, gr, gc, ldg = as_rows(gy)
        x, xr, xc, ldx = as_rows(x)
        assert gr == N * Lout and gc == Cout and xr == N * Lin and xc == Cin, (gr, gc, xr, xc)
        assert dw.is_contiguous() and dw.numel() == Cout * Cin * ks
        assert dbias is None or (dbias.is_contiguous() and dbias.numel() == Cout)
        keep += [gy, x]
        arr[k] = L.WgradJob(gy.data_ptr(), x.data_ptr(), dw.data_ptr(), dbias.data_ptr() if dbias is not None else None,
                            _geom(N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, ldx, ldg, wtm))
    rc = _lib().s2ag_conv1d_nlc_bwd_weight_multi(arr, len(jobs), _stream())
    if rc == L.E_UNSUPPORTED:
        return False
    L.check(rc, 'conv_bwd_weight_multi')
    return True


FUSE_WGRADS = True            # (constants below: A/B switches of r01-r03 whose alternative lost; tests may set them)
FUSE_BWD_PAIR = True


def conv_bwd_pair_raw(gy: Tensor, w: Tensor, x: Tensor, dx: Tensor, dw: Tensor, dbias: Optional[Tensor], N, Lin, Lout,
                      Cin, Cout, ks, stride, pad, dil, wtm) -> bool:
    """dx = data gradient, dw / dbias += weight / bias gradient of one stride-1 layer in ONE launch.  False (nothing
    launched) where the geometry is outside the straight-line kernels: the caller issues the two separate calls."""
    if stride != 1 or Lin != Lout:
        return False
    gy, gr, gc, ldg = as_rows(gy)
    x, xr, xc, ldx = as_rows(x)
    _, dr, dc, lddx = as_rows(dx)
    assert gr == N * Lout and gc == Cout and xr == N * Lin and xc == Cin and dr == xr and dc == Cin and lddx == ldx
    assert dw.is_contiguous() and dw.numel() == Cout * Cin * ks and w.is_contiguous()
    g = _geom(N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, ldx, ldg, wtm)
    rc = _lib().s2ag_conv1d_nlc_bwd_pair(_p(gy), _p(w), _p(x), _p(dx), _p(dw), _p(dbias), C.byref(g), _stream())
    if rc == L.E_UNSUPPORTED:
        return False
    L.check(rc, 'conv_bwd_pair')
    return True


def colsum_raw(x: Tensor, out: Tensor, sq: Optional[Tensor] = None, accumulate=False):
    x, r, c, ld = as_rows(x)
    L.check(_lib().s2ag_colsum(_p(x), r, c, ld, _p(out), _p(sq), int(accumulate), _stream()), 'colsum')


def add_act_raw(a: Tensor, b: Optional[Tensor], y: Tensor, slope: float):
    a, r, c, lda = as_rows(a)
    ldb = 0
    if b is not None:
        b, rb, cb, ldb = as_rows(b)
        assert (rb, cb) == (r, c)
    _, ry, cy, ldy = as_rows(y)
    assert (ry, cy) == (r, c)
    L.check(_lib().s2ag_add_act(_p(a), lda, _p(b), ldb, _p(y), ldy, r, c, float(slope), _stream()), 'add_act')


def epilogue_bwd_raw(dy: Tensor, y: Optional[Tensor], g: Tensor, act, slope, drop_p, noise, site):
    dy, r, c, lddy = as_rows(dy)
    ldy = 0
    if y is not None:
        y, _, _, ldy = as_rows(y)
    _, _, _, ldg = as_rows(g)
    e = _epi(act, slope, drop_p, noise, site)
    L.check(_lib().s2ag_epilogue_bwd(_p(dy), lddy, _p(y), ldy, _p(g), ldg, r, c, C.byref(e), _stream()),
            'epilogue_bwd')


def transpose_raw(src: Tensor, dst: Tensor):
    assert src.is_contiguous() and dst.is_contiguous() and src.dim() == 2
    L.check(_lib().s2ag_transpose(_p(src), src.shape[0], src.shape[1], _p(dst), _stream()), 'transpose')


# ----------------------------------------------------------------------------------------------------
# scheduling diagnostics: device wall-clock stamps that can sit inside a captured graph
# ----------------------------------------------------------------------------------------------------
_STAMPS = {'buf': None, 'names': []}
TRACE_PHASES = False


def stamp(name: str) -> None:
    """If ops.TRACE_PHASES: record the device wall clock when the current stream reaches this point."""
    if not TRACE_PHASES:
        return
    if _STAMPS['buf'] is None:
        _STAMPS['buf'] = torch.zeros(256, dtype=torch.int64, device='cuda')
    names = _STAMPS['names']
    if name not in names:
        names.append(name)
    i = names.index(name)
    L.check(_lib().s2ag_timestamp(C.c_void_p(_STAMPS['buf'].data_ptr() + 8 * i), _stream()), 'timestamp')


def read_stamps():
    """{name: microseconds since the earliest stamp} (synchronises)."""
    if _STAMPS['buf'] is None:
        return {}
    torch.cuda.synchronize()
    v = _STAMPS['buf'][:len(_STAMPS['names'])].tolist()
    t0 = min(v)
    return {n: (x - t0) / 100.0 for n, x in zip(_STAMPS['names'], v)}


# ----------------------------------------------------------------------------------------------------
# noise materialisers (tests / debugging)
# ----------------------------------------------------------------------------------------------------
def dropout_mask(noise: Tensor, site: int, p: float, shape: Sequence[int]) -> Tensor:
    out = torch.empty(tuple(shape), dtype=torch.float32, device=noise.device)
    L.check(_lib().s2ag_dropout_mask(_p(noise), int(site), float(p), out.numel(), _p(out), _stream()), 'dropout_mask')
    return out


def normal_noise(noise: Tensor, site: int, shape: Sequence[int]) -> Tensor:
    out = torch.empty(tuple(shape), dtype=torch.float32, device=noise.device)
    L.check(_lib().s2ag_normal_noise(_p(noise), int(site), out.numel(), _p(out), _stream()), 'normal_noise')
    return out


# ----------------------------------------------------------------------------------------------------
# ticket words of the "last block finalises" reductions
# ----------------------------------------------------------------------------------------------------
_TICKETS = {}
_TICKET_POOL = 1 << 15
_COOP_FLAG = {}
_BARRIERS = {}


class CoopGruTimeout(RuntimeError):
    """A cooperative GRU launch gave up waiting for a peer workgroup: its outputs (and everything computed from them
    since) are invalid."""


def coop_error_flag(device=None) -> Optional[Tensor]:
    """The device's sticky time-out word as a (1,) float32 VIEW of the int32 (non-zero bits <=> timed out), ready to be
    concatenated into the trainer's per-step read-back; None before init_tickets()."""
    key = torch.cuda.current_device() if device is None or torch.device(device).index is None \
        else torch.device(device).index
    f = _COOP_FLAG.get(key)
    return None if f is None else f.view(torch.float32)


def check_coop_flag(host_value) -> None:
    """Raise if a read-back of coop_error_flag() is non-zero (any bit pattern but +0.0)."""
    import struct
    bits = struct.unpack('<I', struct.pack('<f', float(host_value)))[0]
    if bits & 8:
        raise RuntimeError('deterministic mode: a workgroup of an accumulating launch never got its turn (another launch '
                           'ran beside it on a side stream, or the dispatcher did not start workgroups in index order): '
                           'the sums since the last check are not ordered')
    if bits & 4:
        raise CoopGruTimeout('a one-launch BatchNorm kernel timed out waiting for its folding workgroup (its workgroups '
                             'were not co-resident): results since the last check are invalid. Set S2AG_BN_FUSED=0')
    if bits & 2:
        raise RuntimeError('touched-row exchange: a batch held more distinct word ids than the row capacity the trainer '
                           'was built with (args.max_words_per_clip too small): the embedding gradient was truncated')
    if bits:
        raise CoopGruTimeout('a cooperative GRU recurrence timed out waiting for a peer workgroup (the 10 workgroups '
                             'of a group were not co-resident): results since the last check are invalid. Re-run with '
                             'fewer concurrent passes (Processor(..., overlap_passes=False)) or set ops.USE_COOP_GRU = False')


def init_tickets(device) -> None:
    """Allocate the per-device pool of ticket words (zero once; every kernel leaves its word at zero again).  Called
    outside stream capture (Processor.__init__); a lazy first use inside a capture would put the pool in the graph's
    private memory."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _TICKETS:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('ops.init_tickets(device) must run before hipGraph capture (run one eager step first)')
        _TICKETS[key] = [torch.zeros(_TICKET_POOL, dtype=torch.int32, device=dev), 0]
        # sticky time-out flag of the cooperative GRU launches (one word; the trainer reads it with its per-step
        # read-back and raises: a recurrence that lost a peer continues with wrong values)
        _COOP_FLAG[key] = torch.zeros(1, dtype=torch.int32, device=dev)
        L.check(_lib().s2ag_gru_coop_set_error_flag(_p(_COOP_FLAG[key])), 'gru_coop_set_error_flag')
        L.check(_lib().s2ag_bn_set_error_flag(_p(_COOP_FLAG[key])), 'bn_set_error_flag')     # bit 4: one-launch BatchNorm
        # ... and the fused Adam refuses to touch the weights while the word is non-zero (no host round trip)
        L.check(_lib().s2ag_adam_set_guard(_p(_COOP_FLAG[key])), 'adam_set_guard')
        _BARRIERS[key] = [torch.zeros(3 * _TICKET_POOL, dtype=torch.int32, device=dev), 0]


def _barrier(dev):
    """Next triple of barrier words of the one-launch BatchNorm kernels (round robin like _ticket)."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _BARRIERS:
        init_tickets(dev)
    ent = _BARRIERS[key]
    i = ent[1]
    ent[1] = (i + 1) % _TICKET_POOL
    return C.c_void_p(ent[0].data_ptr() + 12 * i)


def _ticket(dev):
    """Next ticket word, round robin: launches that can be in flight together (a few streams x a few kernels) are
    always far fewer than the pool, and a hipGraph replays each launch with the word it captured."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _TICKETS:
        init_tickets(dev)
    ent = _TICKETS[key]
    i = ent[1]
    ent[1] = (i + 1) % _TICKET_POOL
    return C.c_void_p(ent[0].data_ptr() + 4 * i)


def _tickets(dev, n: int):
    """``n`` consecutive ticket words (two-level folds: one per group of 16 partial rows + the global one)."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _TICKETS:
        init_tickets(dev)
    ent = _TICKETS[key]
    i = ent[1] if ent[1] + n <= _TICKET_POOL else 0
    ent[1] = (i + n) % _TICKET_POOL
    return C.c_void_p(ent[0].data_ptr() + 4 * i)


# ----------------------------------------------------------------------------------------------------
# direct gradient accumulation
# ----------------------------------------------------------------------------------------------------
def _grad_slot(p: Optional[Tensor]):
    """If ``p`` is a leaf parameter whose ``.grad`` already exists (the flat arena of optim.ParamArena, zeroed once per
    step), return that tensor: the backward kernel then ACCUMULATES into it and autograd is handed ``None`` --
    no temporary, no zero-fill launch, no AccumulateGrad add.  Otherwise None (ordinary autograd path)."""
    if p is None or not p.is_leaf or not p.requires_grad or p.grad is None:
        return None
    g = p.grad
    if not g.is_contiguous() or g.dtype != torch.float32:
        return None
    return g


# ----------------------------------------------------------------------------------------------------
# conv / linear
# ----------------------------------------------------------------------------------------------------
class _ConvNLC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, geom, act, slope, drop_p, noise, site, w_tm=None):
        _need_cuda(x, w, bias)
        N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, wtm = geom
        x, _, _, _ = as_rows(x)
        ctx.w_leaf, ctx.b_leaf = w, bias                   # for direct accumulation into .grad (see _grad_slot)
        ctx.w_shape = w.shape
        # w_tm: tap-major copy of a reference-layout weight (see tap_major): the forward / data-gradient kernels read
        # the copy, the weight gradient is written in the leaf's own layout -- no staging, nothing to flush
        ctx.wtm_k = 1 if w_tm is not None else wtm
        w = w_tm if w_tm is not None else w.contiguous()
        y = torch.empty(N * Lout, Cout, dtype=torch.float32, device=x.device)
        st = conv_fwd_raw(x, w, bias, y, N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, act, slope, drop_p, noise, site,
                          ctx.wtm_k, stats_out=True if _WANT_STATS[0] else None)
        _LAST_STATS[0] = st
        ctx.geom, ctx.epi = geom, (act, slope, drop_p, site)
        ctx.noise = noise
        ctx.has_bias = bias is not None
        needs_y = (act == L.ACT_LEAKY and slope != 1.0) or act == L.ACT_SIGMOID
        ctx.save_for_backward(x, w, y if needs_y else None)
        return y.view(N, Lout, Cout)

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, wtm = ctx.geom
        act, slope, drop_p, site = ctx.epi
        dy = dy.reshape(N * Lout, Cout)
        if (act != L.ACT_NONE and not (act == L.ACT_LEAKY and slope == 1.0)) or drop_p > 0:
            g = torch.empty(N * Lout, Cout, dtype=torch.float32, device=dy.device)
            epilogue_bwd_raw(dy, y, g, act, slope, drop_p, ctx.noise, site)
        else:
            g, _, _, _ = as_rows(dy)
        dx = dw = db = None
        wslot = _grad_slot(ctx.w_leaf) if ctx.needs_input_grad[1] else None
        bslot = _grad_slot(ctx.b_leaf) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        flops = 2.0 * N * Lout * Cout * Cin * ks
        if (FUSE_BWD_PAIR and ctx.needs_input_grad[0] and wslot is not None and ctx.wtm_k == wtm
                and flops < ASYNC_WGRAD_MIN_FLOPS and (bslot is not None or not (ctx.has_bias and ctx.needs_input_grad[2]))):
            # both backward GEMMs of the layer in one launch (inline weight gradients only: the big GRU ones run
            # beside the next layer's recurrence on a forked stream instead)
            dx = torch.empty(N * Lin, Cin, dtype=torch.float32, device=dy.device)
            if conv_bwd_pair_raw(g, w, x, dx, wslot, bslot, N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, wtm):
                _note_staged(ctx.w_leaf)
                if bslot is not None:
                    _note_staged(ctx.b_leaf)
                return dx.view(x.shape), None, None, None, None, None, None, None, None, None
            dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(N * Lin, Cin, dtype=torch.float32, device=dy.device)
            conv_bwd_data_raw(g, w, dx, N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, False, ctx.wtm_k)
            dx = dx.view(x.shape)
        need_db = ctx.has_bias and ctx.needs_input_grad[2] and bslot is None
        if need_db:
            db = torch.empty(Cout, dtype=torch.float32, device=dy.device)
        if ctx.needs_input_grad[1] and wslot is None:
            dw = torch.empty(ctx.w_shape, dtype=torch.float32, device=dy.device)
            conv_bwd_weight_raw(g, x, dw, N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, False, wtm,
                                dbias=db if need_db else None)
        elif need_db:
            colsum_raw(g, db)
        if wslot is not None or bslot is not None:
            lib = _lib()
            xr, _, _, ldx_ = as_rows(x)
            # strided convs whose window is contiguous in memory (no padding, dilation 1, reference-layout weight: the wave
            # encoder's conv2-4): weight gradient through the LDS transpose read with the window as ONE tap of ks*Cin channels
            flat_tr = (CONV_WGRAD_TR and wslot is not None and not wtm and pad == 0 and dil == 1 and ks > 1 and ldx_ == Cin
                       and Cin % 4 == 0 and Cout % 4 == 0 and Lout >= 32 and flops >= 4e8
                       and lib.s2ag_gru_coop_split_pieces() in (1, 2))

            def leaves():
                if flat_tr:
                    gg, _, _, ldg_ = as_rows(g)
                    job = (L.BF16Wgrad * 1)()
                    job[0] = L.BF16Wgrad(_p(gg), _p(xr), _p(wslot), _p(bslot), N, Lout, Lin, Lin * ldx_, ldx_, ldg_, stride, 0, 0,
                                         1, (ks * Cin + 3) // 4 * 4, ks * Cin, Cout, Cin, Cin * ks, 1, ks, Cin, ks)
                    sc = _gru_wgrad_scratch(g.device, (id(ctx.w_leaf), N, Lout),
                                            int(lib.s2ag_f32_wgrad_tr_scratch_floats(job, 1)))
                    L.check(lib.s2ag_f32_wgrad_tr(job, 1, _p(sc), sc.numel(), _stream()), 'f32_wgrad_tr')
                elif wslot is not None:      # the bias gradient rides along in the same launch
                    conv_bwd_weight_raw(g, x, wslot, N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, True, wtm,
                                        dbias=bslot)
                elif bslot is not None:
                    colsum_raw(g, bslot, accumulate=True)
            run_wgrad(leaves, keep=(g, x), flops=2.0 * N * Lout * Cout * Cin * ks)
            if wslot is not None:
                _note_staged(ctx.w_leaf)        # derived tensors (folded / weight-normed): flush when backward ends
            if bslot is not None:
                _note_staged(ctx.b_leaf)
        return dx, dw, db, None, None, None, None, None, None, None


# big stride-1 tap-major convs without a BatchNorm behind them (the TCN): 21.6 us against 27.5 us for the f32 straight-line
# kernel at M = 4 352, 300 -> 300, 2 taps; +2.1 % on the step
SPLIT_CONV = True
SPLIT_LINEAR = True   # 1-tap layers >= 1 GFLOP (GRU layer-0 projections): 22 vs 33 us
SPLIT_CONV_MIN_FLOPS = 1e9
SPLIT_GEMM = True
SPLIT_GEMM_DX = True
SPLIT_GEMM_MIN_FLOPS = 4e9


def split_planes_raw(x: Tensor) -> Tensor:
    """(rows, K) fp32 -> (3, rows, Kp) bf16 planes: the exact 3-piece split consumed by gemm_split_raw."""
    x, rows, K, ldx = as_rows(x)
    Kp = _lib().s2ag_split_k_padded(K)
    planes = torch.empty(3, rows, Kp, dtype=torch.bfloat16, device=x.device)
    L.check(_lib().s2ag_split_bf16x3(_p(x), rows, K, ldx, _p(planes), _stream()), 'split_bf16x3')
    return planes


def split_planes_cached(w: Tensor) -> Tensor:
    """Planes of a weight matrix (N, K), refreshed when the weight changes (same keying as tap_major)."""
    key = _source_key(w) + ((_GENERATION[0],) if w.requires_grad else ())
    ent = getattr(w, '_s2ag_sp', None)
    if ent is None or ent[0] != key:
        with torch.no_grad():
            ent = (key, split_planes_raw(w.detach()))
        w._s2ag_sp = ent
    return ent[1]


def gemm_split_raw(a_planes: Tensor, w_planes: Tensor, bias: Optional[Tensor], y: Tensor, K: int):
    _, M, Kp = a_planes.shape
    N = w_planes.shape[1]
    assert w_planes.shape[2] == Kp and y.shape == (M, N) and y.is_contiguous()
    L.check(_lib().s2ag_gemm_split_fwd(_p(a_planes), _p(w_planes), _p(bias), _p(y), M, N, K, N, _stream()), 'gemm_split')


# off by default: one GRU layer's three weight gradients take 197 us this way (6 transposed splits + 3 split-K GEMMs)
# against 206 us for the single f32-MFMA wgrad_multi_k launch -- nine launches for nothing (tools/bench history)
SPLIT_WGRAD = False


def split_planes_t_raw(x: Tensor, shift: int = 0, L_: int = 1, colsum: Optional[Tensor] = None) -> Tensor:
    """(rows, cols) fp32 -> (3, cols, Rp) bf16: the transposed split (contraction axis = rows), optionally frame-shifted
    inside clips of ``L_`` frames; ``colsum`` += column sums of x (bias gradient)."""
    x, rows, cols, ldx = as_rows(x)
    Rp = _lib().s2ag_split_k_padded(rows)
    planes = torch.empty(3, cols, Rp, dtype=torch.bfloat16, device=x.device)
    L.check(_lib().s2ag_split_bf16x3_t(_p(x), rows, cols, ldx, int(shift), int(L_), _p(planes), _p(colsum), _stream()),
            'split_bf16x3_t')
    return planes


def gemm_split_acc_raw(a_planes: Tensor, w_planes: Tensor, y: Tensor, K: int):
    """y (M, N) += a w^T from transposed planes a (3, M, Kp), w (3, N, Kp)."""
    _, M, Kp = a_planes.shape
    N = w_planes.shape[1]
    assert w_planes.shape[2] == Kp and y.is_contiguous() and y.numel() == M * N
    L.check(_lib().s2ag_gemm_split_acc(_p(a_planes), _p(w_planes), _p(y), M, N, K, N, _stream()), 'gemm_split_acc')


def tap_major(w: Tensor) -> Tensor:
    """Cached (Cout, k, Cin) copy of a reference-layout (Cout, Cin, k) conv weight, refreshed when the weight changes
    (optimizer step, load_state_dict) and, for trainable weights, at every step boundary (see begin_step).  Frozen
    weights keep one copy for good."""
    key = _source_key(w) + ((_GENERATION[0],) if w.requires_grad else ())
    ent = getattr(w, '_s2ag_tm', None)
    if ent is None or ent[0] != key:
        with torch.no_grad():
            ent = (key, w.detach().permute(0, 2, 1).contiguous())
        w._s2ag_tm = ent
    return ent[1]


TM_COPIES = True
BN_STATS_EPILOGUE = True
_WANT_STATS = [False]
_LAST_STATS = [None]


def conv1d_nlc(x: Tensor, w: Tensor, bias: Optional[Tensor], stride=1, pad=0, dil=1, lout: Optional[int] = None,
               act=L.ACT_NONE, slope=1.0, drop_p=0.0, noise=None, site=0, w_tap_major=False, bn_stats=False,
               tm_copy=False) -> Tensor:
    """x (N, Lin, Cin) channels-last, w (Cout, Cin, k) [or (Cout, k, Cin) if ``w_tap_major``] -> (N, Lout, Cout).
    ``lout`` overrides the usual output length (the TCN's causal conv + chomp is pad = (k-1)*dil on the left with
    lout = Lin).  ``bn_stats``: the output goes straight into a training-mode ``batch_norm_act``; where the kernel has a
    statistics epilogue the column sums ride along (attribute on the returned tensor) and BatchNorm skips its own pass.
    ``tm_copy``: run the forward / data gradient from a cached tap-major copy of ``w`` (straight-line kernels)."""
    N, Lin, Cin = x.shape
    if w_tap_major:
        Cout, ks, Cin_w = w.shape
    else:
        Cout, Cin_w, ks = w.shape
    assert Cin_w == Cin, (w.shape, x.shape)
    if lout is None:
        lout = (Lin + 2 * pad - dil * (ks - 1) - 1) // stride + 1
    _WANT_STATS[0] = bool(bn_stats) and BN_STATS_EPILOGUE
    try:
        w_tm = tap_major(w) if (tm_copy and TM_COPIES and not w_tap_major and ks > 1 and Cin % 4 == 0) else None
        out = _ConvNLC.apply(x, w, bias, (N, Lin, lout, Cin, Cout, ks, stride, pad, dil, int(w_tap_major)), act,
                             float(slope), float(drop_p), noise, site, w_tm)
    finally:
        _WANT_STATS[0] = False
    out = out.view(N, lout, Cout)
    if _LAST_STATS[0] is not None:
        out._s2ag_stats = _LAST_STATS[0]        # (partials, rows): consumed by batch_norm_act on this very tensor
    _LAST_STATS[0] = None
    return out


def linear(x: Tensor, w: Tensor, bias: Optional[Tensor], act=L.ACT_NONE, slope=1.0) -> Tensor:
    """x (..., in) @ w(out, in)^T + bias -> (..., out) with a fused activation."""
    shp = x.shape
    rows = x.numel() // shp[-1]
    # w is passed as the leaf itself (not a view) so its gradient can be accumulated straight into the arena
    y = _ConvNLC.apply(x, w, bias, (rows, 1, 1, shp[-1], w.shape[0], 1, 1, 0, 1, 0), act, float(slope), 0.0, None, 0)
    return y.view(*shp[:-1], w.shape[0])


# ----------------------------------------------------------------------------------------------------
# batch norm (+ leaky activation)
# ----------------------------------------------------------------------------------------------------
_BN_REPEAT = [1]
_BN_PRE = [None]


class bn_repeat:
    """Context: training-mode BatchNorms inside advance their running estimates ``k`` times.  Used where ``k`` forward
    passes of one step would see the very same batch (same input, same weights): the sub-network runs once and the
    running statistics end up exactly where ``k`` separate passes leave them."""

    def __init__(self, k: int):
        self.k = int(k)

    def __enter__(self):
        self.prev = _BN_REPEAT[0]
        _BN_REPEAT[0] = self.k

    def __exit__(self, *a):
        _BN_REPEAT[0] = self.prev


def generation() -> int:
    return _GENERATION[0]


# One-launch BatchNorm (statistics + apply with a grid-wide wait, norm_elementwise.hip): -69 launches per step, kernel time
# = the sum of the two kernels it replaces.  Neutral when it was built (profiles/r02_bn_one_launch.txt: 14 620 / 14 820 vs
# 14 690 / 14 740 clips/s); since the chains of a replayed graph start ~2.5 us later per node captured ahead of them
# (lessons) fewer nodes now pay: +0.95 % (17 890 vs 17 720 clips/s, same-box A/B x 4), so it is the
# default.  <= 64 workgroups per launch, bounded poll, sticky error bit read with the step's losses (check_coop_flag);
# S2AG_BN_FUSED=0 keeps the two launches, whose workgroups never wait for each other.
BN_FUSED = config.mirror('BN_FUSED', globals(), 'BN_FUSED')
# opt-in variant (csrc/bn_foldapply.hip): where the producing conv left its column sums, fold + apply in one launch (21 pairs a step)
BN_FOLD_APPLY = config.mirror('BN_FOLD_APPLY', globals(), 'BN_FOLD_APPLY')


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, nbt, chan_map, slope, training, eps, momentum):
        _need_cuda(x, gamma)
        lib = _lib()
        ctx.in_shape = x.shape
        x, rows, cols, ldx = as_rows(x)
        nchan = gamma.numel()
        dev = x.device
        coef = torch.empty(4, cols, dtype=torch.float32, device=dev)
        pre = _BN_PRE[0]
        _BN_PRE[0] = None
        if training and pre is not None and BN_FOLD_APPLY and lib.s2ag_bn_fold_apply_supported(int(pre[1]), cols, nchan):
            part, prow = pre        # column sums left behind by the producing layer: fold + apply, one launch
            y = torch.empty(rows, cols, dtype=torch.float32, device=dev)
            L.check(lib.s2ag_bn_fold_apply(_p(part), int(prow), rows, cols, _p(chan_map), nchan, _p(gamma), _p(beta), _p(rmean),
                                           _p(rvar), _p(nbt), float(eps), float(momentum), int(_BN_REPEAT[0]), _p(coef[0]),
                                           _p(coef[1]), _p(coef[2]), _p(coef[3]), _p(x), ldx, float(slope), _p(y), cols,
                                           _stream()), 'bn_fold_apply')
            ctx.save_for_backward(x, coef, chan_map)
            ctx.meta = (rows, cols, ldx, nchan, float(slope), bool(training))
            ctx.leaves = (gamma, beta)
            return y
        elif training and pre is not None:
            part, prow = pre        # column sums left behind by the producing layer: only the fold remains
            L.check(lib.s2ag_bn_fold(_p(part), int(prow), rows, cols, _p(chan_map), nchan, _p(gamma), _p(beta), _p(rmean),
                                     _p(rvar), _p(nbt), float(eps), float(momentum), int(_BN_REPEAT[0]), _p(coef[0]),
                                     _p(coef[1]), _p(coef[2]), _p(coef[3]), _stream()), 'bn_fold')
        elif training and BN_FUSED and lib.s2ag_bn_fused_supported(rows, cols):
            # ONE launch for statistics, fold, coefficients and the apply (grid-wide wait inside the kernel)
            nrb = lib.s2ag_bn_fused_partial_rows(rows, cols, max(ldx, cols))
            part = torch.empty(2 * nrb * cols, dtype=torch.float64, device=dev)
            y = torch.empty(rows, cols, dtype=torch.float32, device=dev)
            L.check(lib.s2ag_bn_fwd_fused(_p(x), rows, cols, ldx, _p(chan_map), nchan, _p(gamma), _p(beta), _p(rmean),
                                          _p(rvar), _p(nbt), float(eps), float(momentum), int(_BN_REPEAT[0]), _p(part),
                                          _barrier(dev), _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]), float(slope),
                                          _p(y), cols, _stream()), 'bn_fwd_fused')
            ctx.save_for_backward(x, coef, chan_map)
            ctx.meta = (rows, cols, ldx, nchan, float(slope), bool(training))
            ctx.leaves = (gamma, beta)
            return y
        elif training:
            # one launch: fp64 partial column sums per row block, folded into coefficients by the last block
            nrb = lib.s2ag_bn_partial_rows(rows, cols, ldx)
            part = torch.empty(2 * nrb * cols, dtype=torch.float64, device=dev)
            L.check(lib.s2ag_bn_fwd_stats(_p(x), rows, cols, ldx, _p(chan_map), nchan, _p(gamma), _p(beta), _p(rmean),
                                          _p(rvar), _p(nbt), float(eps), float(momentum), int(_BN_REPEAT[0]), _p(part),
                                          _ticket(dev),
                                          _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]), _stream()),
                    'bn_fwd_stats')
        else:
            L.check(lib.s2ag_bn_coeffs(None, None, _p(chan_map), cols, nchan, rows, _p(gamma), _p(beta), _p(rmean),
                                       _p(rvar), None, float(eps), float(momentum), 0, _p(coef[0]),
                                       _p(coef[1]), _p(coef[2]), _p(coef[3]), _stream()), 'bn_coeffs')
        y = torch.empty(rows, cols, dtype=torch.float32, device=dev)
        L.check(lib.s2ag_bn_apply(_p(x), rows, cols, ldx, _p(coef[0]), _p(coef[1]), float(slope), _p(y), cols,
                                  _stream()), 'bn_apply')
        ctx.save_for_backward(x, coef, chan_map)
        ctx.meta = (rows, cols, ldx, nchan, float(slope), bool(training))
        ctx.leaves = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, coef, chan_map = ctx.saved_tensors
        rows, cols, ldx, nchan, slope, training = ctx.meta
        lib = _lib()
        dy, _, _, lddy = as_rows(dy.reshape(rows, cols))
        dev = dy.device
        dx = torch.empty(rows, cols, dtype=torch.float32, device=dev)
        dgamma = dbeta = None
        if training:
            tmp = torch.empty(2, cols, dtype=torch.float32, device=dev)
            nrb = lib.s2ag_bn_partial_rows(rows, cols, max(ldx, lddy))
            part = torch.empty(2 * nrb * cols, dtype=torch.float32, device=dev)
            sg, sb = _grad_slot(ctx.leaves[0]), _grad_slot(ctx.leaves[1])
            direct = sg is not None and sb is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
            if not direct:
                dgb = torch.empty(2, nchan, dtype=torch.float32, device=dev)
                sg, sb = dgb[0], dgb[1]
                dgamma, dbeta = dgb[0], dgb[1]
            if BN_FUSED and lib.s2ag_bn_fused_supported(rows, cols):
                L.check(lib.s2ag_bn_bwd_fused(_p(x), _p(dy), rows, cols, ldx, lddy, _p(coef[0]), _p(coef[1]), _p(coef[2]),
                                              _p(coef[3]), slope, _p(chan_map), nchan, _p(sg), _p(sb), int(direct),
                                              _p(part), _barrier(dev), _p(tmp[0]), _p(tmp[1]), _p(dx), cols, _stream()),
                        'bn_bwd_fused')
                return dx.view(ctx.in_shape), dgamma, dbeta, None, None, None, None, None, None, None, None
            L.check(lib.s2ag_bn_bwd_stats(_p(x), _p(dy), rows, cols, ldx, lddy, _p(coef[0]), _p(coef[1]), _p(coef[2]),
                                          _p(coef[3]), slope, _p(chan_map), nchan, _p(sg), _p(sb), int(direct),
                                          _p(part), _ticket(dev), _p(tmp[0]), _p(tmp[1]), _stream()), 'bn_bwd_stats')
            c1, c2 = tmp[0], tmp[1]
        else:
            z = torch.zeros(2, cols, dtype=torch.float32, device=dev)
            c1, c2 = z[0], z[1]
        L.check(lib.s2ag_bn_bwd_apply(_p(x), _p(dy), rows, cols, ldx, lddy, _p(coef[0]), _p(coef[1]), _p(coef[2]),
                                      _p(coef[3]), slope, _p(c1), _p(c2), _p(dx), cols, _stream()), 'bn_bwd_apply')
        return dx.view(ctx.in_shape), dgamma, dbeta, None, None, None, None, None, None, None, None


def batch_norm_act(x: Tensor, bn: torch.nn.Module, slope: float = 1.0, chan_map: Optional[Tensor] = None,
                   training: Optional[bool] = None) -> Tensor:
    """BatchNorm over the last (column) axis of a channels-last tensor + leaky(slope).  ``bn`` is a
    torch BatchNorm module used purely as the parameter/buffer container (state_dict compatibility)."""
    tr = bn.training if training is None else training
    shp = x.shape
    _BN_PRE[0] = getattr(x, '_s2ag_stats', None) if tr else None
    y = _BNAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                     bn.num_batches_tracked if tr else None, chan_map, float(slope), bool(tr), float(bn.eps),
                     float(bn.momentum))
    return y.view(shp)


# ----------------------------------------------------------------------------------------------------
# residual add + activation
# ----------------------------------------------------------------------------------------------------
class _AddAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, slope):
        _need_cuda(a, b)
        a_, r, c, _ = as_rows(a)
        y = torch.empty(r, c, dtype=torch.float32, device=a.device)
        add_act_raw(a_, b, y, slope)
        ctx.slope = slope
        ctx.has_b = b is not None
        ctx.save_for_backward(y if slope != 1.0 else None)
        return y.view(a.shape)

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        if ctx.slope == 1.0:
            g = dy
        else:
            g = torch.empty(y.shape, dtype=torch.float32, device=dy.device)
            epilogue_bwd_raw(dy.reshape(y.shape), y, g, L.ACT_LEAKY, ctx.slope, 0.0, None, 0)
            g = g.view(dy.shape)
        return g, (g if ctx.has_b else None), None


def add_act(a: Tensor, b: Optional[Tensor], slope: float) -> Tensor:
    return _AddAct.apply(a, b, float(slope))


# ----------------------------------------------------------------------------------------------------
# clip-resident TemporalConvNet forward of the fp32 step (csrc/tcn_fused32.hip)
# ----------------------------------------------------------------------------------------------------
TCN_FUSED32 = config.mirror('TCN_FUSED32', globals(), 'TCN_FUSED32')
TCN_FUSED32_BWD = True
# the TCN's eight weight gradients right behind its data-gradient chain on the SAME stream: forked, they queue behind the last
# GRU layer's weight gradients on the weight-gradient stream and the step ends waiting for them (15 190 forked vs 15 700
# inline clips/s; layer-by-layer backward 14 980-15 290)
TCN32_WGRAD_INLINE = True
TCN32_WGRAD_BLOCKS = 256


def tcn_fused32_supported(T: int, Cch: int, ks: int, n_blocks: int) -> bool:
    lib = _lib()
    return (TCN_FUSED32 and SPLIT_CONV and 1 <= n_blocks <= 4 and lib.s2ag_gru_coop_split_pieces() == 2
            and bool(lib.s2ag_tcn32_supported(int(T), int(Cch), int(ks))))


class TcnFragments32:
    """hi / lo bf16 planes of the normalised conv weights in MFMA-fragment order, refreshed once per optimizer step."""

    def __init__(self):
        self._key, self._frag = None, None

    def get(self, ws):
        key = (_GENERATION[0],) + tuple((id(w), w._version, w.data_ptr()) for w in ws)
        if key != self._key:
            lib = _lib()
            frag = torch.empty(int(lib.s2ag_tcn32_pack_elems(len(ws))), dtype=torch.bfloat16, device=ws[0].device)
            ptrs = (C.c_void_p * len(ws))(*[w.detach().data_ptr() for w in ws])
            L.check(lib.s2ag_tcn32_pack(ptrs, len(ws), ws[0].shape[0], _p(frag), _stream()), 'tcn32_pack')
            self._key, self._frag = key, frag
        return self._frag


class _FakeCtx:
    """What _ConvNLC.backward reads from its context, for a conv whose forward ran inside the fused kernel."""
    pass


class _TcnFused32(torch.autograd.Function):
    """x (B, T, C) fp32 -> the last TemporalBlock's output; ``params`` = the 2*nb normalised (C, 2, C) tap-major weights
    followed by the 2*nb biases.  Forward: one launch (+ the keep bits).  Backward: the layer-by-layer kernels on the
    tensors the forward left behind (h1, h2, y per block) -- exactly what the per-layer path would have saved."""

    @staticmethod
    def forward(ctx, x, frags, meta, noise, *params):
        dils, sites, drop_p = meta[:3]
        big, noises = (meta[3], meta[4]) if len(meta) > 3 else (None, None)
        nb = len(dils)
        ws, bs = params[:2 * nb], params[2 * nb:]
        lib = _lib()
        B, T, Cch = x.shape
        x2, rows, cols, ldx = as_rows(x)
        if ldx != cols:
            x2 = x2.contiguous()
        nP = 1
        if big is not None:
            # lockstep batch: x is the first B clips of ``big`` (nP * B clips, one pass after the other; pass k draws its keep
            # bits from noises[k]); only x's pass keeps what the backward pass needs
            nP = len(noises)
            assert big.is_contiguous() and big.shape == (nP * B, T, Cch) and x2.data_ptr() == big.data_ptr() and noises[0] is noise
        frag = frags.get(ws)
        saved = torch.empty(3 * nb - 1, rows, Cch, dtype=torch.float32, device=x.device)     # h1, h2, y per block ...
        y_last = torch.empty(nP * rows, Cch, dtype=torch.float32, device=x.device)           # ... the last y of every pass
        a = L.Tcn32()
        a.x, a.wfrag = x2.data_ptr(), frag.data_ptr()
        for b in range(nb):
            a.h1[b], a.h2[b] = saved[3 * b].data_ptr(), saved[3 * b + 1].data_ptr()
            a.y[b] = saved[3 * b + 2].data_ptr() if b < nb - 1 else y_last.data_ptr()
            a.dil[b] = int(dils[b])
            for j in range(2):
                a.bias[2 * b + j] = bs[2 * b + j].data_ptr() if bs[2 * b + j] is not None else None
                a.site[2 * b + j] = int(sites[2 * b + j])
        a.n_blocks, a.n_clips, a.T, a.C = nb, nP * B, T, Cch
        a.drop_p = float(drop_p)
        if drop_p > 0:
            keep = torch.empty(int(lib.s2ag_tcn32_keep_bytes(nP * B, nb)), dtype=torch.uint8, device=x.device)
            a.rng, a.keep = noise.data_ptr(), keep.data_ptr()
        if nP == 1:
            L.check(lib.s2ag_tcn32_fwd(C.byref(a), _stream()), 'tcn32_fwd')
        else:
            rngs = (C.c_void_p * nP)(*[nz.data_ptr() for nz in noises])
            L.check(lib.s2ag_tcn32_fwd_passes(C.byref(a), nP, rngs, B, _stream()), 'tcn32_fwd_passes')
        ctx.meta, ctx.noise, ctx.params, ctx.shape, ctx.frags = (dils, sites, drop_p), noise, params, (B, T, Cch), frags
        ctx.save_for_backward(x2, saved, y_last)
        out = y_last[:rows].view(B, T, Cch)
        if nP == 1:
            return out
        mates = y_last[rows:].view((nP - 1) * B, T, Cch)
        ctx.mark_non_differentiable(mates)
        return out, mates

    @staticmethod
    def backward(ctx, gy, *_mates):
        x2, saved_t, y_last = ctx.saved_tensors
        dils, sites, drop_p = ctx.meta
        nb = len(dils)
        # h1, h2, y per block as before; the last block's y lives at the head of the passes' output buffer
        saved = [saved_t[i] for i in range(3 * nb - 1)] + [y_last[:x2.numel() // x2.shape[-1]]]
        params = ctx.params
        ws, bs = params[:2 * nb], params[2 * nb:]
        B, T, Cch = ctx.shape
        rows = B * T
        grads = [None] * (4 * nb)
        g = gy.reshape(rows, Cch)
        need_w = [ctx.needs_input_grad[4 + k] for k in range(2 * nb)]
        need_b = [bs[k] is not None and ctx.needs_input_grad[4 + 2 * nb + k] for k in range(2 * nb)]
        wsl = [_grad_slot(ws[k]) if need_w[k] else None for k in range(2 * nb)]
        bsl = [_grad_slot(bs[k]) if need_b[k] else None for k in range(2 * nb)]
        if (TCN_FUSED32_BWD and T >= 32 and all(need_w) and all(sl is not None for sl in wsl)
                and all((not nb_) or sl is not None for nb_, sl in zip(need_b, bsl))):
            # the whole chain of data gradients in ONE launch (csrc/tcn_fused32.hip), the eight weight (+ bias) gradients
            # in one transpose-read launch on the weight-gradient stream
            lib = _lib()
            g2, _, _, ldg = as_rows(g)
            if ldg != Cch:
                g2 = g2.contiguous()
            gx = torch.empty(rows, Cch, dtype=torch.float32, device=g2.device)
            gp = torch.empty(2 * nb, rows, Cch, dtype=torch.float32, device=g2.device)
            a = L.Tcn32()
            a.wfrag = ctx.frags.get(ws).data_ptr()
            for b in range(nb):
                a.h1[b], a.h2[b], a.y[b] = saved[3 * b].data_ptr(), saved[3 * b + 1].data_ptr(), saved[3 * b + 2].data_ptr()
                a.gp1[b], a.gp2[b] = gp[2 * b].data_ptr(), gp[2 * b + 1].data_ptr()
                a.dil[b] = int(dils[b])
            a.n_blocks, a.n_clips, a.T, a.C = nb, B, T, Cch
            a.drop_p = float(drop_p)
            a.gy, a.gx = g2.data_ptr(), gx.data_ptr()
            L.check(lib.s2ag_tcn32_bwd(C.byref(a), _stream()), 'tcn32_bwd')
            jobs = (L.BF16Wgrad * (2 * nb))()
            for b in range(nb):
                d = int(dils[b])
                for j in range(2):
                    k = 2 * b + j
                    xin = (x2.view(rows, Cch) if b == 0 else saved[3 * (b - 1) + 2]) if j == 0 else saved[3 * b]
                    jobs[k] = L.BF16Wgrad(_p(gp[k]), _p(xin), _p(wsl[k]), _p(bsl[k]), B, T, T, T * Cch, Cch, Cch, 1, -d, d, 2,
                                          Cch, Cch, Cch, Cch, 2 * Cch, Cch, 1, 0, 2)
            nblk = TCN32_WGRAD_BLOCKS           # no recurrence runs beside this launch: the whole chip
            sc = _gru_wgrad_scratch(g2.device, (id(ctx.frags), 'tcn32', B, T),
                                    int(lib.s2ag_f32_wgrad_tr_scratch_floats_n(jobs, 2 * nb, nblk)))

            def launch(jobs=jobs, sc=sc):
                L.check(lib.s2ag_f32_wgrad_tr_n(jobs, 2 * nb, _p(sc), sc.numel(), nblk, _stream()), 'f32_wgrad_tr')
            if TCN32_WGRAD_INLINE:
                launch()
            else:
                run_wgrad(launch, keep=(gp, saved_t, y_last, x2), flops=2.0 * rows * Cch * Cch * 2 * 2 * nb)
            for k in range(2 * nb):
                _note_staged(ws[k])
                if bs[k] is not None:
                    _note_staged(bs[k])
            return (gx.view(B, T, Cch) if ctx.needs_input_grad[0] else None, None, None, None) + tuple(grads)
        for b in range(nb - 1, -1, -1):
            d = int(dils[b])
            h1, h2, yb = saved[3 * b], saved[3 * b + 1], saved[3 * b + 2]
            xin = x2.view(rows, Cch) if b == 0 else saved[3 * (b - 1) + 2]
            gs = torch.empty(rows, Cch, dtype=torch.float32, device=g.device)
            epilogue_bwd_raw(g, yb, gs, L.ACT_LEAKY, 0.0, 0.0, None, 0)                 # through relu(h2 + x)
            cur = gs
            for j, (xop, yop) in ((1, (h1, h2)), (0, (xin, h1))):
                k = 2 * b + j
                fc = _FakeCtx()
                fc.saved_tensors = (xop, ws[k], yop)
                fc.geom = (B, T, T, Cch, Cch, 2, 1, d, d, 1)
                fc.epi = (L.ACT_LEAKY, 0.0, float(drop_p), int(sites[k]))
                fc.noise, fc.w_leaf, fc.b_leaf, fc.has_bias = ctx.noise, ws[k], bs[k], bs[k] is not None
                fc.wtm_k, fc.w_shape = 1, ws[k].shape
                need_x = True if (j == 1 or b > 0) else ctx.needs_input_grad[0]
                fc.needs_input_grad = (need_x, ctx.needs_input_grad[4 + k],
                                       bs[k] is not None and ctx.needs_input_grad[4 + 2 * nb + k])
                out = _ConvNLC.backward(fc, cur.view(B, T, Cch))
                grads[k], grads[2 * nb + k] = out[1], out[2]
                cur = out[0].reshape(rows, Cch) if out[0] is not None else None
            if cur is None:
                g = None
                break
            gn = torch.empty(rows, Cch, dtype=torch.float32, device=cur.device)
            add_act_raw(cur, gs, gn, 1.0)                                                # + the residual branch
            g = gn
        gx = g.view(B, T, Cch) if (g is not None and ctx.needs_input_grad[0]) else None
        return (gx, None, None, None) + tuple(grads)


def tcn_fused32(x: Tensor, frags: TcnFragments32, ws, biases, dils, sites, drop_p: float, noise, batch=None, noises=None):
    """``batch`` (nP * B, T, C) + ``noises``: x is the first B clips of a lockstep batch of nP passes (see _TcnFused32);
    returns (out of x's pass, outputs of the other passes ((nP - 1) * B, T, C), no autograd)."""
    if batch is None:
        return _TcnFused32.apply(x, frags, (tuple(dils), tuple(sites), float(drop_p)), noise, *ws, *biases)
    return _TcnFused32.apply(x, frags, (tuple(dils), tuple(sites), float(drop_p), batch, tuple(noises)), noise, *ws, *biases)


# ----------------------------------------------------------------------------------------------------
# embedding (+ dropout)
# ----------------------------------------------------------------------------------------------------
class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, table, drop_p, noise, site, out=None):
        _need_cuda(ids, table)
        ids = ids.contiguous().view(-1)
        rows, dim = ids.numel(), table.shape[1]
        if out is None:
            out = torch.empty(rows, dim, dtype=torch.float32, device=table.device)
        else:       # the caller's buffer (a slice of a lockstep batch): rows x dim, contiguous
            assert out.is_contiguous() and out.numel() == rows * dim and out.dtype == torch.float32
            out = out.view(rows, dim)
        e = _epi(L.ACT_NONE, 1.0, drop_p, noise, site)
        L.check(_lib().s2ag_embedding_fwd(_p(ids), _p(table), rows, dim, table.shape[0], _p(out), dim, C.byref(e),
                                          _stream()), 'embedding_fwd')
        ctx.save_for_backward(ids)
        ctx.meta = (table.shape[0], dim, drop_p, site)
        ctx.noise = noise
        ctx.table_leaf = table
        return out

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        n_entries, dim, drop_p, site = ctx.meta
        dy, rows, _, ldg = as_rows(dy)
        e = _epi(L.ACT_NONE, 1.0, drop_p, ctx.noise, site)
        slot = _grad_slot(ctx.table_leaf)
        if slot is not None:        # dense (n_words x dim) gradient: scatter straight into the arena
            L.check(_lib().s2ag_embedding_bwd(_p(ids), _p(dy), ldg, rows, dim, n_entries, _p(slot), 1, C.byref(e),
                                              _stream()), 'embedding_bwd')
            return None, None, None, None, None, None
        dt = torch.empty(n_entries, dim, dtype=torch.float32, device=dy.device)
        L.check(_lib().s2ag_embedding_bwd(_p(ids), _p(dy), ldg, rows, dim, n_entries, _p(dt), 0, C.byref(e),
                                          _stream()), 'embedding_bwd')
        return None, dt, None, None, None, None


def embedding(ids: Tensor, table: Tensor, drop_p: float = 0.0, noise=None, site=0, out: Optional[Tensor] = None) -> Tensor:
    """``out``: write the rows into the caller's contiguous (ids.numel(), dim) buffer instead of a fresh one."""
    res = _Embedding.apply(ids, table, float(drop_p), noise, site, out)
    return res.view(*ids.shape, table.shape[1])


# ----------------------------------------------------------------------------------------------------
# touched-row exchange of the embedding gradient (data parallel; csrc/rows.hip, parallel.GradExchange)
# ----------------------------------------------------------------------------------------------------
_ROW_MARKS = {}


def rows_unique_raw(ids: Tensor, n_entries: int, uids_out: Tensor, flag_overflow: bool = True) -> Tensor:
    """Sorted unique ids of ``ids`` into ``uids_out`` (int32, padded with ``n_entries``); returns the (1,) int32 device
    count of distinct ids (a persistent buffer: copy it before the next call).  More distinct ids than ``uids_out`` holds
    sets bit 1 of the sticky error word (the trainer raises at its next read-back) unless ``flag_overflow`` is False --
    parallel.GradExchange.precheck repairs an overflow with a dense all-reduce and needs no alarm."""
    _need_cuda(ids, uids_out)
    assert ids.dtype == torch.int64 and ids.is_contiguous() and uids_out.dtype == torch.int32
    key = (ids.device.index, int(n_entries))
    if key not in _ROW_MARKS:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('rows_unique_raw: first use must happen before hipGraph capture (one eager step)')
        _ROW_MARKS[key] = (torch.zeros(n_entries, dtype=torch.int32, device=ids.device),
                           torch.zeros(1, dtype=torch.int32, device=ids.device))
    mark, count = _ROW_MARKS[key]
    flag = _COOP_FLAG.get(ids.device.index if ids.device.index is not None else torch.cuda.current_device()) \
        if flag_overflow else None
    L.check(_lib().s2ag_rows_unique(_p(ids), ids.numel(), int(n_entries), uids_out.numel(), _p(mark), _p(uids_out),
                                    _p(count), _p(flag), _stream()), 'rows_unique')
    return count


def rows_pack_raw(dense: Tensor, uids: Tensor, records_out: Tensor) -> None:
    _need_cuda(dense, uids, records_out)
    n_entries, dim = dense.shape
    assert dense.is_contiguous() and records_out.is_contiguous() and records_out.shape == (uids.numel(), dim + 1)
    L.check(_lib().s2ag_rows_pack(_p(dense), _p(uids), uids.numel(), dim, n_entries, _p(records_out), _stream()),
            'rows_pack')


def rows_merge_raw(gathered: Tensor, dense: Tensor) -> None:
    _need_cuda(gathered, dense)
    world, cap, rec = gathered.shape
    n_entries, dim = dense.shape
    assert rec == dim + 1 and gathered.is_contiguous() and dense.is_contiguous()
    L.check(_lib().s2ag_rows_merge(_p(gathered), world, cap, dim, n_entries, _p(dense), _stream()), 'rows_merge')


# ----------------------------------------------------------------------------------------------------
# weight norm
# ----------------------------------------------------------------------------------------------------
class _WeightNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, g, tap_major):
        _need_cuda(v, g)
        rows, cols = v.shape[0], v.numel() // v.shape[0]
        v = v.contiguous()
        ks = v.shape[2] if (tap_major and v.dim() == 3) else 1
        w = torch.empty((v.shape[0], v.shape[2], v.shape[1]) if ks > 1 else v.shape, dtype=torch.float32,
                        device=v.device)
        norm = torch.empty(rows, dtype=torch.float32, device=v.device)
        L.check(_lib().s2ag_weight_norm_fwd(_p(v), _p(g), rows, cols, ks, _p(w), _p(norm), _stream()),
                'weight_norm_fwd')
        ctx.save_for_backward(v, g, norm)
        ctx.ks = ks
        return w

    @staticmethod
    def backward(ctx, dw):
        v, g, norm = ctx.saved_tensors
        rows, cols = v.shape[0], v.numel() // v.shape[0]
        dw = dw.contiguous()
        dv = torch.empty_like(v)
        dg = torch.empty_like(g)
        L.check(_lib().s2ag_weight_norm_bwd(_p(dw), _p(v), _p(g), _p(norm), rows, cols, ctx.ks, _p(dv), _p(dg),
                                            _stream()), 'weight_norm_bwd')
        return dv, dg, None


def weight_norm(v: Tensor, g: Tensor, tap_major: bool = False) -> Tensor:
    """w = g * v / ||v|| per output row.  ``tap_major``: v is (Cout, Cin, k) and w comes out as (Cout, k, Cin), the
    K-contiguous layout the conv tile loader reads with 16-byte loads."""
    return _WeightNorm.apply(v, g, bool(tap_major))


# ----------------------------------------------------------------------------------------------------
# CSR fold (ST-GCN weights)
# ----------------------------------------------------------------------------------------------------
class CSR:
    """Device CSR pair (M and M^T) of a fixed sparse linear map."""

    def __init__(self, mat, device):
        import scipy.sparse as sp
        m = sp.csr_matrix(mat)
        m.sort_indices()
        mt = sp.csr_matrix(m.T)
        mt.sort_indices()
        self.shape = m.shape

        def dev(a, dt):
            return torch.as_tensor(a, dtype=dt).to(device)
        self.fwd = (dev(m.indptr, torch.int32), dev(m.indices, torch.int32), dev(m.data, torch.float32))
        self.bwd = (dev(mt.indptr, torch.int32), dev(mt.indices, torch.int32), dev(mt.data, torch.float32))


class _Fold(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, csr: CSR):
        _need_cuda(w)
        w = w.contiguous()
        rp, ci_, va = csr.fwd
        y = torch.empty(csr.shape[0], dtype=torch.float32, device=w.device)
        L.check(_lib().s2ag_spmv(_p(rp), _p(ci_), _p(va), _p(w), _p(y), csr.shape[0], 0, _stream()), 'spmv')
        ctx.csr = csr
        ctx.wshape = w.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        csr = ctx.csr
        rp, ci_, va = csr.bwd
        dy = dy.contiguous()
        dw = torch.empty(csr.shape[1], dtype=torch.float32, device=dy.device)
        L.check(_lib().s2ag_spmv(_p(rp), _p(ci_), _p(va), _p(dy), _p(dw), csr.shape[1], 0, _stream()), 'spmv^T')
        return dw.view(ctx.wshape), None


def fold(w: Tensor, csr: CSR) -> Tensor:
    return _Fold.apply(w, csr)


# ----------------------------------------------------------------------------------------------------
# derived parameters: folded / weight-normed tensors cached per optimizer step, gradients staged and flushed
# ----------------------------------------------------------------------------------------------------
_PENDING_FLUSH = []
_LOCAL_BACKWARD = [False]


class local_backward:
    """Context for a ``backward()`` whose kernels all run on the CURRENT stream (a pass that lives entirely on one forked
    stream): the end-of-backward flush of derived-parameter gradients then must not join the trainer's other side
    streams (that would tie two forked streams together inside a capture and take them off the trainer's join list)."""

    def __enter__(self):
        self.prev = _LOCAL_BACKWARD[0]
        _LOCAL_BACKWARD[0] = True

    def __exit__(self, *a):
        _LOCAL_BACKWARD[0] = self.prev


_GENERATION = [0]


def begin_step() -> None:
    """Trainer hook at the start of every step: derived tensors are never carried across a step boundary.  (Inside a
    step the producer always precedes its consumers -- also across the hipGraph segments of the step -- so a captured
    graph never reads a derived tensor that only an earlier, un-captured launch wrote.)"""
    _GENERATION[0] += 1


def _source_key(p: Tensor):
    """Identity of a source tensor's current value: its arena's step counter (the fused Adam kernel updates the arena
    through a raw pointer, so torch's version counter does not see it), torch's version counter (copy_,
    load_state_dict, ...), and the storage."""
    ar = getattr(p, '_s2ag_arena', None)
    return (ar.epoch if ar is not None else -1, p._version, p.data_ptr())


def _note_staged(t: Optional[Tensor]) -> None:
    """Called by a backward kernel that has just accumulated into the staging gradient of a derived tensor: the owner
    is flushed when the running backward pass ends (autograd engine callback -- also under hipGraph capture)."""
    owner = getattr(t, '_s2ag_owner', None)
    if owner is None:
        return
    if not _PENDING_FLUSH:
        torch.autograd.Variable._execution_engine.queue_callback(flush_derived)
    if not any(owner is o for o in _PENDING_FLUSH):
        _PENDING_FLUSH.append(owner)


def flush_derived() -> None:
    """Route every staged gradient to its trainable sources (and leave the stages zeroed)."""
    if not _PENDING_FLUSH:
        return
    if not _LOCAL_BACKWARD[0]:
        join_side_streams()      # weight-gradient kernels that fill the stages may still run on forked streams
    while _PENDING_FLUSH:
        _PENDING_FLUSH.pop().flush()


def _leaf_grad(p: Tensor) -> Tensor:
    g = _grad_slot(p)
    if g is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
        g = p.grad
    return g


class _DerivedGroup:
    """Tensors computed from trainable tensors by a fixed map.  ``tensors()`` returns them, recomputing (ONE launch for
    the whole group) only when a source changed -- every forward pass between two optimizer steps shares them.  To the
    autograd ops they are leaves whose ``.grad`` is a persistent staging buffer (zero on entry): weight-gradient
    kernels accumulate into it exactly as into an arena slot, and ``flush()`` (ONE launch) adds the mapped-back
    gradient to the sources' ``.grad`` and clears the stage."""

    def __init__(self, sources):
        self.sources = list(sources)
        self._key = None
        self._out = None
        self._stage = None

    def _alloc(self, shapes, device):
        n = sum(int(np.prod(sh)) for sh in shapes)
        # new values -> new tensors over fresh memory (earlier passes / captured graphs keep the old ones); the stage
        # is allocated (and zeroed) once: every flush leaves it zero
        flat = torch.empty(n, dtype=torch.float32, device=device)
        stage = self._stage
        if stage is None or stage.device != flat.device or stage.numel() != n:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('derived-parameter stages must be created before hipGraph capture '
                                   '(run one eager step first)')
            stage = torch.zeros(n, dtype=torch.float32, device=device)
        out, off = [], 0
        for sh in shapes:
            k = int(np.prod(sh))
            t = flat[off:off + k].view(sh)
            t.requires_grad_(True)
            t.grad = stage[off:off + k].view(sh)
            t._s2ag_owner = self
            out.append(t)
            off += k
        self._out, self._stage = out, stage

    def tensors(self):
        key = (_GENERATION[0],) + tuple(_source_key(p) for p in self.sources)
        if key != self._key:
            with torch.no_grad():
                self._compute()
            self._key = key
        if torch.is_grad_enabled():       # under no_grad no graph is built either way: leave the flags alone
            want = any(p.requires_grad for p in self.sources)
            for t in self._out:
                if t.requires_grad != want:
                    t.requires_grad_(want)
        return self._out


class FoldGroup(_DerivedGroup):
    """y_k = M_k x_k for up to 8 (CSR map, parameter) pairs -- one ST-GCN block's folded weights and biases."""

    def __init__(self, params, csrs, shapes):
        super().__init__(params)
        assert len(params) == len(csrs) == len(shapes) <= L.MAX_JOBS
        self.csrs, self.shapes = list(csrs), [tuple(sh) for sh in shapes]

    def _jobs(self, xs):
        jobs = (L.SpmvJob * len(self.csrs))()
        for k, (csr, x, y) in enumerate(zip(self.csrs, xs, self._out)):
            rp, ci_, va = csr.fwd
            jobs[k] = L.SpmvJob(rp.data_ptr(), ci_.data_ptr(), va.data_ptr(), x.data_ptr(), 0, csr.shape[0])
        return jobs

    def _compute(self):
        dev = self.sources[0].device
        _need_cuda(*self.sources)
        self._alloc(self.shapes, dev)
        srcs = [p.detach().contiguous() for p in self.sources]
        jobs = self._jobs(srcs)
        for k, y in enumerate(self._out):
            jobs[k].y = y.data_ptr()
        L.check(_lib().s2ag_spmv_multi(jobs, len(jobs), 0, _stream()), 'spmv_multi')

    def flush(self):
        grads = [_leaf_grad(p) for p in self.sources]
        jobs = self._jobs(grads)
        for k, y in enumerate(self._out):
            jobs[k].y = y.grad.data_ptr()
        L.check(_lib().s2ag_spmv_multi(jobs, len(jobs), 1, _stream()), 'spmv_multi_flush')


class WeightNormGroup(_DerivedGroup):
    """w_k = g_k v_k / ||v_k|| (torch weight_norm, dim 0) for up to 8 convs; w comes out tap-major (Cout, k, Cin)."""

    def __init__(self, vs, gs):
        super().__init__(list(vs) + list(gs))
        assert len(vs) == len(gs) <= L.MAX_JOBS
        self.vs, self.gs = list(vs), list(gs)
        self._norms = None

    def _jobs(self):
        jobs = (L.WnJob * len(self.vs))()
        for k, (v, g) in enumerate(zip(self.vs, self.gs)):
            rows, cols = v.shape[0], v.numel() // v.shape[0]
            ks = v.shape[2] if v.dim() == 3 else 1
            jobs[k] = L.WnJob(v.data_ptr(), g.data_ptr(), self._out[k].data_ptr(), self._norms[k].data_ptr(), 0, 0, 0,
                              rows, cols, ks)
        return jobs

    def _compute(self):
        dev = self.vs[0].device
        _need_cuda(*self.sources)
        shapes = [(v.shape[0], v.shape[2], v.shape[1]) if v.dim() == 3 else tuple(v.shape) for v in self.vs]
        self._alloc(shapes, dev)
        self._norms = [torch.empty(v.shape[0], dtype=torch.float32, device=dev) for v in self.vs]
        assert all(v.is_contiguous() for v in self.vs)
        jobs = self._jobs()
        L.check(_lib().s2ag_weight_norm_multi(jobs, len(jobs), 0, _stream()), 'weight_norm_multi')

    def flush(self):
        jobs = self._jobs()
        for k, (v, g) in enumerate(zip(self.vs, self.gs)):
            jobs[k].dw = self._out[k].grad.data_ptr()
            jobs[k].dv = _leaf_grad(v).data_ptr()
            jobs[k].dg = _leaf_grad(g).data_ptr()
        L.check(_lib().s2ag_weight_norm_multi(jobs, len(jobs), 1, _stream()), 'weight_norm_multi_flush')


# ----------------------------------------------------------------------------------------------------
# multi-layer bidirectional GRU
# ----------------------------------------------------------------------------------------------------
USE_COOP_GRU = True          # cooperative on-chip-W_hh recurrence where supported (H = 300, 64); else L2-streaming kernels
COOP_GRU_MIN_H = 128         # forward: below this the streaming kernels are used (H = 64 variant exists, measured slower in the full step)
COOP_GRU_BWD_MIN_H = 128     # backward: at H = 64 the streaming BPTT kernel measured faster (tools/bench_gru.py)
_COOP_WS = __import__('collections').deque(maxlen=64)   # recent cooperative workspaces (error words, debugging)


def coop_gru_timeouts() -> int:
    """Number of recent cooperative GRU launches whose peer wait timed out (synchronises).  With the sticky flag armed
    (init_tickets; always in a Processor) this is the flag itself: 0 or 1."""
    lib, bad = _lib(), 0
    torch.cuda.synchronize()
    if _COOP_FLAG:
        return sum(int(f.item() != 0) for f in _COOP_FLAG.values())
    for ws, B, T, H, bwd in list(_COOP_WS):
        off = C.c_longlong(0)
        if bwd < 0:          # a lockstep forward launch over -bwd passes
            L.check(lib.s2ag_gru_coop_fwd_multi_error_word_offset(-bwd, B, T, H, C.byref(off)), 'error_word_offset')
        else:
            L.check(lib.s2ag_gru_coop_error_word_offset(B, T, H, bwd, C.byref(off)), 'error_word_offset')
        bad += int(ws[off.value:off.value + 4].view(torch.int32).item() != 0)
    return bad


def _pair(a: Optional[Tensor], b: Optional[Tensor]) -> Optional[Tensor]:
    """If ``b`` lies directly behind ``a`` in memory (the two directions of a GRU tensor in a parameter arena, see
    net.GRU), return the (2, *shape) view over both -- one GEMM / one reduction then serves both directions."""
    if a is None or b is None or a.shape != b.shape or not a.is_contiguous() or not b.is_contiguous():
        return None
    if a.data_ptr() + 4 * a.numel() != b.data_ptr():
        return None
    if a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr():
        return None
    with torch.no_grad():
        return torch.as_strided(a.detach(), (2,) + tuple(a.shape), (a.numel(),) + tuple(a.stride()))


def _gru_proj_split(inp, wih, wih_r, wih2, bih2, gi, In):
    """gi = inp [W_ih; W_ih_reverse]^T + [b_ih; b_ih_reverse] through the split-operand GEMM.  The planes of the weight
    pair are cached on the forward-direction leaf (the two leaves are adjacent in the arena: one (2*3H, In) matrix)."""
    key = _source_key(wih) + _source_key(wih_r) + ((_GENERATION[0],) if wih.requires_grad else ())
    ent = getattr(wih, '_s2ag_sp2', None)
    if ent is None or ent[0] != key:
        with torch.no_grad():
            ent = (key, split_planes_raw(wih2.reshape(-1, In)))
        wih._s2ag_sp2 = ent
    gemm_split_raw(split_planes_raw(inp), ent[1], bih2.reshape(-1), gi, In)


GRU_WGRAD_TR = True
CONV_WGRAD_TR = True
_GRU_WG_SCRATCH = {}


def _gru_wgrad_scratch(dev, owner, floats):
    """Tile store of a GRU layer's transpose-read weight-gradient launch (one buffer per layer; grown only outside
    hipGraph capture)."""
    key = (dev.index, owner)
    t = _GRU_WG_SCRATCH.get(key)
    if t is None or t.numel() < floats:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('GRU weight-gradient scratch must exist before hipGraph capture (run one eager step first)')
        t = torch.empty(floats, dtype=torch.float32, device=dev)
        _GRU_WG_SCRATCH[key] = t
    return t


class _GRU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, Lyr, training, drop_p, noise, site0, sum_dirs, need_grad, mates, *weights):
        """``mates``: [(x_i, noise_i), ...] -- further passes of the same stack over the same weights, without autograd
        (the trainer's other generator passes of the step): run layer by layer in lockstep with the main pass, their
        recurrences ride in the main pass's cooperative launch (s2ag_gru_coop_fwd_multi).  Returns (out, *mate_outs)
        when given, else out."""
        _need_cuda(x, *weights)
        lib = _lib()
        B, T, I = x.shape
        dev = x.device
        inps = [as_rows(x)[0]] + [as_rows(xm)[0] for xm, _ in mates]
        noises = [noise] + [nm for _, nm in mates]
        nP = len(inps)
        for xm, _ in mates:
            if tuple(xm.shape) != (B, T, I):
                raise ValueError('lockstep passes must share the input shape')
        saved = []
        H3 = 3 * H
        ys = None
        for l in range(Lyr):
            wih, whh, bih, bhh, wih_r, whh_r, bih_r, bhh_r = weights[8 * l:8 * l + 8]
            In = wih.shape[1]
            wih2, bih2 = _pair(wih, wih_r), _pair(bih, bih_r)
            gis = []
            for inp in inps:
                gi = torch.empty(B * T, 2 * H3, dtype=torch.float32, device=dev)
                if (wih2 is not None and bih2 is not None and SPLIT_GEMM and lib.s2ag_gru_coop_split_pieces() != 0
                        and 2.0 * B * T * In * 2 * H3 >= SPLIT_GEMM_MIN_FLOPS):
                    # the big projections are bound by the f32 matrix pipe: same fp32 products on the bf16 pipe from
                    # operands split once (the weight: once per optimizer step), see gemm_sp.hip
                    _gru_proj_split(inp, wih, wih_r, wih2, bih2, gi, In)
                elif wih2 is not None and bih2 is not None:          # both directions' input projections: one GEMM
                    conv_fwd_raw(inp, wih2, bih2, gi, B * T, 1, 1, In, 2 * H3, 1, 1, 0, 1)
                else:
                    conv_fwd_raw(inp, wih, bih, gi[:, :H3], B * T, 1, 1, In, H3, 1, 1, 0, 1)
                    conv_fwd_raw(inp, wih_r, bih_r, gi[:, H3:], B * T, 1, 1, In, H3, 1, 1, 0, 1)
                gis.append(gi)
            whh2 = _pair(whh, whh_r)
            if whh2 is None:
                whh2 = torch.stack((whh, whh_r))
            bhh2 = _pair(bhh, bhh_r)
            if bhh2 is None:
                bhh2 = torch.stack((bhh, bhh_r))
            last = l == Lyr - 1
            use_drop = bool(training) and drop_p > 0 and not last
            ys = [torch.empty(B * T, 2 * H, dtype=torch.float32, device=dev) for _ in range(nP)]
            gates = torch.empty(2, B * T, 4 * H, dtype=torch.float32, device=dev) if need_grad else None
            ydrops = [torch.empty_like(y) if use_drop else None for y in ys]
            coop = USE_COOP_GRU and H >= COOP_GRU_MIN_H and lib.s2ag_gru_coop_supported(H)
            if coop and nP > 1 and lib.s2ag_gru_coop_fwd_multi_supported(nP, B, H):
                arr = lambda ts: (C.c_void_p * nP)(*[None if t is None else t.data_ptr() for t in ts])
                ws = torch.empty(lib.s2ag_gru_coop_fwd_multi_workspace_bytes(nP, B, T, H), dtype=torch.uint8, device=dev)
                L.check(lib.s2ag_gru_coop_fwd_multi(nP, arr(gis), _p(whh2), _p(bhh2), arr(ys), arr(ydrops),
                                                    arr([gates] + [None] * (nP - 1)), B, T, H,
                                                    drop_p if use_drop else 0.0, arr(noises), site0 + l, _p(ws),
                                                    _stream()), 'gru_coop_fwd_multi')
                _COOP_WS.append((ws, B, T, H, -nP))
            else:
                for i in range(nP):
                    e = _epi(L.ACT_NONE, 1.0, drop_p if use_drop else 0.0, noises[i], site0 + l)
                    g_i = gates if i == 0 else None
                    if coop:
                        ws = torch.empty(lib.s2ag_gru_coop_workspace_bytes(B, T, H, 0), dtype=torch.uint8, device=dev)
                        L.check(lib.s2ag_gru_coop_fwd(_p(gis[i]), _p(whh2), _p(bhh2), _p(ys[i]), _p(ydrops[i]), _p(g_i), B,
                                                      T, H, C.byref(e), _p(ws), _stream()), 'gru_coop_fwd')
                        if H > 64:                   # H = 64 runs without an exchange: its error word is never written
                            _COOP_WS.append((ws, B, T, H, 0))
                    else:
                        whhT = None
                        if lib.s2ag_gru_seq_needs_transposed(H):
                            whhT = torch.empty(2, H, H3, dtype=torch.float32, device=dev)
                            transpose_raw(whh2[0], whhT[0])
                            transpose_raw(whh2[1], whhT[1])
                        L.check(lib.s2ag_gru_seq_fwd(_p(gis[i]), _p(whh2), _p(whhT), _p(bhh2), _p(ys[i]), _p(ydrops[i]),
                                                     _p(g_i), B, T, H, C.byref(e), _stream()), 'gru_seq_fwd')
            saved += [inps[0], ys[0], gates]
            inps = ydrops if use_drop else ys
        outs = []
        for y in ys:
            if sum_dirs:
                out = torch.empty(B * T, H, dtype=torch.float32, device=dev)
                add_act_raw(y[:, :H], y[:, H:], out, 1.0)
                outs.append(out.view(B, T, H))
            else:
                outs.append(y.view(B, T, 2 * H))
        ctx.meta = (B, T, H, Lyr, bool(training), float(drop_p), site0, bool(sum_dirs))
        ctx.noise = noise
        ctx.n_w = len(weights)
        ctx.w_leaves = weights
        ctx.n_mates = nP - 1
        ctx.save_for_backward(*weights, *[s for s in saved])
        ctx.none_mask = [s is None for s in saved]
        if nP == 1:
            return outs[0]
        ctx.mark_non_differentiable(*outs[1:])
        return tuple(outs)

    @staticmethod
    def backward(ctx, dout, *_mate_grads):
        B, T, H, Lyr, training, drop_p, site0, sum_dirs = ctx.meta
        lib = _lib()
        tens = ctx.saved_tensors
        weights, saved = tens[:ctx.n_w], tens[ctx.n_w:]
        dev = dout.device
        H3 = 3 * H
        grads = [None] * ctx.n_w
        dy, _, _, lddy = as_rows(dout.reshape(B * T, -1))
        dir_stride = 0 if sum_dirs else H
        dx = None
        for l in range(Lyr - 1, -1, -1):
            wih, whh, bih, bhh, wih_r, whh_r, bih_r, bhh_r = weights[8 * l:8 * l + 8]
            inp, y, gates = saved[3 * l:3 * l + 3]
            In = wih.shape[1]
            last = l == Lyr - 1
            use_drop = training and drop_p > 0 and not last
            whh2 = _pair(whh, whh_r)
            if whh2 is None:
                whh2 = torch.stack((whh, whh_r))
            dgi = torch.empty(B * T, 2 * H3, dtype=torch.float32, device=dev)
            dgh = torch.empty(2, B * T, H3, dtype=torch.float32, device=dev)
            e = _epi(L.ACT_NONE, 1.0, drop_p if use_drop else 0.0, ctx.noise, site0 + l)
            if USE_COOP_GRU and H >= COOP_GRU_BWD_MIN_H and lib.s2ag_gru_coop_supported(H):
                ws = torch.empty(lib.s2ag_gru_coop_workspace_bytes(B, T, H, 1), dtype=torch.uint8, device=dev)
                L.check(lib.s2ag_gru_coop_bwd(_p(dy), lddy, dir_stride, _p(whh2), _p(y), _p(gates), _p(dgi), _p(dgh),
                                              B, T, H, C.byref(e), _p(ws), _stream()), 'gru_coop_bwd')
                if H > 64:
                    _COOP_WS.append((ws, B, T, H, 1))
            else:
                L.check(lib.s2ag_gru_seq_bwd(_p(dy), lddy, dir_stride, _p(whh2), _p(y), _p(gates), _p(dgi), _p(dgh),
                                             B, T, H, C.byref(e), _stream()), 'gru_seq_bwd')
            # input gradient FIRST: it is the critical path (the next layer's recurrence waits for it).  The weight
            # gradients are forked after it, so they run beside the next layer's cooperative recurrence (which leaves
            # 96 CUs idle) instead of competing with this GEMM.
            if l > 0 or ctx.needs_input_grad[0]:
                dx = torch.empty(B * T, In, dtype=torch.float32, device=dev)
                wih2 = _pair(wih, wih_r)
                if (wih2 is not None and SPLIT_GEMM and SPLIT_GEMM_DX and lib.s2ag_gru_coop_split_pieces() != 0
                        and 2.0 * B * T * In * 2 * H3 >= SPLIT_GEMM_MIN_FLOPS):
                    # dx = dgi [W_ih; W_ih_reverse]: the same split-operand GEMM with the transposed weight pair's planes
                    # (refreshed once per optimizer step) and a split pass over dgi
                    wih_leaf, wih_r_leaf = ctx.w_leaves[8 * l], ctx.w_leaves[8 * l + 4]
                    key = _source_key(wih_leaf) + _source_key(wih_r_leaf) + ((_GENERATION[0],) if wih_leaf.requires_grad
                                                                             else ())
                    ent = getattr(wih_leaf, '_s2ag_spT', None)
                    if ent is None or ent[0] != key:
                        with torch.no_grad():
                            wt = torch.empty(In, 2 * H3, dtype=torch.float32, device=dev)
                            transpose_raw(wih2.reshape(-1, In), wt)
                            ent = (key, split_planes_raw(wt))
                        wih_leaf._s2ag_spT = ent
                    gemm_split_raw(split_planes_raw(dgi), ent[1], None, dx, 2 * H3)
                elif wih2 is not None:
                    conv_bwd_data_raw(dgi, wih2, dx, B * T, 1, 1, In, 2 * H3, 1, 1, 0, 1, False)
                else:
                    conv_bwd_data_raw(dgi[:, :H3], wih, dx, B * T, 1, 1, In, H3, 1, 1, 0, 1, False)
                    conv_bwd_data_raw(dgi[:, H3:], wih_r, dx, B * T, 1, 1, In, H3, 1, 1, 0, 1, True)
            # parameter gradients
            base = 8 * l
            need = [ctx.needs_input_grad[10 + base + i] for i in range(8)]
            slots = [_grad_slot(ctx.w_leaves[base + i]) if need[i] else None for i in range(8)]
            pair_ih = _pair(slots[0], slots[4]) if all(need) else None         # (2, 3H, In) gradient of both W_ih
            pair_bi = _pair(slots[2], slots[6]) if all(need) else None
            if all(sl is not None for sl in slots) and pair_ih is not None and pair_bi is not None:
                # arena layout: the two directions are adjacent, so dW_ih / db_ih of both are one launch each
                def leaves(dgi=dgi, dgh=dgh, y=y, inp=inp, In=In, pair_ih=pair_ih, pair_bi=pair_bi, slots=slots, l=l):
                    if (GRU_WGRAD_TR and lib.s2ag_gru_coop_split_pieces() in (1, 2) and T >= 32 and In % 4 == 0 and H % 4 == 0
                            and 2.0 * B * T * (In + H) * 2 * H3 >= SPLIT_GEMM_MIN_FLOPS):
                        # the layer's three weight gradients on the bf16 pipe through the LDS transpose read
                        # (csrc/wgrad_tr.hip, fp32 rows split into two bf16 pieces by the loader): one launch + reduce
                        xin, _, _, ldi = as_rows(inp)
                        jobs = (L.BF16Wgrad * 3)()
                        jobs[0] = L.BF16Wgrad(_p(dgi), _p(xin), _p(pair_ih), _p(pair_bi), B, T, T, T * ldi, ldi, 2 * H3, 1, 0,
                                              0, 1, In, In, 2 * H3, In, In, 0, 1, 0, 1)
                        for d in range(2):
                            jobs[1 + d] = L.BF16Wgrad(C.c_void_p(dgh[d].data_ptr()), C.c_void_p(y.data_ptr() + 4 * d * H),
                                                      _p(slots[4 * d + 1]), _p(slots[4 * d + 3]), B, T, T, T * 2 * H, 2 * H,
                                                      H3, 1, -1 if d == 0 else 1, 0, 1, H, H, H3, H, H, 0, 1, 0, 1)
                        need_f = int(lib.s2ag_f32_wgrad_tr_scratch_floats(jobs, 3))
                        sc = _gru_wgrad_scratch(dev, (id(ctx.w_leaves[8 * l]), B, T), need_f)
                        L.check(lib.s2ag_f32_wgrad_tr(jobs, 3, _p(sc), sc.numel(), _stream()), 'f32_wgrad_tr')
                        return
                    if (SPLIT_WGRAD and SPLIT_GEMM and lib.s2ag_gru_coop_split_pieces() != 0
                            and 2.0 * B * T * In * 2 * H3 >= SPLIT_GEMM_MIN_FLOPS):
                        # weight gradients on the bf16 pipe: transposed splits (contraction over clips * frames; the bias
                        # gradients are the column sums of the same pass), split-K GEMMs accumulating into the arena
                        gT = split_planes_t_raw(dgi, colsum=pair_bi.view(-1))
                        gemm_split_acc_raw(gT, split_planes_t_raw(inp), pair_ih.view(-1, In), B * T)
                        for d in range(2):
                            ghT = split_planes_t_raw(dgh[d], colsum=slots[4 * d + 3])
                            hT = split_planes_t_raw(y[:, d * H:(d + 1) * H], shift=-1 if d == 0 else 1, L_=T)
                            gemm_split_acc_raw(ghT, hT, slots[4 * d + 1], B * T)
                        return
                    if FUSE_WGRADS and conv_bwd_weight_multi_raw(
                            [(dgi, inp, pair_ih, pair_bi.view(-1), (B * T, 1, 1, In, 2 * H3, 1, 1, 0, 1, 0))] +
                            [(dgh[d], y[:, d * H:(d + 1) * H], slots[4 * d + 1], slots[4 * d + 3],
                              (B, T, T, H, H3, 1, 1, 1 if d == 0 else -1, 1, 0)) for d in range(2)]):
                        return                     # dW_ih (both directions) and the two dW_hh in one launch
                    conv_bwd_weight_raw(dgi, inp, pair_ih, B * T, 1, 1, In, 2 * H3, 1, 1, 0, 1, True,
                                        dbias=pair_bi.view(-1))
                    for d in range(2):
                        # dW_hh = sum_t dgh_t^T h_{t-1}: h_{prev} is y shifted by one frame (zero at the boundary)
                        conv_bwd_weight_raw(dgh[d], y[:, d * H:(d + 1) * H], slots[4 * d + 1], B, T, T, H, H3, 1, 1,
                                            1 if d == 0 else -1, 1, True, dbias=slots[4 * d + 3])
                big = H >= COOP_GRU_BWD_MIN_H and Lyr > 1 and 2.0 * B * T * 2 * H3 * (In + H) >= ASYNC_WGRAD_MIN_FLOPS
                if big and l == 0 and GRU_WGRAD_L0 == 2:
                    # the LAST big weight-gradient launch of the pass is issued when the backward pass ends: issued at
                    # its place it ended up ahead of the rest of the backward pass (text TCN, encoders) on one hardware
                    # queue of the replayed graph, and that chain waited ~0.3 ms for it (+2.4 % on the step)
                    defer_wgrad(leaves, keep=(dgi, dgh, y, inp))
                elif not big and GRU_WGRAD_SMALL_DEFER:
                    defer_wgrad(leaves, keep=(dgi, dgh, y, inp))
                else:
                    run_wgrad(leaves, keep=(dgi, dgh, y, inp), flops=2.0 * B * T * 2 * H3 * (In + H))
            else:
                for d, (w_ih, w_hh) in enumerate(((wih, whh), (wih_r, whh_r))):
                    gsl = dgi[:, d * H3:(d + 1) * H3]
                    bd = base + 4 * d
                    if not any(need[4 * d:4 * d + 4]):
                        continue                  # frozen weights (D inside the generator step): no weight-grad kernels
                    sl = slots[4 * d:4 * d + 4]
                    direct = all(q is not None for q in sl)
                    dwi = sl[0] if direct else torch.empty_like(w_ih)
                    dbi = sl[2] if direct else torch.empty(H3, dtype=torch.float32, device=dev)
                    dwh = sl[1] if direct else torch.empty_like(w_hh)
                    dbh = sl[3] if direct else torch.empty(H3, dtype=torch.float32, device=dev)

                    def leaves(gsl=gsl, dwi=dwi, dbi=dbi, dwh=dwh, dbh=dbh, d=d, direct=direct, dgh=dgh, y=y, inp=inp,
                               In=In):
                        conv_bwd_weight_raw(gsl, inp, dwi, B * T, 1, 1, In, H3, 1, 1, 0, 1, direct, dbias=dbi)
                        conv_bwd_weight_raw(dgh[d], y[:, d * H:(d + 1) * H], dwh, B, T, T, H, H3, 1, 1,
                                            1 if d == 0 else -1, 1, direct, dbias=dbh)
                    if direct:       # parameter-gradient leaves: off the dx critical path
                        run_wgrad(leaves, keep=(dgi, dgh, y, inp), flops=2.0 * B * T * 2 * H3 * (In + H))
                    else:
                        leaves()
                        grads[bd], grads[bd + 1], grads[bd + 2], grads[bd + 3] = dwi, dwh, dbi, dbh
            dy, lddy, dir_stride = dx, 2 * H, H
        dxo = dx.view(B, T, -1) if ctx.needs_input_grad[0] else None
        return (dxo, None, None, None, None, None, None, None, None, None, *grads)


def gru(x: Tensor, weights: Sequence[Tensor], hidden: int, layers: int, training: bool, drop_p: float, noise,
        site0: int, sum_dirs: bool, mates=()):
    """weights: per layer [w_ih, w_hh, b_ih, b_hh, w_ih_reverse, w_hh_reverse, b_ih_reverse, b_hh_reverse].
    ``mates`` = [(x_i, noise_i), ...]: no-grad passes of the same stack run in lockstep (see _GRU.forward); the result
    is then (out, *mate_outs)."""
    need_grad = torch.is_grad_enabled() and (x.requires_grad or any(w.requires_grad for w in weights))
    mates = tuple((xm.detach(), nm) for xm, nm in mates)
    return _GRU.apply(x, hidden, layers, bool(training), float(drop_p), noise, site0, bool(sum_dirs), need_grad,
                      mates, *weights)


# ----------------------------------------------------------------------------------------------------
# re-parametrisation and losses
# ----------------------------------------------------------------------------------------------------
class _Reparam(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, log_var, noise, site):
        _need_cuda(mu, log_var)
        mu, log_var = mu.contiguous(), log_var.contiguous()
        z = torch.empty_like(mu)
        L.check(_lib().s2ag_reparam_fwd(_p(mu), _p(log_var), mu.numel(), _p(noise), site, _p(z), _stream()),
                'reparam_fwd')
        ctx.save_for_backward(log_var)
        ctx.noise, ctx.site = noise, site
        return z

    @staticmethod
    def backward(ctx, dz):
        (log_var,) = ctx.saved_tensors
        dz = dz.contiguous()
        dmu = torch.zeros_like(log_var)
        dlv = torch.zeros_like(log_var)
        L.check(_lib().s2ag_reparam_bwd(_p(dz), _p(log_var), dz.numel(), _p(ctx.noise), ctx.site, _p(dmu), _p(dlv),
                                        _stream()), 'reparam_bwd')
        return dmu, dlv, None, None


def reparametrize(mu: Tensor, log_var: Tensor, noise: Tensor, site: int) -> Tensor:
    return _Reparam.apply(mu, log_var, noise, site)


def make_pre_seq(target: Tensor, n_pre: int) -> Tensor:
    """(B, T, D) target poses -> (B, T, D + 1) seed sequence: the first ``n_pre`` frames with 1 in the extra column, zero
    elsewhere (processor_v2.py:786-789) -- one launch."""
    _need_cuda(target)
    target = target.contiguous()
    B, T, D = target.shape
    pre = torch.empty(B, T, D + 1, dtype=torch.float32, device=target.device)
    L.check(_lib().s2ag_make_pre_seq(_p(target), _p(pre), B, T, D, int(n_pre), _stream()), 'make_pre_seq')
    return pre


class _ContextCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, *parts):
        """parts: (B, T, c_k) tensors; z: (B, cz) or None -- broadcast over the frames and appended last."""
        B, T = parts[0].shape[:2]
        srcs = [as_rows(t) for t in parts]
        if z is not None:
            srcs.append(as_rows(z))
        n = len(srcs)
        total = sum(sr[2] for sr in srcs)
        out = torch.empty(B, T, total, dtype=torch.float32, device=parts[0].device)
        L.check(_lib().s2ag_concat_cols((C.c_void_p * n)(*[sr[0].data_ptr() for sr in srcs]),
                                        (C.c_int * n)(*[sr[2] for sr in srcs]), (C.c_int * n)(*[sr[3] for sr in srcs]),
                                        (C.c_int * n)(*([0] * len(parts) + ([1] if z is not None else []))), n, _p(out),
                                        B * T, T, _stream()), 'concat_cols')
        ctx.cols = [sr[2] for sr in srcs]
        ctx.has_z = z is not None
        ctx.bt = (B, T)
        return out

    @staticmethod
    def backward(ctx, g):
        B, T = ctx.bt
        g = g.contiguous()
        grads, c0 = [], 0
        n_parts = len(ctx.cols) - (1 if ctx.has_z else 0)
        for k in range(n_parts):
            grads.append(g[..., c0:c0 + ctx.cols[k]] if ctx.needs_input_grad[1 + k] else None)   # views, like CatBackward
            c0 += ctx.cols[k]
        dz = None
        if ctx.has_z and ctx.needs_input_grad[0]:
            cz = ctx.cols[-1]
            dz = torch.empty(B, cz, dtype=torch.float32, device=g.device)
            L.check(_lib().s2ag_sum_frames(_p(g), g.shape[-1], c0, cz, B, T, _p(dz), _stream()), 'sum_frames')
        return (dz, *grads)


def context_cat(parts: Sequence[Tensor], z: Optional[Tensor] = None) -> Tensor:
    """[parts ... | z broadcast over the frames] along the channel axis in ONE launch (the reference's two torch.cat +
    repeat, net/multimodal_context_net_v2.py:522-536); the gradient of z is its sum over the frames."""
    _need_cuda(*parts)
    if len(parts) + (z is not None) > 4 or (z is not None and z.shape[-1] > 64):
        x = torch.cat(tuple(parts), dim=2)
        return x if z is None else torch.cat((x, z.unsqueeze(1).expand(-1, x.shape[1], -1)), dim=2)
    return _ContextCat.apply(z, *parts)


_UNIT_ROOT = [False]
_ONES = {}


def backward_from(loss: Tensor) -> None:
    """``loss.backward()`` for a loss made by the functions below, without the ATen launches of the implicit root: the
    gradient of the root is a persistent 1.0 (no fill) and the loss functions hand out the gradients their forward
    kernel already wrote instead of multiplying them by that 1.0 (was: 1 fill + 1-4 elementwise launches per call, on
    the critical path between the loss and the first backward kernel)."""
    key = (loss.device, loss.dtype)
    one = _ONES.get(key)
    if one is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('run one eager step before hipGraph capture (persistent root gradient)')
        one = torch.ones((), dtype=loss.dtype, device=loss.device)
        _ONES[key] = one
    _UNIT_ROOT[0] = True
    try:
        torch.autograd.backward([loss], [one.view(loss.shape)])
    finally:
        _UNIT_ROOT[0] = False


class _DisLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d_real, d_fake):
        _need_cuda(d_real, d_fake)
        d_real, d_fake = d_real.contiguous(), d_fake.contiguous()
        B = d_real.numel()
        loss = torch.empty(1, dtype=torch.float32, device=d_real.device)
        gr, gf = torch.empty_like(d_real), torch.empty_like(d_fake)
        L.check(_lib().s2ag_dis_loss(_p(d_real), _p(d_fake), B, _p(loss), _p(gr), _p(gf), _stream()), 'dis_loss')
        ctx.save_for_backward(gr, gf)
        return loss.view(())

    @staticmethod
    def backward(ctx, dl):
        gr, gf = ctx.saved_tensors
        if _UNIT_ROOT[0]:
            return gr, gf
        return gr * dl, gf * dl


def dis_loss(d_real: Tensor, d_fake: Tensor) -> Tensor:
    return _DisLoss.apply(d_real, d_fake)


class _DisLossHalf(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d, real):
        _need_cuda(d)
        d = d.contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=d.device)
        g = torch.empty_like(d)
        a = (_p(d), None) if real else (None, _p(d))
        b = (_p(g), None) if real else (None, _p(g))
        L.check(_lib().s2ag_dis_loss(a[0], a[1], d.numel(), _p(loss), b[0], b[1], _stream()), 'dis_loss')
        ctx.save_for_backward(g)
        return loss.view(())

    @staticmethod
    def backward(ctx, dl):
        (g,) = ctx.saved_tensors
        if _UNIT_ROOT[0]:
            return g, None
        return g * dl, None


def dis_loss_half(d: Tensor, real: bool) -> Tensor:
    """-mean(log(d + 1e-8)) for the real half, -mean(log(1 - d + 1e-8)) for the fake half: their sum is ``dis_loss``."""
    return _DisLossHalf.apply(d, bool(real))


class _GenLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, dis_out, mu, log_var, target, out_tri, out_rand, z, z_rand, weights):
        reg = out_rand is not None          # False: the branch without the regulariser (processor_v2.py:933-934)
        if reg:
            _need_cuda(out, dis_out, mu, log_var)
            mu, log_var, out_rand, z, z_rand = [t.contiguous() for t in (mu, log_var, out_rand, z, z_rand)]
        else:
            _need_cuda(out, dis_out)
            mu = log_var = z = z_rand = None
        out, dis_out, target = out.contiguous(), dis_out.contiguous(), target.contiguous()
        out_tri = out_tri.contiguous() if out_tri is not None else None
        B = out.shape[0]
        TP = out.numel() // B
        ZD = mu.numel() // B if reg else 1
        dev = out.device
        scratch = torch.empty(B, 8, dtype=torch.float32, device=dev)
        comps = torch.empty(8, dtype=torch.float32, device=dev)
        g_out, g_dis = torch.empty_like(out), torch.empty_like(dis_out)
        g_mu, g_lv = (torch.empty_like(mu), torch.empty_like(log_var)) if reg else (None, None)
        w = (C.c_float * 4)(*[float(v) for v in weights])
        L.check(_lib().s2ag_gen_loss(_p(out), _p(target), _p(out_tri), _p(dis_out), _p(out_rand), _p(z), _p(z_rand),
                                     _p(mu), _p(log_var), B, TP, ZD, w, _p(scratch), _p(comps), _p(g_out), _p(g_dis),
                                     _p(g_mu), _p(g_lv), _stream()), 'gen_loss')
        ctx.reg = reg
        ctx.save_for_backward(*((g_out, g_dis, g_mu, g_lv) if reg else (g_out, g_dis)))
        ctx.mark_non_differentiable(comps)
        return comps[0].clone().view(()), comps

    @staticmethod
    def backward(ctx, dl, _dc):
        g = list(ctx.saved_tensors) + ([] if ctx.reg else [None, None])
        if not _UNIT_ROOT[0]:
            g = [None if t is None else t * dl for t in g]
        return (*g, None, None, None, None, None, None)


def gen_loss(out, dis_out, mu, log_var, target, out_tri, out_rand, z, z_rand, weights):
    """Returns (total, comps[8]); comps = {total, huber, gen_error, div_reg, kld, l1, l1_tri, 0}.
    weights = (regression, gan, div_reg, kld).  ``out_rand=None``: the branch without the regulariser
    (processor_v2.py:933-934) -- total = regression + GAN term; the divergence and KLD terms are not evaluated (upstream
    never forms exp(z_log_var) there, so an overflowing log-variance cannot poison the loss), mu / log_var receive no
    gradient from the loss, ``z`` / ``z_rand`` / ``mu`` / ``log_var`` and weights[2:] are ignored."""
    if out_rand is None:
        return _GenLoss.apply(out, dis_out, None, None, target.detach(), None if out_tri is None else out_tri.detach(),
                              None, None, None, tuple(weights))
    return _GenLoss.apply(out, dis_out, mu, log_var, target.detach(), None if out_tri is None else out_tri.detach(),
                          out_rand.detach(), z.detach(), z_rand.detach(), tuple(weights))


# ----------------------------------------------------------------------------------------------------
# evaluation metrics (processor_v2.py:738-774)
# ----------------------------------------------------------------------------------------------------
_POSE_MEAN = {}


def pose_metrics(out: Tensor, target: Tensor, mean_dir_vec, n_pre: int) -> Tensor:
    """(3,) float64 on the device: L1 of the direction vectors, MAE of the joint coordinates behind the seed poses and the
    acceleration difference of processor_v2.py:738-774 (push_samples); ``mean_dir_vec``: 27 numbers (any nesting)."""
    _need_cuda(out, target)
    import numpy as np
    B, T, P = out.shape
    assert P == 27 and target.shape == out.shape, (out.shape, target.shape)
    mean = np.ascontiguousarray(np.array(mean_dir_vec, dtype=np.float64).reshape(-1))
    assert mean.size == 27
    ent = _POSE_MEAN.get(out.device.index)
    if ent is None or not np.array_equal(ent[0], mean):
        ent = _POSE_MEAN[out.device.index] = (mean, torch.from_numpy(mean).to(out.device))
    sums = torch.empty(3, dtype=torch.float64, device=out.device)
    L.check(_lib().s2ag_pose_metrics(_p(out.contiguous().float()), _p(target.contiguous().float()), _p(ent[1]), B, T,
                                     int(n_pre), _p(sums), _stream()), 'pose_metrics')
    return sums / sums.new_tensor([B * T * 27, B * (T - int(n_pre)) * 30, B * (T - 2) * 30])


# ----------------------------------------------------------------------------------------------------
# stream-level parallelism of independent branches
# ----------------------------------------------------------------------------------------------------
_DIRTY_STREAMS = []      # side streams that carried work since the last join_side_streams()
_KEEPALIVE = []          # tensors read by side-stream kernels: released only after the join
ASYNC_WGRAD = True       # weight/bias gradient kernels (leaves of the backward graph) run beside the dx chain
_WGRAD_STREAMS = {}
_MAIN_STREAM = [None]


_DET_WORDS = {}
_DET_ON = [False]        # the explicit state of the mode (never inferred from ASYNC_WGRAD, which tests may set on their own)


def det_flavour() -> bool:
    """True when the loaded library is the deterministic BUILD FLAVOUR (libs2ag_hip_det.so: build.py --det, selected with
    S2AG_HIP_LIB).  The release library carries no trace of the mode in its kernels (include/s2ag_hip.h)."""
    return bool(_lib().s2ag_det_flavour())


def set_deterministic(on: bool, device=None) -> None:
    """Deterministic mode (config switch DETERMINISTIC; debug; needs the det build flavour): the library orders every
    accumulating launch's atomics by workgroup index (s2ag_set_deterministic: csrc/s2ag_common.h det_enter / det_leave /
    S2AG_DET_WAVES_BEGIN..END) and weight-gradient kernels stay on the stream of their backward pass.  The trainer additionally runs
    the passes of a step on ONE stream (Processor(..., deterministic=True)): two runs from the same state then give
    bit-identical gradients and weights, at the price of serialised accumulation phases.  Both precision modes.
    The mode is process-wide library state: every Processor sets it to ITS value on construction (on or off), and while it
    is on nothing may be launched on a side stream (all accumulating launches share one turn word; two concurrent launches
    would take each other's turns) -- run_wgrad / wgrad_launcher stay inline and mark_side_stream raises."""
    global ASYNC_WGRAD
    if not on:
        if _DET_ON[0]:                     # (only what switching it ON changed is changed back)
            L.check(_lib().s2ag_set_deterministic(None, None), 'set_deterministic')
            ASYNC_WGRAD = True
        _DET_ON[0] = False
        return
    if not det_flavour():
        raise RuntimeError('deterministic mode is a build flavour of the library: build it with `python -m '
                           'speech2affective_gestures_amd.build --det` and start the process with '
                           'S2AG_HIP_LIB=<package>/libs2ag_hip_det.so (the release kernels carry no ordering code)')
    dev = torch.device('cuda' if device is None else device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if _DIRTY_STREAMS:
        raise RuntimeError('set_deterministic(True) with work pending on side streams: join_side_streams() first')
    if key not in _DET_WORDS:
        _DET_WORDS[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    init_tickets(dev)                                     # the sticky error word: bit 3 = a workgroup's turn never came
    # re-arming after a time-out: once bit 3 is up det_enter stops waiting while det_leave keeps storing, so the turn word can
    # be left anywhere; switching the mode on again starts from a zero turn word and a clear bit 3 (the other bits of the
    # error word -- cooperative GRU / BatchNorm time-outs -- are the trainer's to read and are left alone)
    torch.cuda.synchronize(dev)
    _DET_WORDS[key].zero_()
    _COOP_FLAG[key].bitwise_and_(~8)
    L.check(_lib().s2ag_set_deterministic(_p(_DET_WORDS[key]), _p(_COOP_FLAG[key])), 'set_deterministic')
    _DET_ON[0] = True
    ASYNC_WGRAD = False


def deterministic() -> bool:
    return _DET_ON[0]


def set_main_stream(stream=None) -> None:
    """The trainer names the stream its step runs on; only from there are weight-gradient kernels forked (forking
    again from an already forked stream crashes hipGraph capture on ROCm 7.2)."""
    _MAIN_STREAM[0] = stream if stream is not None else torch.cuda.current_stream()


ASYNC_WGRAD_MIN_FLOPS = 3e9


def run_wgrad(fn, keep=(), flops=float('inf')) -> None:
    """Launch ``fn`` (kernels that only ACCUMULATE into parameter gradients) on the weight-gradient stream if we are
    on the main stream, else inline.  Nothing downstream in the backward pass depends on them; the trainer joins the
    stream (join_side_streams) before the optimizer reads the gradient arena."""
    cur = torch.cuda.current_stream()
    main = _MAIN_STREAM[0]
    if _DET_ON[0] or not ASYNC_WGRAD or main is None or cur != main or flops < ASYNC_WGRAD_MIN_FLOPS:
        fn()
        return
    dev = cur.device_index
    if dev not in _WGRAD_STREAMS:
        _WGRAD_STREAMS[dev] = torch.cuda.Stream(device=dev)
    s = _WGRAD_STREAMS[dev]
    s.wait_stream(cur)
    mark_side_stream(s)
    _KEEPALIVE.extend(keep)
    with torch.cuda.stream(s):
        fn()


GRU_WGRAD_L0 = 2
GRU_WGRAD_SMALL_DEFER = True


def wgrad_launcher(fn, keep=()):
    """Bind ``fn`` (weight-gradient kernels, see run_wgrad) to what has been queued on the current stream SO FAR (an
    event, not the stream) and return a callable that issues it on the weight-gradient stream -- to be called after the
    next kernels of the dependent chain have been issued.  Why the order of ISSUE matters: hipGraph (ROCm 7.2) walks
    the captured graph depth first and lets the FIRST captured successor of a node inherit its hardware queue, every
    further successor takes the next of the four queues; a side chain issued right at the fork therefore keeps the
    queue and the dependent chain hops to another one at every fork (15-35 us per cross-queue edge, and chains that
    happen to land on one queue serialise).  Returns None when forking is not armed (then ``fn`` has run inline)."""
    cur = torch.cuda.current_stream()
    main = _MAIN_STREAM[0]
    if _DET_ON[0] or not ASYNC_WGRAD or main is None or cur != main:
        fn()
        return None
    ev = torch.cuda.Event()
    ev.record(cur)
    dev = cur.device_index

    def later(keep=tuple(keep)):
        if dev not in _WGRAD_STREAMS:
            _WGRAD_STREAMS[dev] = torch.cuda.Stream(device=dev)
        s = _WGRAD_STREAMS[dev]
        s.wait_event(ev)
        mark_side_stream(s)
        _KEEPALIVE.extend(keep)
        with torch.cuda.stream(s):
            fn()
    return later


def defer_wgrad(fn, keep=()) -> None:
    """Like run_wgrad, but ``fn`` is ISSUED when the running backward pass ends (autograd engine callback), ordered only
    behind what has been queued on the current stream so far."""
    later = wgrad_launcher(fn, keep)
    if later is not None:
        torch.autograd.Variable._execution_engine.queue_callback(later)


def mark_side_stream(s) -> None:
    if _DET_ON[0]:
        raise RuntimeError('deterministic mode is on: every launch must stay on one stream (all accumulating launches '
                           'share one turn word); construct the trainer with deterministic=True or switch the mode off')
    if not any(s is t for t in _DIRTY_STREAMS):
        _DIRTY_STREAMS.append(s)


def join_side_streams() -> None:
    """Make the current stream wait for every side stream used since the last call.  Needed after ``backward()``:
    backward kernels run on the stream of their forward op, and because weight gradients are accumulated in place
    (no AccumulateGrad node) autograd does not join those streams itself."""
    cur = torch.cuda.current_stream()
    while _DIRTY_STREAMS:
        s = _DIRTY_STREAMS.pop()
        if s is not cur:
            cur.wait_stream(s)
    _KEEPALIVE.clear()
    _MAIN_STREAM[0] = None       # forking is armed per trainer phase: set_main_stream() ... join_side_streams()


class BranchStreams:
    """A few side streams owned by one module.  ``run([f0, f1, ...])`` executes f0 on the current stream and the
    others on the side streams (forked after everything queued so far, joined before returning) -- the encoder
    branches of a generator are chains of small kernels that each fill a fraction of the chip.  Capturable: forks
    and joins are event waits that hipGraph records as edges."""

    def __init__(self, n_side: int):
        self.n_side = n_side
        self._streams = {}

    def _get(self, device):
        key = (device.index if device.index is not None else torch.cuda.current_device())
        if key not in self._streams:
            self._streams[key] = [torch.cuda.Stream(device=device) for _ in range(self.n_side)]
        return self._streams[key]

    def run(self, fns, device, enabled=True):
        if not enabled or len(fns) <= 1 or _NO_BRANCH_DEPTH[0] > 0:
            return [f() for f in fns]
        cur = torch.cuda.current_stream(device)
        sides = self._get(device)
        out = [None] * len(fns)
        used = []
        for i, f in enumerate(fns[1:]):
            s = sides[i % len(sides)]
            s.wait_stream(cur)
            mark_side_stream(s)
            with torch.cuda.stream(s):
                out[i + 1] = f()
            used.append(s)
        out[0] = fns[0]()
        for s in used:
            cur.wait_stream(s)
        return out


PARALLEL_BRANCHES = False     # measured: fork/join overhead exceeds the overlap gained (20.9 vs 19.0 ms/step); kept for study
_NO_BRANCH_DEPTH = [0]


class sequential_branches:
    """Context: modules inside run their branches sequentially.  The trainer wraps passes that already run on a forked
    stream with it -- forking again from a fork (the same module's branch streams entered from two different parent
    streams inside one capture) crashes hipGraph capture on ROCm 7.2."""

    def __enter__(self):
        _NO_BRANCH_DEPTH[0] += 1

    def __exit__(self, *a):
        _NO_BRANCH_DEPTH[0] -= 1
