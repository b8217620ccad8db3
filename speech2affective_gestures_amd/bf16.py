"""bf16 mode of the Conv1d hot path: host side of csrc/conv_bf16.hip.

``S2AG_PRECISION=bf16`` (or ``with bf16.precision('bf16'):``) makes the wave encoder and the text TCN keep their
activations in HBM as bf16 (fp32 accumulation on the bf16 matrix pipe, BatchNorm statistics from fp64 column sums of the
rounded outputs, fp32 master weights converted once per optimizer step, fp32 weight gradients) -- BASELINE configs[1] /
configs[3] "bf16", SURVEY section 7 hard part 5.  The reference has no reduced-precision path, so this mode has its own
parity tests with their own, measured tolerance (tests/test_gpu_bf16.py); fp32 stays the default and the mode in which
the 1e-3 bar is proven.  What enters and leaves the two encoders is fp32 (raw audio, token ids -> (B, T, 32) features).

Layout: an activation is a torch.bfloat16 tensor (clips, frames, Cp); where a layer is read tap by tap (the TCN) Cp is the
channel count rounded up to 32 and the pad channels are zeros (every producer writes them).
"""
import ctypes as C
import os
from typing import Optional

import torch

from . import _lib as L
from . import config
from . import ops

Tensor = torch.Tensor
_ENV = str(config.get('PRECISION')).lower()
_MODE = [_ENV in ('bf16', 'bfloat16', 'bf16_step')]
# 'bf16_step' (BASELINE configs[1] "bf16"): besides the bf16 Conv1d path, every large matrix product of the step -- the
# cooperative GRU's recurrence, its input projections and input gradients, the GRU / TCN / wave-encoder weight gradients
# -- takes its fp32 operands as ONE bf16 piece (one product on the bf16 matrix pipe instead of the default three of the
# two-piece split); storage, accumulation, BatchNorm statistics, master weights and Adam stay fp32.
_STEP = [_ENV == 'bf16_step']


def enabled() -> bool:
    return _MODE[0]


def step_mode() -> bool:
    return _STEP[0]




class precision:
    """Context manager: ``precision('bf16')`` (the Conv1d path in bf16), ``precision('bf16_step')`` (that + single-piece
    bf16 products in the GRU, its projections and the weight gradients) / ``precision('fp32')``."""

    def __init__(self, mode: str):
        m = mode.lower()
        self.on = m in ('bf16', 'bfloat16', 'bf16_step')
        self.step = m == 'bf16_step'

    def __enter__(self):
        self.prev = (_MODE[0], _STEP[0])
        # the split-piece count is a library-wide setting: step mode forces one piece; LEAVING step mode (a nested
        # precision('fp32') / ('bf16'), or either under S2AG_PRECISION=bf16_step) must give the products their default
        # pieces back, or an 'fp32' reference run would silently keep 8-mantissa-bit products
        if self.step:
            self.prev_pieces = _lib().s2ag_gru_coop_set_split_pieces(1)
        elif self.prev[1]:
            self.prev_pieces = _lib().s2ag_gru_coop_set_split_pieces(-1)      # -1: S2AG_GRU_SPLIT or the default two
        else:
            self.prev_pieces = None
        _MODE[0], _STEP[0] = self.on, self.step

    def __exit__(self, *a):
        _MODE[0], _STEP[0] = self.prev
        if self.prev_pieces is not None:
            _lib().s2ag_gru_coop_set_split_pieces(self.prev_pieces)


def pad32(c: int) -> int:
    return (c + 31) // 32 * 32


def _lib():
    return L.load()


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


if _STEP[0]:                                   # S2AG_PRECISION=bf16_step: single-piece products from the first launch on
    _lib().s2ag_gru_coop_set_split_pieces(1)


def _rows16(t: Tensor):
    """(tensor, rows, ld) of a contiguous bf16 activation whose last axis is the (padded) channel axis."""
    assert t.dtype == torch.bfloat16 and t.is_cuda, 'bf16 activation on the GPU expected'
    if not t.is_contiguous():
        t = t.contiguous()
    return t, t.numel() // t.shape[-1], t.shape[-1]


# ----------------------------------------------------------------------------------------------------
# casts
# ----------------------------------------------------------------------------------------------------
def to_bf16_raw(x: Tensor, ld: Optional[int] = None) -> Tensor:
    x, rows, cols, ldx = ops.as_rows(x)
    ld = cols if ld is None else ld
    y = torch.empty(rows, ld, dtype=torch.bfloat16, device=x.device)
    L.check(_lib().s2ag_bf16_cast(_p(x), ldx, rows, cols, _p(y), ld, 1, _s()), 'bf16_cast')
    return y


def to_f32_raw(x: Tensor, cols: Optional[int] = None) -> Tensor:
    x, rows, ld = _rows16(x)
    cols = ld if cols is None else cols
    y = torch.empty(rows, cols, dtype=torch.float32, device=x.device)
    L.check(_lib().s2ag_bf16_cast(_p(x), ld, rows, cols, _p(y), cols, 0, _s()), 'bf16_cast')
    return y


class _ToBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ld):
        ctx.shape = x.shape
        return to_bf16_raw(x, ld).view(*x.shape[:-1], ld)

    @staticmethod
    def backward(ctx, dy):
        return to_f32_raw(dy, ctx.shape[-1]).view(ctx.shape), None


class _ToF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cols):
        ctx.ld = x.shape[-1]
        ctx.shape = x.shape
        return to_f32_raw(x, cols).view(*x.shape[:-1], cols)

    @staticmethod
    def backward(ctx, dy):
        return to_bf16_raw(dy.contiguous(), ctx.ld).view(ctx.shape), None


def to_bf16(x: Tensor, ld: Optional[int] = None) -> Tensor:
    return _ToBF16.apply(x, x.shape[-1] if ld is None else ld)


def to_f32(x: Tensor, cols: Optional[int] = None) -> Tensor:
    return _ToF32.apply(x, x.shape[-1] if cols is None else cols)


# ----------------------------------------------------------------------------------------------------
# weights: fp32 masters -> bf16 operand layouts, one launch per group and optimizer step
# ----------------------------------------------------------------------------------------------------
class WeightPack:
    """bf16 operand layouts of a module's conv weights.  ``add`` registers what a layer needs; ``get(name)`` returns
    the tensors, refreshing ALL registered layouts with one launch when a source changed (optimizer step, load) or a
    new training step began (ops.begin_step: derived tensors are never carried across steps)."""

    def __init__(self):
        self.entries = {}          # name -> dict(src_fn, jobs=[(key, shape, fields)], )
        self._key = None
        self._out = {}

    def add(self, name, src_fn, layout, Cout, Cin, ks, stride=1):
        """``src_fn()`` -> the fp32 weight; ``layout``: 'tap_major' (Cout, ks, Cin) or 'reference' (Cout, Cin, ks)."""
        if name not in self.entries:
            self.entries[name] = dict(src=src_fn, layout=layout, Cout=Cout, Cin=Cin, ks=ks, stride=stride)

    def _jobs(self, e, w):
        Cout, Cin, ks, s = e['Cout'], e['Cin'], e['ks'], e['stride']
        if e['layout'] == 'tap_major':
            so, st, sc = ks * Cin, Cin, 1
        else:
            so, st, sc = Cin * ks, 1, ks
        jobs = []
        flat = e['layout'] == 'reference' and Cin % 8 == 0          # wave-encoder convs: contiguous windows
        if flat:
            Kp = (ks * Cin + 63) // 64 * 64
            jobs.append(('fwd', (Cout, 1, Kp), dict(rows=Cout, taps=1, Cp=Kp, cols=Cin, tap0=0, tap_step=1, src_taps=ks,
                                                    s_o=so, s_t=st, s_c=sc, flat_cin=Cin)))
        else:
            Cp = pad32(Cin)
            jobs.append(('fwd', (Cout, ks, Cp), dict(rows=Cout, taps=ks, Cp=Cp, cols=Cin, tap0=0, tap_step=1, src_taps=ks,
                                                     s_o=so, s_t=st, s_c=sc, flat_cin=0)))
        CpO = pad32(Cout)
        if s == 1:      # data gradient = the same kernel with tap-flipped, transposed weights
            jobs.append(('dgrad', (Cin, ks, CpO), dict(rows=Cin, taps=ks, Cp=CpO, cols=Cout, tap0=ks - 1, tap_step=-1,
                                                       src_taps=ks, s_o=sc, s_t=st, s_c=so, flat_cin=0)))
        else:           # poly-phase: phase r uses the taps r, r + s, r + 2s, ...
            nt = (ks + s - 1) // s
            for r in range(s):
                jobs.append((('phase', r), (Cin, nt, CpO), dict(rows=Cin, taps=nt, Cp=CpO, cols=Cout, tap0=r, tap_step=s,
                                                               src_taps=ks, s_o=sc, s_t=st, s_c=so, flat_cin=0)))
        return jobs

    def _refresh(self):
        srcs = {n: e['src']() for n, e in self.entries.items()}
        key = (ops.generation(),) + tuple((id(w), w._version, w.data_ptr()) for w in srcs.values())
        if key == self._key:
            return
        with torch.no_grad():
            plan = []
            for n, e in self.entries.items():
                for tag, shape, f in self._jobs(e, srcs[n]):
                    plan.append((n, tag, shape, f, srcs[n]))
            assert len(plan) <= L.BF16_MAX_PACK, 'too many weight layouts in one WeightPack'
            dev = next(iter(srcs.values())).device
            total = sum(sh[0] * sh[1] * sh[2] for _, _, sh, _, _ in plan)
            flat = torch.empty(total + 64, dtype=torch.bfloat16, device=dev)      # + slack: nothing reads past it, but cheap
            arr = (L.BF16PackJob * len(plan))()
            out, off = {}, 0
            for k, (n, tag, sh, f, w) in enumerate(plan):
                cnt = sh[0] * sh[1] * sh[2]
                t = flat[off:off + cnt].view(sh)
                arr[k] = L.BF16PackJob(w.detach().data_ptr(), t.data_ptr(), f['rows'], f['taps'], f['Cp'], f['cols'],
                                       f['tap0'], f['tap_step'], f['src_taps'], f['s_o'], f['s_t'], f['s_c'],
                                       f['flat_cin'])
                if isinstance(tag, tuple):          # phases of one layer are adjacent: one (phases, Cin, nt, CpO) tensor
                    if tag[1] == 0:
                        s = self.entries[n]['stride']
                        out[(n, 'phases')] = flat[off:off + s * cnt].view((s,) + sh)
                else:
                    out[(n, tag)] = t
                off += cnt
            L.check(_lib().s2ag_bf16_pack_weights(arr, len(plan), _s()), 'bf16_pack_weights')
        self._out, self._key = out, key

    def get(self, name, what):
        self._refresh()
        return self._out[(name, what)]


# ----------------------------------------------------------------------------------------------------
# conv (implicit GEMM), forward + both gradients
# ----------------------------------------------------------------------------------------------------
def _conv_launch(x, w16, bias, y, N, Lq, Lin, x_clip, ldx, pos, ks, Cp, Cvalid, Cout, CoutS, y_clip, y_row, out_f32,
                 epi=None, phases=1, w_phase=0, y_phase=0, q_total=0, mask_cols=0, want_stats=False, post=None):
    lib = _lib()
    py, pact, pcols, pslope, pdrop, prng, psite = (None, 0, 0, 1.0, 0.0, None, 0) if post is None else post
    a = L.BF16Conv(_p(x), _p(w16), _p(bias), _p(y), N, Lq, Lin, x_clip, ldx, pos[0], pos[1], pos[2], ks, Cp, Cvalid, Cout,
                   CoutS, y_clip, y_row, 0, int(out_f32), phases, w_phase, y_phase, q_total, mask_cols, _p(py), int(pact),
                   int(pcols), float(pslope), float(pdrop), _p(prng) if pdrop > 0 else None, int(psite))
    e = epi if epi is not None else L.Epilogue(L.ACT_NONE, 1.0, 0.0, None, 0)
    if want_stats:
        rows = lib.s2ag_bf16_conv_stats_rows(N * Lq)
        part = torch.empty(2 * rows * Cout, dtype=torch.float64, device=x.device)
        got = C.c_int(0)
        L.check(lib.s2ag_bf16_conv(C.byref(a), C.byref(e), _p(part), C.byref(got), _s()), 'bf16_conv')
        return part[:2 * got.value * Cout], got.value
    L.check(lib.s2ag_bf16_conv(C.byref(a), C.byref(e), None, None, _s()), 'bf16_conv')
    return None


class _Conv16(torch.autograd.Function):
    """x (N, Lin, ldx) bf16 -> y (N, Lout, CoutS) bf16 [or fp32 (N, Lout, Cout)].  ``w`` is the fp32 leaf (or derived
    tensor) whose .grad slot receives the weight gradient; ``pack`` / ``name`` give its bf16 layouts."""

    @staticmethod
    def forward(ctx, x, w, bias, pack, name, geom, epi, flags):
        N, Lin, Lout, Cin, Cout, ks, stride, pad, dil = geom
        act, slope, drop_p, noise, site = epi
        out_f32, want_stats, pad_out = flags
        x, rows, ldx = _rows16(x)
        assert rows == N * Lin
        w16 = pack.get(name, 'fwd')
        flat = w16.shape[1] == 1 and ks > 1
        if flat:
            assert ldx == Cin and pad == 0 and dil == 1
            Cp, Cvalid, kk, pos = w16.shape[2], ks * Cin, 1, (stride, 0, 0)
        else:
            Cp, kk, pos = w16.shape[2], ks, (stride, -pad, dil)
            Cvalid = min(Cp, ldx)
            assert ldx % 8 == 0 and ldx >= Cin
        CoutS = Cout if out_f32 else (pad32(Cout) if pad_out else Cout)
        y = torch.empty(N * Lout, CoutS, dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
        e = L.Epilogue(act, float(slope), float(drop_p), _p(noise) if drop_p > 0 else None, int(site))
        st = _conv_launch(x, w16, bias, y, N, Lout, Lin, Lin * ldx, ldx, pos, kk, Cp, Cvalid, Cout, CoutS, Lout * CoutS,
                          CoutS, out_f32, e, mask_cols=Cout, want_stats=want_stats)
        ctx.geom, ctx.epi, ctx.noise = geom, (act, slope, drop_p, site), noise
        ctx.pack, ctx.name, ctx.flat, ctx.out_f32, ctx.ldx, ctx.CoutS = pack, name, flat, out_f32, ldx, CoutS
        ctx.w_leaf, ctx.b_leaf = w, bias
        need_y = (act == L.ACT_LEAKY and not out_f32)
        ctx.save_for_backward(x, y if need_y else None)
        # epilogue-backward fusion: if x is the output of a _Conv16 with an activation / dropout epilogue (it told us
        # through _PRODUCER), this layer's data gradient applies that epilogue's derivative itself
        ctx.prev = _PRODUCER.pop(x.data_ptr(), None) if (stride == 1 and FUSE_EPILOGUE_BWD) else None
        ctx.token = None
        if (need_y or drop_p > 0) and not out_f32:
            ctx.token = object()
            _PRODUCER.clear() if len(_PRODUCER) > 64 else None
            _PRODUCER[y.data_ptr()] = (y, act, Cout, float(slope), float(drop_p), noise, int(site), ctx.token)
        _Conv16.last_stats = st
        return y.view(N, Lout, CoutS)

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        N, Lin, Lout, Cin, Cout, ks, stride, pad, dil = ctx.geom
        act, slope, drop_p, site = ctx.epi
        lib = _lib()
        pack, name = ctx.pack, ctx.name
        if ctx.out_f32:                         # fp32 gradient of an fp32 output (the decoder Linear): one cast
            g = to_bf16_raw(dy.reshape(N * Lout, Cout).contiguous(), pad32(Cout))
            ldg = g.shape[1]
        else:
            dy, _, ldg = _rows16(dy)
            if ctx.token is not None and _FUSED.pop(ctx.token, None) == dy.data_ptr():
                g = dy                          # the consumer's data gradient already applied this layer's epilogue
            elif act != L.ACT_NONE or drop_p > 0:
                g = torch.empty_like(dy)
                e = L.Epilogue(act, float(slope), float(drop_p), _p(ctx.noise) if drop_p > 0 else None, int(site))
                L.check(lib.s2ag_bf16_epilogue_bwd(_p(dy), _p(y), N * Lout, Cout, ldg, C.byref(e), _p(g), _s()),
                        'bf16_epilogue_bwd')
            else:
                g = dy
        dx = None
        if ctx.needs_input_grad[0]:
            ldx = ctx.ldx
            dx = torch.empty(N * Lin, ldx, dtype=torch.bfloat16, device=g.device)
            CpO = pad32(Cout)
            gv = min(CpO, ldg)
            if stride == 1:
                wd = pack.get(name, 'dgrad')                                   # (Cin, ks, CpO)
                post = None
                if ctx.prev is not None and ctx.prev[0].shape[-1] == ldx:
                    post = ctx.prev[:7]
                    if len(_FUSED) > 64:
                        _FUSED.clear()
                    _FUSED[ctx.prev[7]] = dx.data_ptr()
                _conv_launch(g, wd, None, dx, N, Lin, Lout, Lout * ldg, ldg, (1, pad - (ks - 1) * dil, dil), ks, CpO, gv,
                             Cin, ldx, Lin * ldx, ldx, False, post=post)
            else:
                assert pad == 0 and dil == 1
                wph = pack.get(name, 'phases')                                 # (s, Cin, nt, CpO)
                nt = wph.shape[2]
                Lq = (Lin + stride - 1) // stride
                _conv_launch(g, wph, None, dx, N, Lq, Lout, Lout * ldg, ldg, (1, 0, -1), nt, CpO, gv, Cin, ldx, Lin * ldx,
                             stride * ldx, False, phases=stride, w_phase=Cin * nt * CpO, y_phase=ldx, q_total=Lin)
            dx = dx.view(N, Lin, ldx)
        # weight (+ bias) gradient, accumulated into the leaf's gradient slot
        wleaf, bleaf = ctx.w_leaf, ctx.b_leaf
        dw = db = None
        if ctx.needs_input_grad[1] or (bleaf is not None and ctx.needs_input_grad[2]):
            wslot = ops._grad_slot(wleaf) if ctx.needs_input_grad[1] else None
            bslot = ops._grad_slot(bleaf) if (bleaf is not None and ctx.needs_input_grad[2]) else None
            if ctx.needs_input_grad[1] and wslot is None:
                dw = torch.zeros_like(wleaf)
                wslot = dw
            if bleaf is not None and ctx.needs_input_grad[2] and bslot is None:
                db = torch.zeros_like(bleaf)
                bslot = db
            tap_major = pack.entries[name]['layout'] == 'tap_major'
            if wleaf.dim() == 2:
                d_co, d_t, d_c = Cin, 0, 1
            elif tap_major:
                d_co, d_t, d_c = ks * Cin, Cin, 1
            else:
                d_co, d_t, d_c = Cin * ks, 1, ks
            if ctx.flat:
                Kp64 = (ks * Cin + 63) // 64 * 64
                a = L.BF16Wgrad(_p(g), _p(x), _p(wslot), _p(bslot), N, Lout, Lin, Lin * ctx.ldx, ctx.ldx, ldg, stride, 0, 0,
                                1, Kp64, ks * Cin, Cout, Cin, d_co, d_t, d_c, Cin, ks)
            else:
                Cp64 = (max(ctx.ldx, Cin) + 63) // 64 * 64
                a = L.BF16Wgrad(_p(g), _p(x), _p(wslot), _p(bslot), N, Lout, Lin, Lin * ctx.ldx, ctx.ldx, ldg, stride, -pad,
                                dil, ks, Cp64, ctx.ldx, Cout, Cin, d_co, d_t, d_c, 0, ks)
            if wslot is not None:
                sc = None
                if ctx.flat and WGRAD_TR and Lout >= 32:      # (the transpose-read loader steps 32 rows with one clip wrap)
                    sc = _wgrad_scratch(g.device, (id(pack), name), int(lib.s2ag_bf16_conv_wgrad_tr_scratch_floats(C.byref(a), 1)))
                elif ctx.flat and SPLIT_WGRAD:
                    sc = _wgrad_scratch(g.device, (id(pack), name), int(lib.s2ag_bf16_conv_wgrad_scratch_floats(C.byref(a))))

                def launch(a=a, sc=sc):
                    if sc is not None and WGRAD_TR and Lout >= 32:
                        L.check(lib.s2ag_bf16_conv_wgrad_tr(C.byref(a), 1, _p(sc), sc.numel(), _s()), 'bf16_conv_wgrad_tr')
                    elif sc is not None:
                        L.check(lib.s2ag_bf16_conv_wgrad_split(C.byref(a), _p(sc), sc.numel(), _s()), 'bf16_conv_wgrad_split')
                    else:
                        L.check(lib.s2ag_bf16_conv_wgrad(C.byref(a), _s()), 'bf16_conv_wgrad')
                # a leaf of the backward graph: big ones run beside the data-gradient chain (ops.run_wgrad forks them to
                # the weight-gradient stream when the trainer armed it; the trainer joins before the optimizer)
                ops.run_wgrad(launch, keep=(g, x), flops=2.0 * N * Lout * Cout * Cin * ks)
                if dw is None:
                    ops._note_staged(wleaf)
                if bslot is not None and db is None:
                    ops._note_staged(bleaf)
            elif bslot is not None:
                ops.colsum_raw(to_f32_raw(g, Cout), bslot, accumulate=True)
        return dx, dw, db, None, None, None, None, None


def conv(x: Tensor, w: Tensor, bias: Optional[Tensor], pack: WeightPack, name: str, Cin: int, Cout: int, ks: int,
         stride=1, pad=0, dil=1, lout=None, act=L.ACT_NONE, slope=1.0, drop_p=0.0, noise=None, site=0, out_f32=False,
         bn_stats=False, pad_out=False):
    """bf16 conv1d on channels-last rows.  Returns y, with ``y._s2ag_stats`` set when ``bn_stats``."""
    N, Lin = x.shape[0], x.shape[1]
    if lout is None:
        lout = (Lin + 2 * pad - dil * (ks - 1) - 1) // stride + 1
    y = _Conv16.apply(x, w, bias, pack, name, (N, Lin, lout, Cin, Cout, ks, stride, pad, dil),
                      (act, float(slope), float(drop_p), noise, site), (bool(out_f32), bool(bn_stats), bool(pad_out)))
    if bn_stats and _Conv16.last_stats is not None:
        y._s2ag_stats = _Conv16.last_stats
    _Conv16.last_stats = None
    return y


_Conv16.last_stats = None
SPLIT_WGRAD = True
WGRAD_TR = True      # LDS transpose-read kernels (csrc/wgrad_tr.hip)
_WG_SCRATCH = {}


def _wgrad_scratch(dev, owner, floats):
    """Tile store of the split weight-gradient launches: one buffer per layer (passes of a step that run on forked streams
    never share one), grown only outside hipGraph capture."""
    key = (dev.index, owner)
    t = _WG_SCRATCH.get(key)
    if t is None or t.numel() < floats:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('bf16 weight-gradient scratch must exist before hipGraph capture (run one eager step first)')
        t = torch.empty(floats, dtype=torch.float32, device=dev)
        _WG_SCRATCH[key] = t
    return t


FUSE_EPILOGUE_BWD = True
_PRODUCER = {}      # data_ptr of a conv output -> (y, act, cols, slope, drop_p, noise, site, token) of its epilogue
_FUSED = {}         # token -> data_ptr of the gradient tensor that already carries that epilogue's derivative


# ----------------------------------------------------------------------------------------------------
# clip-resident TemporalConvNet (csrc/tcn_fused.hip): all blocks in one launch forward, one for the data gradients
# ----------------------------------------------------------------------------------------------------
FUSE_TCN = True


def tcn_fused_supported(T: int, C: int, ks: int, n_blocks: int) -> bool:
    return (FUSE_TCN and 256 < C <= 320 and 1 <= n_blocks <= L.TCN_MAX_BLOCKS
            and _lib().s2ag_bf16_tcn_clips_per_block(int(T), int(C), int(ks)) > 0)


class TcnFragments:
    """The 2 * n_blocks normalised conv weights in MFMA-fragment order (forward and data-gradient operand of every conv),
    refreshed with one launch when a weight changed or a new step began (like WeightPack)."""

    def __init__(self):
        self._key, self._frag = None, None

    def get(self, ws):
        key = (ops.generation(),) + tuple((id(w), w._version, w.data_ptr()) for w in ws)
        if key != self._key:
            lib = _lib()
            C_ = ws[0].shape[0]
            assert all(tuple(w.shape) == (C_, 2, C_) and w.is_contiguous() for w in ws), 'tap-major (C, 2, C) weights'
            frag = torch.empty(int(lib.s2ag_bf16_tcn_pack_elems(len(ws))), dtype=torch.bfloat16, device=ws[0].device)
            ptrs = (C.c_void_p * len(ws))(*[w.detach().data_ptr() for w in ws])
            L.check(lib.s2ag_bf16_tcn_pack(ptrs, len(ws), C_, _p(frag), _s()), 'bf16_tcn_pack')
            self._key, self._frag = key, frag
        return self._frag


class _TcnFused16(torch.autograd.Function):
    """x (N, T, 320) bf16 -> y (N, T, 320) bf16 through all TemporalBlocks.  ``params`` = the 2*nb normalised weights
    (fp32 (C, 2, C) gradient stages) followed by the 2*nb biases."""

    @staticmethod
    def forward(ctx, x, frags, meta, noise, *params):
        dils, sites, drop_p = meta
        nb = len(dils)
        ws, bs = params[:2 * nb], params[2 * nb:]
        x, rows, ld = _rows16(x)
        N, T = ctx_shape = x.shape[0], x.shape[1]
        assert ld == 320 and x.dim() == 3
        C_ = ws[0].shape[0]
        frag = frags.get(ws)
        saved = torch.empty(2 * nb, rows, 320, dtype=torch.bfloat16, device=x.device)      # h1[b], y[b]
        sb = int(_lib().s2ag_bf16_tcn_sign_bytes(N, T))
        signs = torch.empty(nb, sb, dtype=torch.uint8, device=x.device)
        a = L.BF16Tcn()
        a.x, a.wfrag = x.data_ptr(), frag.data_ptr()
        for b in range(nb):
            a.h1[b], a.sign[b], a.y[b] = saved[2 * b].data_ptr(), signs[b].data_ptr(), saved[2 * b + 1].data_ptr()
            a.dil[b] = int(dils[b])
            for j in range(2):
                a.bias[2 * b + j] = bs[2 * b + j].data_ptr() if bs[2 * b + j] is not None else None
                a.site[2 * b + j] = int(sites[2 * b + j])
        a.n_blocks, a.n_clips, a.T, a.C = nb, N, T, C_
        a.drop_p = float(drop_p)
        a.rng = noise.data_ptr() if drop_p > 0 else None
        if drop_p > 0:
            keep = torch.empty(int(_lib().s2ag_bf16_tcn_keep_bytes(N, T, nb)), dtype=torch.uint8, device=x.device)
            a.keep = keep.data_ptr()
        L.check(_lib().s2ag_bf16_tcn_fwd(C.byref(a), _s()), 'bf16_tcn_fwd')
        ctx.meta, ctx.frags, ctx.noise, ctx.params, ctx.shape = meta, frags, noise, params, ctx_shape
        ctx.save_for_backward(x, saved, signs)
        return saved[2 * nb - 1].view(N, T, 320)

    @staticmethod
    def backward(ctx, gy):
        x, saved, signs = ctx.saved_tensors
        dils, sites, drop_p = ctx.meta
        nb = len(dils)
        params = ctx.params
        ws, bs = params[:2 * nb], params[2 * nb:]
        N, T = ctx.shape
        rows = N * T
        C_ = ws[0].shape[0]
        lib = _lib()
        gy, _, ldg = _rows16(gy)
        assert ldg == 320
        gx = torch.empty(rows, 320, dtype=torch.bfloat16, device=gy.device)
        gp = torch.empty(2 * nb, rows, 320, dtype=torch.bfloat16, device=gy.device)
        a = L.BF16Tcn()
        a.x, a.wfrag = x.data_ptr(), ctx.frags.get(ws).data_ptr()
        for b in range(nb):
            a.h1[b], a.sign[b], a.y[b] = saved[2 * b].data_ptr(), signs[b].data_ptr(), saved[2 * b + 1].data_ptr()
            a.gp1[b], a.gp2[b] = gp[2 * b].data_ptr(), gp[2 * b + 1].data_ptr()
            a.dil[b] = int(dils[b])
        a.n_blocks, a.n_clips, a.T, a.C = nb, N, T, C_
        a.drop_p = float(drop_p)
        a.rng = ctx.noise.data_ptr() if drop_p > 0 else None
        a.gy, a.gx = gy.data_ptr(), gx.data_ptr()
        L.check(lib.s2ag_bf16_tcn_bwd(C.byref(a), _s()), 'bf16_tcn_bwd')
        # the 2*nb weight (+ bias) gradients: one launch, accumulated into the leaves' gradient slots
        grads = [None] * (4 * nb)
        jobs = (L.BF16Wgrad * (2 * nb))()
        nj = 0
        for b in range(nb):
            for j in range(2):
                k = 2 * b + j
                need_w = ctx.needs_input_grad[4 + k]
                need_b = bs[k] is not None and ctx.needs_input_grad[4 + 2 * nb + k]
                if not (need_w or need_b):
                    continue
                wslot = ops._grad_slot(ws[k]) if need_w else None
                bslot = ops._grad_slot(bs[k]) if need_b else None
                if need_w and wslot is None:
                    grads[k] = wslot = torch.zeros_like(ws[k])
                if need_b and bslot is None:
                    grads[2 * nb + k] = bslot = torch.zeros_like(bs[k])
                if wslot is None:                                   # bias only: rare (frozen weights), plain column sum
                    ops.colsum_raw(to_f32_raw(gp[k], C_), bslot, accumulate=True)
                    continue
                xin = (x.view(rows, 320) if b == 0 else saved[2 * b - 1]) if j == 0 else saved[2 * b]
                d = int(dils[b])
                jobs[nj] = L.BF16Wgrad(_p(gp[k]), _p(xin), _p(wslot), _p(bslot), N, T, T, T * 320, 320, 320, 1, -d, d, 2, 320,
                                       320, C_, C_, 2 * C_, C_, 1, 0, 2)
                nj += 1
                if grads[k] is None:
                    ops._note_staged(ws[k])
        if nj:
            sc = None
            if WGRAD_TR and T >= 32:
                sc = _wgrad_scratch(gy.device, (id(ctx.frags), 'tcn'), int(lib.s2ag_bf16_conv_wgrad_tr_scratch_floats(jobs, nj)))

            def launch(jobs=jobs, nj=nj, sc=sc):
                if sc is not None:
                    L.check(lib.s2ag_bf16_conv_wgrad_tr(jobs, nj, _p(sc), sc.numel(), _s()), 'bf16_conv_wgrad_tr')
                else:
                    L.check(lib.s2ag_bf16_conv_wgrad_multi(jobs, nj, _s()), 'bf16_conv_wgrad_multi')
            ops.run_wgrad(launch, keep=(gp, saved, x), flops=2.0 * rows * C_ * C_ * 2 * nj)
        return (gx.view(N, T, 320), None, None, None) + tuple(grads)


def tcn_fused(x: Tensor, frags: TcnFragments, ws, biases, dils, sites, drop_p: float, noise) -> Tensor:
    return _TcnFused16.apply(x, frags, (tuple(dils), tuple(sites), float(drop_p)), noise, *ws, *biases)


# ----------------------------------------------------------------------------------------------------
# the one-channel wave conv (conv1 of the wave encoder): fp32 waveform in, bf16 out
# ----------------------------------------------------------------------------------------------------
class _ConvC1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, wav, w, bias, stride, pad, want_stats):
        lib = _lib()
        N, Lin = wav.shape
        Cout, _, ks = w.shape
        Lout = (Lin + 2 * pad - (ks - 1) - 1) // stride + 1
        wav = wav.contiguous()
        g = L.ConvGeom(N, Lin, Lout, 1, Cout, ks, stride, pad, 1, 1, Cout, 0)
        y = torch.empty(N, Lout, Cout, dtype=torch.bfloat16, device=wav.device)
        st = None
        if want_stats:
            rows = lib.s2ag_bf16_conv_c1_rows(C.byref(g))
            part = torch.empty(2 * rows * Cout, dtype=torch.float64, device=wav.device)
            got = C.c_int(0)
            L.check(lib.s2ag_bf16_conv_c1_fwd(_p(wav), _p(w), _p(bias), _p(y), C.byref(g), _p(part), C.byref(got), _s()),
                    'bf16_conv_c1_fwd')
            st = (part[:2 * got.value * Cout], got.value)
        else:
            L.check(lib.s2ag_bf16_conv_c1_fwd(_p(wav), _p(w), _p(bias), _p(y), C.byref(g), None, None, _s()),
                    'bf16_conv_c1_fwd')
        ctx.geom = (N, Lin, Lout, Cout, ks, stride, pad)
        ctx.w_leaf, ctx.b_leaf = w, bias
        ctx.save_for_backward(wav)
        _ConvC1.last_stats = st
        return y

    @staticmethod
    def backward(ctx, dy):
        (wav,) = ctx.saved_tensors
        N, Lin, Lout, Cout, ks, stride, pad = ctx.geom
        dy, _, ldg = _rows16(dy)
        g = L.ConvGeom(N, Lin, Lout, 1, Cout, ks, stride, pad, 1, 1, ldg, 0)
        w, b = ctx.w_leaf, ctx.b_leaf
        wslot = ops._grad_slot(w)
        bslot = ops._grad_slot(b) if b is not None else None
        dw = db = None
        if wslot is None:
            dw = torch.zeros_like(w)
            wslot = dw
        if b is not None and bslot is None:
            db = torch.zeros_like(b)
            bslot = db
        L.check(_lib().s2ag_bf16_conv_c1_wgrad(_p(dy), _p(wav), _p(wslot), _p(bslot), C.byref(g), _s()), 'bf16_conv_c1_wgrad')
        return None, dw, db, None, None, None


_ConvC1.last_stats = None


def conv_c1(wav: Tensor, w: Tensor, bias: Optional[Tensor], stride: int, pad: int, bn_stats=False) -> Tensor:
    y = _ConvC1.apply(wav, w, bias, int(stride), int(pad), bool(bn_stats))
    if bn_stats and _ConvC1.last_stats is not None:
        y._s2ag_stats = _ConvC1.last_stats
    _ConvC1.last_stats = None
    return y


# ----------------------------------------------------------------------------------------------------
# BatchNorm (+ leaky) on bf16 rows
# ----------------------------------------------------------------------------------------------------
_BN_SCRATCH = {}


def _bn_scratch(dev, cols):
    key = (dev.index, cols)
    if key not in _BN_SCRATCH:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('bf16 BatchNorm scratch must exist before hipGraph capture (run one eager step first)')
        _BN_SCRATCH[key] = torch.zeros(2 * cols, dtype=torch.float32, device=dev)
    return _BN_SCRATCH[key]


class _BNAct16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, nbt, slope, training, eps, momentum, stats):
        lib = _lib()
        shape = x.shape
        x, rows, ld = _rows16(x)
        cols = gamma.numel()
        assert ld == cols and cols % 8 == 0, 'bf16 BatchNorm: unpadded channel axis, multiple of 8'
        dev = x.device
        coef = torch.empty(4, cols, dtype=torch.float32, device=dev)
        if training:
            assert stats is not None, 'bf16 BatchNorm takes its batch statistics from the producing conv'
            part, prow = stats
            L.check(lib.s2ag_bn_fold(_p(part), int(prow), rows, cols, None, cols, _p(gamma), _p(beta), _p(rmean), _p(rvar),
                                     _p(nbt), float(eps), float(momentum), int(ops._BN_REPEAT[0]), _p(coef[0]),
                                     _p(coef[1]), _p(coef[2]), _p(coef[3]), _s()), 'bn_fold')
        else:
            L.check(lib.s2ag_bn_coeffs(None, None, None, cols, cols, rows, _p(gamma), _p(beta), _p(rmean), _p(rvar), None,
                                       float(eps), float(momentum), 0, _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]),
                                       _s()), 'bn_coeffs')
        y = torch.empty_like(x)
        L.check(lib.s2ag_bf16_bn_apply(_p(x), rows, cols, ld, _p(coef[0]), _p(coef[1]), float(slope), _p(y), _s()),
                'bf16_bn_apply')
        ctx.save_for_backward(x, coef)
        ctx.meta = (rows, cols, ld, float(slope), bool(training))
        ctx.leaves = (gamma, beta)
        _bn_scratch(dev, cols)
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x, coef = ctx.saved_tensors
        rows, cols, ld, slope, training = ctx.meta
        shape = dy.shape
        dy, _, _ = _rows16(dy)
        dev = dy.device
        dx = torch.empty_like(dy)
        sg, sb = ops._grad_slot(ctx.leaves[0]), ops._grad_slot(ctx.leaves[1])
        dgamma = dbeta = None
        if sg is None or sb is None:
            dgb = torch.zeros(2, cols, dtype=torch.float32, device=dev)
            sg, sb = dgb[0], dgb[1]
            dgamma, dbeta = dgb[0], dgb[1]
        tmp = torch.empty(2, cols, dtype=torch.float32, device=dev)
        assert training, 'bf16 BatchNorm backward is only defined in training mode'
        L.check(_lib().s2ag_bf16_bn_bwd(_p(x), _p(dy), rows, cols, ld, _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]),
                                        slope, _p(sg), _p(sb), _p(_bn_scratch(dev, cols)), _p(tmp[0]), _p(tmp[1]),
                                        _p(dx), _s()), 'bf16_bn_bwd')
        return dx.view(shape), dgamma, dbeta, None, None, None, None, None, None, None, None


def batch_norm_act(x: Tensor, bn: torch.nn.Module, slope: float = 1.0) -> Tensor:
    tr = bn.training
    stats = getattr(x, '_s2ag_stats', None) if tr else None
    return _BNAct16.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked if tr else None,
                          float(slope), bool(tr), float(bn.eps), float(bn.momentum), stats)


# ----------------------------------------------------------------------------------------------------
# residual add + ReLU, embedding
# ----------------------------------------------------------------------------------------------------
class _AddAct16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, slope, cols):
        a_, rows, ld = _rows16(a)
        b_, _, _ = _rows16(b)
        y = torch.empty_like(a_)
        L.check(_lib().s2ag_bf16_add_act(_p(a_), _p(b_), a_.numel(), float(slope), _p(y), _s()), 'bf16_add_act')
        ctx.meta = (rows, cols, ld, float(slope))
        ctx.save_for_backward(y)
        return y.view(a.shape)

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        rows, cols, ld, slope = ctx.meta
        dy, _, _ = _rows16(dy)
        g = torch.empty_like(dy)
        e = L.Epilogue(L.ACT_LEAKY, slope, 0.0, None, 0)
        L.check(_lib().s2ag_bf16_epilogue_bwd(_p(dy), _p(y), rows, cols, ld, C.byref(e), _p(g), _s()), 'bf16_epilogue_bwd')
        return g, g, None, None


def add_act(a: Tensor, b: Tensor, slope: float, cols: int) -> Tensor:
    return _AddAct16.apply(a, b, float(slope), int(cols))


class _Embedding16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, table, drop_p, noise, site, ld):
        ids_ = ids.contiguous().view(-1)
        rows, dim = ids_.numel(), table.shape[1]
        out = torch.empty(rows, ld, dtype=torch.bfloat16, device=table.device)
        e = L.Epilogue(L.ACT_NONE, 1.0, float(drop_p), _p(noise) if drop_p > 0 else None, int(site))
        L.check(_lib().s2ag_bf16_embedding_fwd(_p(ids_), _p(table), rows, dim, table.shape[0], _p(out), ld, C.byref(e), _s()),
                'bf16_embedding_fwd')
        ctx.save_for_backward(ids_)
        ctx.meta = (table.shape[0], dim, float(drop_p), int(site), ld)
        ctx.noise, ctx.table_leaf = noise, table
        return out.view(*ids.shape, ld)

    @staticmethod
    def backward(ctx, dy):
        (ids_,) = ctx.saved_tensors
        n_entries, dim, drop_p, site, ld = ctx.meta
        dy, rows, _ = _rows16(dy)
        e = L.Epilogue(L.ACT_NONE, 1.0, drop_p, _p(ctx.noise) if drop_p > 0 else None, site)
        slot = ops._grad_slot(ctx.table_leaf)
        dt = None
        if slot is None:
            dt = torch.zeros(n_entries, dim, dtype=torch.float32, device=dy.device)
            slot = dt
        L.check(_lib().s2ag_bf16_embedding_bwd(_p(ids_), _p(dy), ld, rows, dim, n_entries, _p(slot), C.byref(e), _s()),
                'bf16_embedding_bwd')
        return None, dt, None, None, None, None


def embedding(ids: Tensor, table: Tensor, drop_p: float = 0.0, noise=None, site=0) -> Tensor:
    return _Embedding16.apply(ids, table, float(drop_p), noise, int(site), pad32(table.shape[1]))


# ----------------------------------------------------------------------------------------------------
# the wave encoder with BatchNorm folded into the neighbouring convs (csrc/wave_fused.hip)
# ----------------------------------------------------------------------------------------------------
WAVE_FUSED = config.mirror('WAVE_FUSED', globals(), 'WAVE_FUSED')
_WAVE_LAYERS = ((16, 32), (32, 64), (64, 32))          # (Cin, Cout) of conv2..4: 15 taps, stride 6, no padding


def wave_fused_supported(fe) -> bool:
    """``fe`` = WavEncoder.feat_extractor (net/multimodal_context_net_v2.py:17-28)."""
    if not WAVE_FUSED:
        return False
    c1 = fe[0]
    ok = (c1.in_channels, c1.out_channels, c1.kernel_size[0], c1.stride[0]) == (1, 16, 15, 5)
    for i, (ci, co) in zip((3, 6, 9), _WAVE_LAYERS):
        c = fe[i]
        ok = ok and (c.in_channels, c.out_channels, c.kernel_size[0], c.stride[0], c.padding[0], c.dilation[0]) == \
            (ci, co, 15, 6, 0, 1)
    from . import wave12                       # BatchNorm / activation flavour the fused launches fold
    slopes = [wave12.leaky_slope(fe[i]) for i in (2, 5, 8)]
    return ok and all(wave12.bn_foldable(fe[i]) for i in (1, 4, 7)) and slopes[0] is not None and slopes == [slopes[0]] * 3


class _WaveFused16(torch.autograd.Function):
    """WavEncoder.forward in training mode: (N, samples) fp32 -> (N, frames, 32) fp32.  The three BatchNorms never run as
    kernels of their own: forward, a conv stores its raw output (bf16) + column-sum partials and the next conv applies
    scale / shift / LeakyReLU in its loader; backward, a data gradient emits dz = da * leaky'(.) + the column sums of dz
    and dz * xhat, and the consumers of dy = A dz + C y + B form it in their loaders (wave_fused.hip)."""

    @staticmethod
    def forward(ctx, wav, pack, bns, slope, *params):
        lib = _lib()
        w1, b1, g1, e1, w2, b2, g2, e2, w3, b3, g3, e3, w4, b4 = params
        N, Lin0 = wav.shape
        dev = wav.device
        wav = wav.contiguous()
        pad1 = 1600
        L1 = (Lin0 + 2 * pad1 - 15) // 5 + 1
        lens = [L1]
        for _ in _WAVE_LAYERS:
            lens.append((lens[-1] - 15) // 6 + 1)
        geom1 = L.ConvGeom(N, Lin0, L1, 1, 16, 15, 5, pad1, 1, 1, 16, 0)
        gb = ((g1, e1), (g2, e2), (g3, e3))
        keep = []                                                   # ctypes structs must outlive their launches

        def fold_args(k, prow):
            """BatchNorm k's fold, done by the producing conv's last workgroup: -> (coef (4, C), s2ag_bn_fold_args)"""
            bn, (gamma, beta) = bns[k], gb[k]
            coef = torch.empty(4, gamma.numel(), dtype=torch.float32, device=dev)
            fa = L.BnFoldArgs(ops._tickets(dev, 1 + (prow + 15) // 16), _p(gamma), _p(beta), _p(bn.running_mean), _p(bn.running_var),
                              _p(bn.num_batches_tracked), float(bn.eps), float(bn.momentum), int(ops._BN_REPEAT[0]),
                              _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]))
            keep.append(fa)
            return coef, fa
        from . import wave12
        use12 = wave12.ENABLED and b1 is not None
        if use12:
            # conv1's (N, L1, 16) output never exists in HBM: statistics pass, then conv1 -> BatchNorm 1 -> conv2 in one
            # launch; the backward pass recomputes it from the waveform (wave12.hip)
            pk12 = wave12.packed_weights(w1, w2)
            coef = wave12.stats(wav, pk12, b1, bns[0], g1, e1, True, pad1)
            y2, _, _, coef2 = wave12.forward(wav, pk12, b1, coef, slope, b2, False, fold=(bns[1], g2, e2), pad=pad1)
            ys, coefs = [pk12, y2], [coef, coef2]
        else:
            y1 = torch.empty(N, L1, 16, dtype=torch.bfloat16, device=dev)
            prow = lib.s2ag_wave_conv1_fwd_rows(C.byref(geom1))
            part = torch.empty(2 * (prow + (prow + 15) // 16) * 16, dtype=torch.float64, device=dev)
            coef, fa = fold_args(0, prow)
            L.check(lib.s2ag_wave_conv1_fwd(_p(wav), _p(w1), _p(b1), _p(y1), C.byref(geom1), _p(part), C.byref(fa), _s()),
                    'wave_conv1_fwd')
            ys, coefs = [y1], [coef]
        names = ('c3', 'c6', 'c9')
        bs = (b2, b3, b4)
        for k, (ci, co) in enumerate(_WAVE_LAYERS):
            if use12 and k == 0:
                continue
            last = k == 2
            w16 = pack.get(names[k], 'fwd')                         # (co, 1, KP)
            Lin, Lout = lens[k], lens[k + 1]
            y = torch.empty(N, Lout, co, dtype=torch.float32 if last else torch.bfloat16, device=dev)
            st, fa, coef = None, None, None
            if not last:
                prow = lib.s2ag_wave_fwd_rows(N, Lout, ci, co)
                st = torch.empty(2 * (prow + (prow + 15) // 16) * co, dtype=torch.float64, device=dev)
                coef, fa = fold_args(k + 1, prow)
            L.check(lib.s2ag_wave_conv_fwd(_p(ys[-1]), _p(coefs[-1][0]), _p(coefs[-1][1]), float(slope), _p(w16),
                                           int(w16.shape[2]), _p(bs[k]), _p(y), int(last), _p(st),
                                           C.byref(fa) if fa is not None else None, N, Lin, Lout, ci, co, _s()), 'wave_conv_fwd')
            if not last:
                coefs.append(coef)
                ys.append(y)
        ctx.pack, ctx.slope, ctx.lens, ctx.params, ctx.pad1, ctx.use12 = pack, float(slope), lens, params, pad1, use12
        ctx.save_for_backward(wav, *ys, *coefs)             # ys[0]: conv1's output -- or, without it, the packed weights
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        wav, y1, y2, y3, c1, c2, c3 = ctx.saved_tensors
        w1, b1, g1, e1, w2, b2, g2, e2, w3, b3, g3, e3, w4, b4 = ctx.params
        lens, slope, pack = ctx.lens, ctx.slope, ctx.pack
        N, dev = wav.shape[0], wav.device
        g = g.contiguous().float()
        grads = [None] * 14

        def slot(idx):
            p = ctx.params[idx]
            if p is None or not ctx.needs_input_grad[4 + idx]:
                return None
            s = ops._grad_slot(p)
            if s is None:
                s = grads[idx] = torch.zeros_like(p)
            else:
                ops._note_staged(p)
            return s
        ys, coefs = (y1, y2, y3), (c1, c2, c3)
        names = ('c3', 'c6', 'c9')
        widx = (4, 8, 12)                                           # positions of w2, w3, w4 in params (bias = +1)
        gidx = (2, 6, 10)                                           # gamma of BatchNorm 1..3 (beta = +1)
        dz, yy, cabc = g, None, None                               # the operands of dy of the layer being processed
        for k in ((2, 1) if ctx.use12 else (2, 1, 0)):
            ci, co = _WAVE_LAYERS[k]
            Lin, Lout = lens[k], lens[k + 1]
            g_f32 = k == 2
            yp, cp = ys[k], coefs[k]
            wslot, bslot = slot(widx[k]), slot(widx[k] + 1)
            ca, cb, cc = (None, None, None) if g_f32 else (cabc[0], cabc[1], cabc[2])
            if wslot is not None:
                nb = lib.s2ag_wave_wgrad_blocks(N, Lout, ci, co)
                part = torch.empty(nb * co * 15 * ci + nb * co, dtype=torch.float32, device=dev)

                def launch(dz=dz, yy=yy, ca=ca, cb=cb, cc=cc, g_f32=g_f32, yp=yp, cp=cp, part=part, nb=nb, wslot=wslot,
                           bslot=bslot, Lin=Lin, Lout=Lout, ci=ci, co=co):
                    L.check(lib.s2ag_wave_conv_wgrad(_p(dz), _p(yy), _p(ca), _p(cb), _p(cc), int(g_f32), _p(yp), _p(cp[0]),
                                                     _p(cp[1]), slope, _p(part), _p(part[nb * co * 15 * ci:]), _p(wslot),
                                                     _p(bslot), N, Lin, Lout, ci, co, _s()), 'wave_conv_wgrad')
                # a leaf of the backward graph: the two big ones run beside the data-gradient chain when the trainer armed
                # the weight-gradient stream (ops.run_wgrad); the trainer joins before the optimizer
                ops.run_wgrad(launch, keep=(dz, yy, yp, cp, part, cabc), flops=2.0 * N * Lout * co * ci * 15)
            wph = pack.get(names[k], 'phases')                      # (6, ci, 3, CPO)
            prow = lib.s2ag_wave_dgrad_rows(N, Lin, ci)
            st = torch.empty(2 * (prow + (prow + 15) // 16) * ci, dtype=torch.float64, device=dev)
            dzp = torch.empty(N, Lin, ci, dtype=torch.bfloat16, device=dev)
            cabc = torch.empty(3, ci, dtype=torch.float32, device=dev)
            # the workgroup that finishes last folds the partial sums: gamma / beta gradients + the coefficients of dy_{k}
            L.check(lib.s2ag_wave_conv_dgrad(_p(dz), _p(yy), _p(ca), _p(cb), _p(cc), int(g_f32), _p(wph), int(wph.shape[3]),
                                             _p(yp), _p(cp[0]), _p(cp[1]), _p(cp[2]), _p(cp[3]), slope, _p(dzp), _p(st),
                                             ops._tickets(dev, 1 + (prow + 15) // 16), _p(ctx.params[gidx[k]]), _p(slot(gidx[k])),
                                             _p(slot(gidx[k] + 1)),
                                             _p(cabc[0]), _p(cabc[1]), _p(cabc[2]), N, Lin, Lout, ci, co, _s()),
                    'wave_conv_dgrad')
            dz, yy = dzp, yp
        if ctx.use12:
            # conv2's two gradients, BatchNorm 1's backward and conv1's weight gradient: ONE launch over dy2 and the waveform
            from . import wave12
            slots = {'w1': slot(0), 'g1': slot(2), 'e1': slot(3), 'w2': slot(4)}
            for i, b in ((1, b1), (5, b2)):                         # exactly zero: these biases feed a BatchNorm
                if b is not None and ctx.needs_input_grad[4 + i] and ops._grad_slot(b) is None:
                    grads[i] = torch.zeros_like(b)
            wave12.backward(wav, y1, b1, c1, g1, slope, dz, yy, cabc, slots, ctx.pad1)
            return (None, None, None, None) + tuple(grads)
        wslot, bslot = slot(0), slot(1)
        if wslot is not None:
            geom1 = L.ConvGeom(N, wav.shape[1], lens[0], 1, 16, 15, 5, ctx.pad1, 1, 1, 16, 0)
            part = torch.empty(lib.s2ag_wave_conv1_wgrad_blocks(C.byref(geom1)) * 256, dtype=torch.float32, device=dev)
            L.check(lib.s2ag_wave_conv1_wgrad(_p(dz), _p(yy), _p(cabc[0]), _p(cabc[1]), _p(cabc[2]), _p(wav), _p(part), _p(wslot),
                                              _p(bslot), C.byref(geom1), _s()), 'wave_conv1_wgrad')
        return (None, None, None, None) + tuple(grads)


def wave_encoder_fused(wav: Tensor, fe, pack: WeightPack) -> Tensor:
    """Training-mode WavEncoder on ``fe`` = its feat_extractor; ``pack`` holds the bf16 layouts of fe[3], fe[6], fe[9]."""
    return _WaveFused16.apply(wav, pack, (fe[1], fe[4], fe[7]), float(fe[2].negative_slope),
                              fe[0].weight, fe[0].bias, fe[1].weight, fe[1].bias, fe[3].weight, fe[3].bias, fe[4].weight,
                              fe[4].bias, fe[6].weight, fe[6].bias, fe[7].weight, fe[7].bias, fe[9].weight, fe[9].bias)
