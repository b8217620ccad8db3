"""Head of the wave encoder -- feat_extractor[0..3] of WavEncoder (net/multimodal_context_net_v2.py:18-21 of the
reference: Conv1d(1,16,15,s5,p1600) BatchNorm1d(16) LeakyReLU(0.3) Conv1d(16,32,15,s6)) -- without conv1's output in HBM
(csrc/wave12.hip): statistics pass, forward, and ONE backward launch that recomputes conv1 from the waveform.

Two users: the fused bf16 encoder (bf16._WaveFused16: z2 in bf16, BatchNorm 2 folded by the producing launch) and the
fp32 mode (``head_f32``: z2 in fp32 with its column-sum partials for ops.batch_norm_act, operands as two bf16 pieces)."""
import ctypes as C
import os
from typing import Optional

import torch
from torch import Tensor

from . import _lib as L
from . import config
from . import ops

ENABLED = config.mirror('WAVE12', globals(), 'ENABLED')
PAD1 = 1600


def _lib():
    return L.load()


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def lengths(n_samples: int, pad: int = PAD1):
    l1 = (n_samples + 2 * pad - 15) // 5 + 1
    return l1, (l1 - 15) // 6 + 1


def bn_foldable(bn) -> bool:
    """A BatchNorm the fused launches can fold: affine, with running estimates and a NUMERIC momentum (momentum=None --
    cumulative averaging -- and track_running_stats=False take the layer-by-layer path)."""
    return (isinstance(bn, torch.nn.modules.batchnorm._BatchNorm) and bn.affine and bn.track_running_stats
            and bn.running_mean is not None and bn.running_var is not None and isinstance(bn.momentum, (int, float)))


def leaky_slope(act) -> Optional[float]:
    return float(act.negative_slope) if isinstance(act, torch.nn.LeakyReLU) else None


def supported(fe) -> bool:
    """``fe`` = WavEncoder.feat_extractor: the geometry (and the BatchNorm / activation flavour) the kernels are written for."""
    if not ENABLED:
        return False
    c1, c2 = fe[0], fe[3]
    return ((c1.in_channels, c1.out_channels, c1.kernel_size[0], c1.stride[0], c1.dilation[0]) == (1, 16, 15, 5, 1)
            and (c2.in_channels, c2.out_channels, c2.kernel_size[0], c2.stride[0], c2.padding[0], c2.dilation[0]) ==
            (16, 32, 15, 6, 0, 1) and c1.bias is not None and bn_foldable(fe[1]) and leaky_slope(fe[2]) is not None)


def packed_weights(w1: Tensor, w2: Tensor) -> Tensor:
    """bf16 operand layouts (two pieces each) of conv1 / conv2, refreshed when a weight changes or a new step begins
    (same keying as bf16.WeightPack); cached on w2."""
    key = tuple((id(w), w._version, w.data_ptr()) for w in (w1, w2))
    if w1.requires_grad or w2.requires_grad:                 # derived tensors of trainable weights never cross a step boundary
        key += (ops.generation(),)
    ent = getattr(w2, '_s2ag_w12', None)
    if ent is None or ent[0] != key:
        lib = _lib()
        with torch.no_grad():
            out = torch.empty(lib.s2ag_wave12_pack_elems(), dtype=torch.bfloat16, device=w1.device)
            L.check(lib.s2ag_wave12_pack(_p(w1.detach().contiguous()), _p(w2.detach().contiguous()), _p(out), _s()),
                    'wave12_pack')
        ent = (key, out)
        w2._s2ag_w12 = ent
    return ent[1]


def _check_wav(wav: Tensor) -> None:
    """The kernels read (clips, samples) fp32 rows; there is no CPU path (ops.py raises likewise)."""
    if not (wav.is_cuda and wav.dtype == torch.float32 and wav.dim() == 2 and wav.is_contiguous()):
        raise RuntimeError('wave12: the waveform must be a contiguous (clips, samples) float32 CUDA tensor, got '
                           f'{tuple(wav.shape)} {wav.dtype} on {wav.device}')


def fold_args(bn, gamma, beta, prow: int, dev, keep: list):
    """BatchNorm fold done by the producing launch's last workgroup: -> (coef (4, C): scale, shift, mean, invstd; args)"""
    coef = torch.empty(4, gamma.numel(), dtype=torch.float32, device=dev)
    fa = L.BnFoldArgs(ops._tickets(dev, 1 + (prow + 15) // 16), _p(gamma), _p(beta), _p(bn.running_mean), _p(bn.running_var),
                      _p(bn.num_batches_tracked), float(bn.eps), float(bn.momentum), int(ops._BN_REPEAT[0]),
                      _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]))
    keep.append(fa)
    return coef, fa


def stats(wav: Tensor, pk: Tensor, b1: Tensor, bn1, gamma1: Tensor, beta1: Tensor, round_bf16: bool, pad: int = PAD1) -> Tensor:
    """Batch statistics of conv1's output -> BatchNorm 1's running estimates and coefficients (4, 16)."""
    lib = _lib()
    _check_wav(wav)
    N, Lin = wav.shape
    L1, _ = lengths(Lin, pad)
    prow = lib.s2ag_wave12_stats_rows(N, L1)
    part = torch.empty(2 * (prow + (prow + 15) // 16) * 16, dtype=torch.float64, device=wav.device)
    keep = []
    coef, fa = fold_args(bn1, gamma1, beta1, prow, wav.device, keep)
    L.check(lib.s2ag_wave12_stats(_p(wav), _p(pk), _p(b1), _p(part), C.byref(fa), int(round_bf16), N, Lin, L1, pad, _s()),
            'wave12_stats')
    return coef


def forward(wav: Tensor, pk: Tensor, b1: Tensor, coef1: Tensor, slope: float, b2: Optional[Tensor], out_f32: bool,
            fold=None, pad: int = PAD1):
    """-> (z2 (N, L2, 32), partial column sums (2, rows (+ groups), 32) fp64, rows).  ``fold`` = (bn2, gamma2, beta2):
    BatchNorm 2's coefficients from the same launch (returned as a 4th element)."""
    lib = _lib()
    _check_wav(wav)
    N, Lin = wav.shape
    L1, L2 = lengths(Lin, pad)
    dev = wav.device
    z2 = torch.empty(N, L2, 32, dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
    prow = lib.s2ag_wave12_fwd_rows(N, L2)
    part = torch.empty(2 * (prow + (prow + 15) // 16) * 32, dtype=torch.float64, device=dev)
    keep, coef2, fa = [], None, None
    if fold is not None:
        coef2, fa = fold_args(fold[0], fold[1], fold[2], prow, dev, keep)
    L.check(lib.s2ag_wave12_fwd(_p(wav), _p(pk), _p(b1), _p(coef1[0]), _p(coef1[1]), float(slope), _p(b2), _p(z2), int(out_f32),
                                _p(part), C.byref(fa) if fa is not None else None, N, Lin, L1, L2, pad, _s()), 'wave12_fwd')
    return z2, part, prow, coef2


def backward(wav: Tensor, pk: Tensor, b1: Tensor, coef1: Tensor, gamma1: Tensor, slope: float, dz: Tensor, z2: Optional[Tensor],
             cabc2: Optional[Tensor], slots: dict, pad: int = PAD1) -> Tensor:
    """One launch (+ the fold of its partial tiles) for everything behind z2.  ``dz``: fp32 gradient w.r.t. z2, or -- with
    ``z2`` / ``cabc2`` (3, 32) -- the bf16 operands of dy2 = ca dz + cc z2 + cb.  ``slots``: accumulation targets
    {'w1', 'g1', 'e1', 'w2'} -> fp32 tensor or None (the two biases feed BatchNorms: zero gradient).  Returns ca / cb / cc of BatchNorm 1's backward (3, 16)."""
    lib = _lib()
    N, Lin = wav.shape
    L1, L2 = lengths(Lin, pad)
    dev = wav.device
    f32 = z2 is None
    nb = lib.s2ag_wave12_bwd_blocks(N, L1, int(f32))
    ng = (nb + 15) // 16
    scratch = torch.empty(nb * 7680 + (nb + ng) * 528, dtype=torch.float32, device=dev)
    st = torch.empty(2 * (nb + ng) * 16, dtype=torch.float64, device=dev)
    cabc1 = torch.empty(3, 16, dtype=torch.float32, device=dev)
    o1 = nb * 7680
    a = L.Wave12Bwd(_p(wav), _p(pk), _p(b1), _p(coef1[0]), _p(coef1[1]), _p(coef1[2]), _p(coef1[3]), _p(gamma1), float(slope),
                    _p(dz), int(f32), _p(z2), *((None, None, None) if f32 else (_p(cabc2[0]), _p(cabc2[1]), _p(cabc2[2]))),
                    _p(scratch), _p(scratch[o1:]), _p(st), ops._tickets(dev, 1 + ng),
                    _p(slots.get('g1')), _p(slots.get('e1')), _p(cabc1[0]), _p(cabc1[1]), _p(cabc1[2]), _p(slots.get('w2')),
                    _p(slots.get('w1')), N, Lin, L1, L2, pad)
    L.check(lib.s2ag_wave12_bwd(C.byref(a), _s()), 'wave12_bwd')
    return cabc1


def act_signs(wav: Tensor, pk: Tensor, b1: Tensor, coef1: Tensor, round_bf16: bool, pad: int = PAD1) -> Tensor:
    """(N, L1, 16) bool: the branch the LeakyReLU behind BatchNorm 1 takes inside the fused launches (parity tests)."""
    N, Lin = wav.shape
    L1, _ = lengths(Lin, pad)
    out = torch.empty(N, L1, 16, dtype=torch.uint8, device=wav.device)
    L.check(_lib().s2ag_wave12_act_signs(_p(wav), _p(pk), _p(b1), _p(coef1[0]), _p(coef1[1]), int(round_bf16), _p(out), N, Lin, L1,
                                         pad, _s()), 'wave12_act_signs')
    return out.bool()


class _HeadF32(torch.autograd.Function):
    """fp32 mode: (N, samples) waveform -> z2 (N, L2, 32) fp32, the raw output of conv2 (BatchNorm 2 follows outside)."""

    @staticmethod
    def forward(ctx, wav, bn1, slope_pad, w1, b1, g1, e1, w2, b2):
        slope, pad, box = slope_pad
        wav = wav.contiguous()
        pk = packed_weights(w1, w2)
        coef1 = stats(wav, pk, b1, bn1, g1, e1, False, pad)
        z2, part, prow, _ = forward(wav, pk, b1, coef1, slope, b2, True, pad=pad)
        box['stats'] = (part, prow)          # handed back through the caller's own object (no class-level state)
        ctx.slope, ctx.pad, ctx.params = float(slope), int(pad), (w1, b1, g1, e1, w2, b2)
        ctx.save_for_backward(wav, pk, coef1)
        return z2

    @staticmethod
    def backward(ctx, dz2):
        wav, pk, coef1 = ctx.saved_tensors
        w1, b1, g1, e1, w2, b2 = ctx.params
        names = ('w1', 'b1', 'g1', 'e1', 'w2', 'b2')
        grads = [None] * 6
        slots = {}
        for i, (nm, p) in enumerate(zip(names, ctx.params)):
            if p is None or not ctx.needs_input_grad[3 + i]:
                continue
            if nm in ('b1', 'b2'):              # identically zero: both biases feed a BatchNorm
                if ops._grad_slot(p) is None:
                    grads[i] = torch.zeros_like(p)
                continue
            s = ops._grad_slot(p)
            if s is None:
                s = grads[i] = torch.zeros_like(p)
            else:
                ops._note_staged(p)
            slots[nm] = s
        backward(wav, pk, b1, coef1, g1, ctx.slope, dz2.contiguous().float(), None, None, slots, ctx.pad)
        return (None, None, None) + tuple(grads)


def head_f32(wav: Tensor, fe) -> Tensor:
    """Training-mode feat_extractor[0..3] in fp32 mode; the result carries the column-sum partials BatchNorm 2 folds
    (``_s2ag_stats``, as ops.conv1d_nlc(bn_stats=True) leaves them)."""
    box = {}
    z2 = _HeadF32.apply(wav, fe[1], (leaky_slope(fe[2]), int(fe[0].padding[0]), box), fe[0].weight, fe[0].bias, fe[1].weight,
                        fe[1].bias, fe[3].weight, fe[3].bias)
    z2._s2ag_stats = box['stats']
    return z2
