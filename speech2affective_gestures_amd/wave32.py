"""Training-mode WavEncoder of the fp32 mode with ALL THREE BatchNorms folded into the neighbouring convs -- the fp32
counterpart of bf16._WaveFused16 (net/multimodal_context_net_v2.py:14-33 of the reference).

  head   conv1 -> BatchNorm 1 -> LeakyReLU -> conv2 from the waveform (wave12.py / csrc/wave12.hip), BatchNorm 2's fold in
         the same launch
  tail   conv3 / conv4 on fp32 rows (csrc/wave_fused.hip, the *32 entry points): a conv applies the previous BatchNorm +
         LeakyReLU in its loader and leaves its own column sums behind (folded by its last workgroup); backward, a data
         gradient emits dz = da * leaky'(.) + the BatchNorm backward sums, and the consumers of dy = A dz + C y + B form it
         in their loaders.  Forward products on the f32 matrix pipe (fp32 arithmetic); the gradients take their operands as
         two bf16 pieces (the fp32 mode's treatment of every large gradient).

Against the layer-by-layer fp32 path this removes two BatchNorm apply passes forward, two statistics + two apply passes
backward and every a_i tensor.  STATE: written and cross-compiled without access to a GPU; NOT yet run.  It is therefore off
by default (``S2AG_WAVE_TAIL32=1`` routes WavEncoder.forward through it; tests/test_gpu_wave32.py checks it against the
layer-by-layer path and skips unless the variable is set)."""
import ctypes as C
import os

import torch
from torch import Tensor

from . import _lib as L
from . import config
from . import ops
from . import wave12

ENABLED = config.mirror('WAVE_TAIL32', globals(), 'ENABLED')
_LAYERS = ((32, 64), (64, 32))                    # (Cin, Cout) of conv3, conv4: 15 taps, stride 6, no padding


def _lib():
    return L.load()


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def supported(fe) -> bool:
    """``fe`` = WavEncoder.feat_extractor: the geometry the kernels are written for (and the head's)."""
    if not (ENABLED and wave12.supported(fe)):
        return False
    ok = True
    for i, (ci, co) in zip((6, 9), _LAYERS):
        c = fe[i]
        ok = ok and (c.in_channels, c.out_channels, c.kernel_size[0], c.stride[0], c.padding[0], c.dilation[0]) == \
            (ci, co, 15, 6, 0, 1)
    slopes = [wave12.leaky_slope(fe[i]) for i in (2, 5, 8)]
    return ok and all(wave12.bn_foldable(fe[i]) for i in (4, 7)) and slopes == [slopes[0]] * 3


def tail_lengths(l2: int):
    l3 = (l2 - 15) // 6 + 1
    return l3, (l3 - 15) // 6 + 1


def pack_views(buf: Tensor):
    """-> ((k3, p3), (k4, p4)): byte-offset views into the pack (k-major fp32 of the forward, two-piece phase form)"""
    lib = _lib()
    out = []
    for layer in (0, 1):
        ok, op = lib.s2ag_wave_tail32_pack_offset(layer, 0), lib.s2ag_wave_tail32_pack_offset(layer, 1)
        out.append((buf[ok:], buf[op:]))
    return tuple(out)


def packed_tail(w3: Tensor, w4: Tensor) -> Tensor:
    """Operand layouts of conv3 / conv4 (one launch), refreshed when a weight changes or a new step begins; cached on w4."""
    key = tuple((id(w), w._version, w.data_ptr()) for w in (w3, w4))
    if w3.requires_grad or w4.requires_grad:
        key += (ops.generation(),)
    ent = getattr(w4, '_s2ag_t32', None)
    if ent is None or ent[0] != key:
        lib = _lib()
        with torch.no_grad():
            out = torch.empty(int(lib.s2ag_wave_tail32_pack_bytes()), dtype=torch.uint8, device=w3.device)
            L.check(lib.s2ag_wave_tail32_pack(_p(w3.detach().contiguous()), _p(w4.detach().contiguous()), _p(out), _s()),
                    'wave_tail32_pack')
        ent = (key, out)
        w4._s2ag_t32 = ent
    return ent[1]


def conv_fwd32(x: Tensor, coef: Tensor, slope: float, wk: Tensor, bias, ci: int, co: int, fold=None):
    """y = conv(leaky(coef[0] x + coef[1])) + bias on fp32 rows (N, Lin, ci) -> (N, Lout, co).  ``fold`` = (bn, gamma, beta):
    also the coefficients (4, co) of the BatchNorm behind y, folded in the same launch.  -> (y, coef_out | None)"""
    lib = _lib()
    N, Lin, _ = x.shape
    Lout = (Lin - 15) // 6 + 1
    y = torch.empty(N, Lout, co, dtype=torch.float32, device=x.device)
    st, fa, coef_out, keep = None, None, None, []
    if fold is not None:
        prow = lib.s2ag_wave_fwd_rows(N, Lout, ci, co)
        st = torch.empty(2 * (prow + (prow + 15) // 16) * co, dtype=torch.float64, device=x.device)
        coef_out, fa = wave12.fold_args(fold[0], fold[1], fold[2], prow, x.device, keep)
    L.check(lib.s2ag_wave_conv_fwd32(_p(x), _p(coef[0]), _p(coef[1]), float(slope), _p(wk), _p(bias), _p(y), _p(st),
                                     C.byref(fa) if fa is not None else None, N, Lin, Lout, ci, co, _s()), 'wave_conv_fwd32')
    return y, coef_out


class _WaveFused32(torch.autograd.Function):
    """(N, samples) fp32 -> (N, frames, 32) fp32; see the module docstring."""

    @staticmethod
    def forward(ctx, wav, bns, slope_pad, *params):
        w1, b1, g1, e1, w2, b2, g2, e2, w3, b3, g3, e3, w4, b4 = params
        slope, pad = slope_pad
        wav = wav.contiguous()
        pk12 = wave12.packed_weights(w1, w2)
        coef1 = wave12.stats(wav, pk12, b1, bns[0], g1, e1, False, pad)
        z2, _, _, coef2 = wave12.forward(wav, pk12, b1, coef1, slope, b2, True, fold=(bns[1], g2, e2), pad=pad)
        pkt = packed_tail(w3, w4)
        (k3, _), (k4, _) = pack_views(pkt)
        z3, coef3 = conv_fwd32(z2, coef2, slope, k3, b3, 32, 64, fold=(bns[2], g3, e3))
        out, _ = conv_fwd32(z3, coef3, slope, k4, b4, 64, 32)
        ctx.slope, ctx.pad, ctx.params = float(slope), int(pad), params
        ctx.save_for_backward(wav, pk12, pkt, z2, z3, coef1, coef2, coef3)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        wav, pk12, pkt, z2, z3, coef1, coef2, coef3 = ctx.saved_tensors
        w1, b1, g1, e1, w2, b2, g2, e2, w3, b3, g3, e3, w4, b4 = ctx.params
        slope = ctx.slope
        N, dev = wav.shape[0], wav.device
        g = g.contiguous().float()
        grads = [None] * 14

        def slot(idx):
            p = ctx.params[idx]
            if p is None or not ctx.needs_input_grad[3 + idx]:
                return None
            s = ops._grad_slot(p)
            if s is None:
                s = grads[idx] = torch.zeros_like(p)
            else:
                ops._note_staged(p)
            return s
        (_, p3), (_, p4) = pack_views(pkt)
        # (layer, dz, y, cabc, previous raw output, its BatchNorm's coefficients, phase weights, slots of w / b / gamma / beta)
        dz, yy, cabc = g, None, None
        for (ci, co), yp, cp, wph, wi, gi, gamma in (((64, 32), z3, coef3, p4, 12, 10, g3), ((32, 64), z2, coef2, p3, 8, 6, g2)):
            Lin, Lout = yp.shape[1], dz.shape[1]
            is_dy = cabc is None
            ca, cb, cc = (None, None, None) if is_dy else (cabc[0], cabc[1], cabc[2])
            wslot, bslot = slot(wi), slot(wi + 1)
            if wslot is not None:
                nb = lib.s2ag_wave_wgrad_blocks(N, Lout, ci, co)
                part = torch.empty(nb * co * 15 * ci + nb * co, dtype=torch.float32, device=dev)

                def launch(dz=dz, yy=yy, ca=ca, cb=cb, cc=cc, is_dy=is_dy, yp=yp, cp=cp, part=part, nb=nb, wslot=wslot,
                           bslot=bslot, Lin=Lin, Lout=Lout, ci=ci, co=co):
                    L.check(lib.s2ag_wave_conv_wgrad32(_p(dz), _p(yy), _p(ca), _p(cb), _p(cc), int(is_dy), _p(yp), _p(cp[0]),
                                                       _p(cp[1]), slope, _p(part), _p(part[nb * co * 15 * ci:]), _p(wslot),
                                                       _p(bslot), N, Lin, Lout, ci, co, _s()), 'wave_conv_wgrad32')
                ops.run_wgrad(launch, keep=(dz, yy, yp, cp, part, cabc), flops=2.0 * N * Lout * co * ci * 15)
            prow = lib.s2ag_wave_dgrad_rows(N, Lin, ci)
            st = torch.empty(2 * (prow + (prow + 15) // 16) * ci, dtype=torch.float64, device=dev)
            dzp = torch.empty(N, Lin, ci, dtype=torch.float32, device=dev)
            cabc_p = torch.empty(3, ci, dtype=torch.float32, device=dev)
            L.check(lib.s2ag_wave_conv_dgrad32(_p(dz), _p(yy), _p(ca), _p(cb), _p(cc), int(is_dy), _p(wph), _p(yp), _p(cp[0]),
                                               _p(cp[1]), _p(cp[2]), _p(cp[3]), slope, _p(dzp), _p(st),
                                               ops._tickets(dev, 1 + (prow + 15) // 16), _p(gamma), _p(slot(gi)), _p(slot(gi + 1)),
                                               _p(cabc_p[0]), _p(cabc_p[1]), _p(cabc_p[2]), N, Lin, Lout, ci, co, _s()),
                    'wave_conv_dgrad32')
            dz, yy, cabc = dzp, yp, cabc_p
        # conv2's two gradients, BatchNorm 1's backward and conv1's weight gradient: one launch over dy2's operands + the waveform
        slots = {'w1': slot(0), 'g1': slot(2), 'e1': slot(3), 'w2': slot(4)}
        for i, b in ((1, b1), (5, b2)):                             # exactly zero: these biases feed a BatchNorm
            if b is not None and ctx.needs_input_grad[3 + i] and ops._grad_slot(b) is None:
                grads[i] = torch.zeros_like(b)
        wave12.backward(wav, pk12, b1, coef1, g1, slope, dz, yy, cabc, slots, ctx.pad)
        return (None, None, None) + tuple(grads)


def encoder_f32(wav: Tensor, fe) -> Tensor:
    """Training-mode WavEncoder.forward on ``fe`` = its feat_extractor (fp32 mode)."""
    return _WaveFused32.apply(wav, (fe[1], fe[4], fe[7]), (float(fe[2].negative_slope), int(fe[0].padding[0])),
                              fe[0].weight, fe[0].bias, fe[1].weight, fe[1].bias, fe[3].weight, fe[3].bias, fe[4].weight,
                              fe[4].bias, fe[6].weight, fe[6].bias, fe[7].weight, fe[7].bias, fe[9].weight, fe[9].bias)


def act_signs(out: Tensor, fe):
    """Branch decisions of the three LeakyReLUs inside the fused launches of ``out = encoder_f32(wav, fe)`` (parity tests):
    -> [(N, L1, 16), (N, L2, 32), (N, L3, 64)] bool"""
    wav, pk12, _, z2, z3, coef1, coef2, coef3 = out.grad_fn.saved_tensors
    s1 = wave12.act_signs(wav, pk12, fe[0].bias, coef1, False, out.grad_fn.pad)
    return [s1, torch.addcmul(coef2[1], z2, coef2[0]) > 0, torch.addcmul(coef3[1], z3, coef3[0]) > 0]
