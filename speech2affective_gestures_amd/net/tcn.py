"""Temporal convolutional network.  Mirrors ``net.tcn`` of the reference (net/tcn.py:7-64): Chomp1d,
TemporalBlock, TemporalConvNet with identical constructors and ``state_dict`` keys (including the
``net.0`` / ``net.4`` aliases of ``conv1`` / ``conv2`` and the weight_g / weight_v split of weight_norm).

MI355X formulation: activations stay channels-last (B, T, C); the causal dilated conv + chomp is ONE
implicit GEMM whose two taps read frames t-d and t (left padding only, output length T), with bias,
ReLU and the counter-based dropout fused in the epilogue.  The torch modules below are parameter
containers; their own ``forward`` is never used.
"""
import torch
import torch.nn as nn
from torch.nn.utils import weight_norm

from .. import ops
from .._lib import ACT_LEAKY
from ..noise import new_site, noise_pass


class Chomp1d(nn.Module):
    def __init__(self, chomp_size):
        super().__init__()
        self.chomp_size = chomp_size

    def forward(self, x):
        return x[:, :, :-self.chomp_size].contiguous()


class TemporalBlock(nn.Module):
    def __init__(self, n_inputs, n_outputs, kernel_size, stride, dilation, padding, dropout=0.2):
        super().__init__()
        if stride != 1 or padding != (kernel_size - 1) * dilation:
            raise NotImplementedError('causal TCN block: stride 1, padding (k-1)*dilation (net/tcn.py:57-59)')
        self.dilation, self.kernel_size, self.p = dilation, kernel_size, dropout
        self.conv1 = weight_norm(nn.Conv1d(n_inputs, n_outputs, kernel_size, stride=stride, padding=padding,
                                           dilation=dilation))
        self.chomp1 = Chomp1d(padding)
        self.relu1 = nn.ReLU()
        self.dropout1 = nn.Dropout(dropout)
        self.conv2 = weight_norm(nn.Conv1d(n_outputs, n_outputs, kernel_size, stride=stride, padding=padding,
                                           dilation=dilation))
        self.chomp2 = Chomp1d(padding)
        self.relu2 = nn.ReLU()
        self.dropout2 = nn.Dropout(dropout)
        self.net = nn.Sequential(self.conv1, self.chomp1, self.relu1, self.dropout1,
                                 self.conv2, self.chomp2, self.relu2, self.dropout2)
        self.downsample = nn.Conv1d(n_inputs, n_outputs, 1) if n_inputs != n_outputs else None
        self.relu = nn.ReLU()
        self.sites = (new_site(), new_site())

    def forward_nlc(self, x, noise, weights=None):
        """x (B, T, C) channels-last.  ``weights``: the two normalised (Cout, k, Cin) weights if the caller already
        has them (TemporalConvNet computes all of its blocks' in one launch per optimizer step)."""
        out = x
        pad = (self.kernel_size - 1) * self.dilation
        for i, (conv, site) in enumerate(((self.conv1, self.sites[0]), (self.conv2, self.sites[1]))):
            w = weights[i] if weights is not None else \
                ops.weight_norm(conv.weight_v, conv.weight_g, tap_major=True)          # (Cout, k, Cin)
            p = self.p if self.training else 0.0
            out = ops.conv1d_nlc(out, w, conv.bias, pad=pad, dil=self.dilation, lout=x.shape[1], act=ACT_LEAKY,
                                 slope=0.0, drop_p=p, noise=noise, site=site, w_tap_major=True)
        res = x if self.downsample is None else ops.conv1d_nlc(x, self.downsample.weight, self.downsample.bias)
        return ops.add_act(out, res, 0.0)

    def forward_nlc_bf16(self, x, noise, weights, pack, idx):
        """bf16 mode: x (B, T, Cp) bf16 with zero pad channels; ``weights`` = the two normalised fp32 (Cout, k, Cin)
        tensors (gradient stages), ``pack`` their bf16 layouts under the names c{idx}a / c{idx}b."""
        from .. import bf16
        out = x
        pad = (self.kernel_size - 1) * self.dilation
        C = self.conv1.out_channels
        p = self.p if self.training else 0.0
        for j, (conv, site) in enumerate(((self.conv1, self.sites[0]), (self.conv2, self.sites[1]))):
            out = bf16.conv(out, weights[j], conv.bias, pack, f'c{idx}{"ab"[j]}', conv.in_channels, conv.out_channels,
                            self.kernel_size, pad=pad, dil=self.dilation, lout=x.shape[1], act=ACT_LEAKY, slope=0.0,
                            drop_p=p, noise=noise, site=site, pad_out=True)
        return bf16.add_act(out, x, 0.0, C)

    def forward(self, x):
        """reference layout (B, C, T)"""
        with noise_pass(x.device) as nz:
            return self.forward_nlc(x.transpose(1, 2).contiguous(), nz).transpose(1, 2).contiguous()


class TemporalConvNet(nn.Module):
    def __init__(self, num_inputs, num_channels, kernel_size=2, dropout=0.2):
        super().__init__()
        layers = []
        for i, out_channels in enumerate(num_channels):
            d = 2 ** i
            in_channels = num_inputs if i == 0 else num_channels[i - 1]
            layers.append(TemporalBlock(in_channels, out_channels, kernel_size, stride=1, dilation=d,
                                        padding=(kernel_size - 1) * d, dropout=dropout))
        self.network = nn.Sequential(*layers)
        self._groups = None

    def _weight_groups(self):
        """ops.WeightNormGroup per <= 8 convs: w = g v/||v|| of every block, recomputed once per optimizer step."""
        if self._groups is None:
            convs = [c for blk in self.network for c in (blk.conv1, blk.conv2)]
            n = ops.L.MAX_JOBS
            self._groups = [ops.WeightNormGroup([c.weight_v for c in convs[i:i + n]],
                                                [c.weight_g for c in convs[i:i + n]]) for i in range(0, len(convs), n)]
        return self._groups

    def _fused32_ok(self, x, blks):
        return (x.dim() == 3 and all(b.kernel_size == 2 and b.p == blks[0].p and b.downsample is None for b in blks)
                and ops.tcn_fused32_supported(x.shape[1], blks[0].conv1.out_channels, 2, len(blks))
                and x.shape[2] == blks[0].conv1.in_channels == blks[0].conv1.out_channels)

    def forward_nlc(self, x, noise, batch=None, noises=None):
        """``batch`` / ``noises``: x is the first pass of a lockstep batch (ops.tcn_fused32); then (out, mate outputs)."""
        ws = [w for g in self._weight_groups() for w in g.tensors()]
        blks = list(self.network)
        if self._fused32_ok(x, blks):
            # clip-resident forward (csrc/tcn_fused32.hip): every block in ONE launch; backward layer by layer
            if self.__dict__.get('_frag32') is None:
                self.__dict__['_frag32'] = ops.TcnFragments32()
            p = blks[0].p if self.training else 0.0
            return ops.tcn_fused32(x, self.__dict__['_frag32'], ws, [c.bias for b in blks for c in (b.conv1, b.conv2)],
                                   [b.dilation for b in blks], [s for b in blks for s in b.sites], p, noise,
                                   batch=batch, noises=noises)
        assert batch is None, 'lockstep batches need the clip-resident kernel (check lockstep_capable first)'
        for i, blk in enumerate(blks):
            x = blk.forward_nlc(x, noise, weights=ws[2 * i:2 * i + 2])
        return x

    def lockstep_capable(self, T, C):
        blks = list(self.network)
        return self._fused32_ok(torch.empty(0, T, C), blks)

    def bf16_capable(self):
        """The bf16 path covers the shape the S2AG text encoder uses: no down-sampling residual (in == out channels)."""
        return all(blk.downsample is None for blk in self.network)

    def forward_nlc_bf16(self, x, noise, decoder=None):
        """bf16 mode (bf16.py): x (B, T, Cp) bf16.  The weight-normed fp32 weights stay the gradient stages; their bf16
        forward / data-gradient layouts (and the decoder Linear's) are refreshed with one launch per optimizer step.
        Returns bf16 (B, T, Cp), or the decoder's fp32 (B, T, out) when ``decoder`` is given."""
        from .. import bf16
        ws = [w for g in self._weight_groups() for w in g.tensors()]
        blks = list(self.network)
        fused = (x.shape[-1] == 320 and all(b.kernel_size == 2 and b.p == blks[0].p for b in blks)
                 and bf16.tcn_fused_supported(x.shape[1], blks[0].conv1.out_channels, 2, len(blks)))
        if (self.__dict__.get('_pack16') is None or self.__dict__.get('_pack16_dec') is not decoder
                or self.__dict__.get('_pack16_fused') != fused):
            pk = bf16.WeightPack()
            for i, blk in enumerate(self.network):
                if fused:
                    break           # the clip-resident kernels take their weights in fragment order (bf16.TcnFragments)
                for j, conv in enumerate((blk.conv1, blk.conv2)):
                    pk.add(f'c{i}{"ab"[j]}', (lambda k=2 * i + j: self.__dict__['_cur_ws'][k]), 'tap_major', conv.out_channels,
                           conv.in_channels, blk.kernel_size)
            if decoder is not None:
                pk.add('dec', (lambda d=decoder: d.weight), 'tap_major', decoder.out_features, decoder.in_features, 1)
            # plain attributes (NOT nn.Module attributes: the decoder must not become a sub-module of the TCN)
            self.__dict__['_pack16'], self.__dict__['_pack16_dec'], self.__dict__['_pack16_fused'] = pk, decoder, fused
        self.__dict__['_cur_ws'] = ws
        p = blks[0].p if self.training else 0.0
        if fused:
            # clip-resident path (csrc/tcn_fused.hip): every block in one launch, forward and backward
            if self.__dict__.get('_frag16') is None:
                self.__dict__['_frag16'] = bf16.TcnFragments()
            x = bf16.tcn_fused(x, self.__dict__['_frag16'], ws, [c.bias for b in blks for c in (b.conv1, b.conv2)],
                               [b.dilation for b in blks], [s for b in blks for s in b.sites], p, noise)
        else:
            for i, blk in enumerate(blks):
                x = blk.forward_nlc_bf16(x, noise, ws[2 * i:2 * i + 2], self.__dict__['_pack16'], i)
        if decoder is not None:
            return bf16.conv(x, decoder.weight, decoder.bias, self.__dict__['_pack16'], 'dec', decoder.in_features,
                             decoder.out_features, 1, out_f32=True)
        return x

    def forward(self, x):
        with noise_pass(x.device) as nz:
            return self.forward_nlc(x.transpose(1, 2).contiguous(), nz).transpose(1, 2).contiguous()
