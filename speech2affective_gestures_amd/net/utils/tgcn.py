"""Spatial-temporal graph convolution block.  Mirrors ``net.utils.tgcn`` of the reference
(net/utils/tgcn.py:15-71 ConvTemporalGraphical, :133-218 STGraphConv): same constructors, same
``state_dict`` keys (gcn.conv.*, tcn.{0,2,3}.*, residual.{0,1}.*), same ``forward(x, A) -> (y, A)`` on
(N, C, T, V) tensors.

MI355X formulation.  With x kept channels-last as (N, T, V*C) every piece of the block is a 1-D
convolution over T with a dense (V*Cout, V*Cin, kt) weight:
  * gcn conv (kt x 1) followed by einsum('nkctv,kvw->nctw'):  W'[(w,c),dt,(v,ci)] = sum_k W[k*Cout+c,ci,dt] A[k,v,w]
  * tcn conv (kt x kv, zero padded over vertices):             W'[(w,c),(v,ci),dt] = W[c,ci,dt,v-w+kv//2]
  * residual 1x1 conv:                                         W'[(w,c),(v,ci)]    = [v==w] W[c,ci]
so the whole block runs on the one MFMA implicit-GEMM kernel family (ops.conv1d_nlc).  The maps
W -> W' are fixed sparse linear maps; they are applied every step by a CSR kernel (ops.fold) and
their transposes route the gradients back, so the trainable tensors stay in the reference layout.
The caller may choose the column orders of input and output (``in_col``/``out_col``) -- AffEncoder uses
this to make the regrouping of edges into body parts (multimodal_context_net_v2.py:161-167) free.
"""
import numpy as np
import torch
import torch.nn as nn

from ... import ops
from ..._lib import ACT_NONE


def zero(x):
    return 0


def identity(x):
    return x


class ConvTemporalGraphical(nn.Module):
    """Parameter container + reference-layout forward of the graph convolution (tgcn.py:15-71)."""

    def __init__(self, in_channels, out_channels, A_channels, temporal_kernel_size, temporal_stride=1,
                 temporal_padding=0, temporal_dilation=1, bias=True):
        super().__init__()
        if temporal_stride != 1 or temporal_dilation != 1:
            raise NotImplementedError('S2AG hot path uses temporal stride/dilation 1 (tgcn.py:171-174 call site)')
        self.in_channels, self.out_channels, self.A_channels = in_channels, out_channels, A_channels
        self.kt, self.pad = temporal_kernel_size, temporal_padding
        self.conv = nn.Conv2d(in_channels, out_channels * A_channels, kernel_size=(temporal_kernel_size, 1),
                              padding=(temporal_padding, 0), bias=bias)
        self._folds = {}

    def folds(self, A: torch.Tensor, in_col: np.ndarray, out_col: np.ndarray):
        key = (A.data_ptr(), in_col.tobytes(), out_col.tobytes(), str(A.device))
        if key not in self._folds:
            self._folds[key] = _gcn_fold(A.detach().cpu().numpy(), self.in_channels, self.out_channels, self.kt,
                                         in_col, out_col, self.conv.weight.device)
        return self._folds[key]

    def forward_nlc(self, x, A, in_col, out_col):
        wf, bf = self.folds(A, in_col, out_col)
        V = A.shape[1]
        w = ops.fold(self.conv.weight, wf).view(V * self.out_channels, self.kt, V * self.in_channels)
        b = ops.fold(self.conv.bias, bf) if self.conv.bias is not None else None
        return ops.conv1d_nlc(x, w, b, pad=self.pad, w_tap_major=True)

    def forward(self, x, A):
        n, c, t, v = x.shape
        in_col, out_col = default_cols(v, c), default_cols(v, self.out_channels)
        y = self.forward_nlc(x.permute(0, 2, 3, 1).reshape(n, t, v * c), A, in_col, out_col)
        return y.view(n, y.shape[1], v, self.out_channels).permute(0, 3, 1, 2).contiguous(), A


def default_cols(V: int, C: int) -> np.ndarray:
    """column index of (vertex v, channel c) in the plain (v, c) order"""
    return (np.arange(V)[:, None] * C + np.arange(C)[None, :]).astype(np.int64)


def _gcn_fold(A, Cin, Cout, kt, in_col, out_col, device):
    K, V, _ = A.shape
    rows, cols, vals = [], [], []
    ks, vs, ws = np.nonzero(A)
    dt = np.arange(kt)
    for k, v, w in zip(ks, vs, ws):
        a = A[k, v, w]
        c, ci, d = np.meshgrid(np.arange(Cout), np.arange(Cin), dt, indexing='ij')
        dst = (out_col[w, c] * kt + d) * (V * Cin) + in_col[v, ci]          # tap-major (V*Cout, kt, V*Cin)
        src = ((k * Cout + c) * Cin + ci) * kt + d
        rows.append(dst.ravel())
        cols.append(src.ravel())
        vals.append(np.full(dst.size, a))
    import scipy.sparse as sp
    wmat = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                         shape=(V * Cout * V * Cin * kt, K * Cout * Cin * kt))
    # bias: b'[(w,c)] = sum_k b[k*Cout+c] * sum_v A[k,v,w]
    colsum = A.sum(1)                                             # (K, W)
    k, w, c = np.meshgrid(np.arange(K), np.arange(V), np.arange(Cout), indexing='ij')
    keep = colsum[k, w] != 0
    bmat = sp.coo_matrix((colsum[k, w][keep], (out_col[w, c][keep], (k * Cout + c)[keep])),
                         shape=(V * Cout, K * Cout))
    return ops.CSR(wmat, device), ops.CSR(bmat, device)


def _vertex_conv_fold(V, Cin, Cout, kt, kv, in_col, out_col, device):
    """Conv2d (Cout, Cin, kt, kv) zero-padded by kv//2 over vertices -> dense (V*Cout, V*Cin, kt)."""
    import scipy.sparse as sp
    rows, cols = [], []
    for w in range(V):
        for dv in range(kv):
            v = w + dv - kv // 2
            if v < 0 or v >= V:
                continue
            c, ci, d = np.meshgrid(np.arange(Cout), np.arange(Cin), np.arange(kt), indexing='ij')
            rows.append(((out_col[w, c] * kt + d) * (V * Cin) + in_col[v, ci]).ravel())   # tap-major
            cols.append((((c * Cin + ci) * kt + d) * kv + dv).ravel())
    r, cc = np.concatenate(rows), np.concatenate(cols)
    wmat = sp.coo_matrix((np.ones(r.size), (r, cc)), shape=(V * Cout * V * Cin * kt, Cout * Cin * kt * kv))
    w, c = np.meshgrid(np.arange(V), np.arange(Cout), indexing='ij')
    bmat = sp.coo_matrix((np.ones(w.size), (out_col[w, c].ravel(), c.ravel())), shape=(V * Cout, Cout))
    return ops.CSR(wmat, device), ops.CSR(bmat, device)


class STGraphConv(nn.Module):
    def __init__(self, in_channels, out_channels, A_channels, kernel_size, stride=(1, 1), padding=(0, 0), dropout=0,
                 activation='LeakyRelU', residual=True):
        super().__init__()
        assert len(kernel_size) == 2
        assert kernel_size[0] % 2 == 1
        if tuple(stride) != (1, 1) or dropout != 0:
            raise NotImplementedError('S2AG hot path uses stride (1,1) and dropout 0 (multimodal_context_net_v2.py:'
                                      '126-138)')
        if tuple(padding) != (kernel_size[0] // 2, kernel_size[1] // 2):
            raise NotImplementedError("only 'same' padding is used on the hot path")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kt, self.kv = kernel_size
        self.gcn = ConvTemporalGraphical(in_channels, out_channels, A_channels, kernel_size[0],
                                         temporal_stride=stride[0], temporal_padding=padding[0])
        self.tcn = nn.Sequential(
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, kernel_size, stride, padding),
            nn.BatchNorm2d(out_channels),
            nn.Dropout(dropout, inplace=True),
        )
        if not residual:
            self.residual = zero
        elif (in_channels == out_channels) and (stride == 1):      # never true: stride is a tuple (tgcn.py:195)
            self.residual = identity
        else:
            self.residual = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=stride),
                                          nn.BatchNorm2d(out_channels))
        if activation.lower() == 'leakyrelu':
            self.slope = 0.01
        elif activation.lower() == 'relu':
            self.slope = 0.0
        else:
            raise ValueError(activation)
        self._cache = {}
        self._groups = {}

    def _aux(self, V, in_col, out_col, device):
        key = (V, in_col.tobytes(), out_col.tobytes(), str(device))
        if key not in self._cache:
            tf = _vertex_conv_fold(V, self.out_channels, self.out_channels, self.kt, self.kv, out_col, out_col, device)
            rf = _vertex_conv_fold(V, self.in_channels, self.out_channels, 1, 1, in_col, out_col, device) \
                if isinstance(self.residual, nn.Module) else None
            cmap = np.empty(V * self.out_channels, dtype=np.int32)
            for c in range(self.out_channels):
                cmap[out_col[:, c]] = c
            self._cache[key] = (tf, rf, torch.from_numpy(cmap).to(device))
        return self._cache[key]

    def _fold_group(self, A, in_col, out_col, device):
        """All folded tensors of the block as ONE ops.FoldGroup (one launch per optimizer step, shared by every
        forward pass in between; gradients staged and flushed in one launch)."""
        key = (A.data_ptr(), in_col.tobytes(), out_col.tobytes(), str(device))
        if key not in self._groups:
            V, Co, Ci = A.shape[1], self.out_channels, self.in_channels
            (tw, tb), rf, cmap = self._aux(V, in_col, out_col, device)
            gw, gb = self.gcn.folds(A, in_col, out_col)
            params = [self.gcn.conv.weight, self.gcn.conv.bias, self.tcn[2].weight, self.tcn[2].bias]
            csrs = [gw, gb, tw, tb]
            shapes = [(V * Co, self.kt, V * Ci), (V * Co,), (V * Co, self.kt, V * Co), (V * Co,)]
            if isinstance(self.residual, nn.Module):
                params += [self.residual[0].weight, self.residual[0].bias]
                csrs += list(rf)
                shapes += [(V * Co, 1, V * Ci), (V * Co,)]
            self._groups[key] = (ops.FoldGroup(params, csrs, shapes), cmap)
        return self._groups[key]

    def forward_nlc(self, x, A, in_col, out_col):
        """x (N, T, V*Cin) with column order ``in_col`` -> (N, T, V*Cout) in column order ``out_col``."""
        grp, cmap = self._fold_group(A, in_col, out_col, x.device)
        f = grp.tensors()
        g = ops.conv1d_nlc(x, f[0], f[1], pad=self.gcn.pad, w_tap_major=True, bn_stats=True)                  # graph conv + einsum
        h = ops.batch_norm_act(g, self.tcn[0], slope=0.0, chan_map=cmap)                       # BN2d + ReLU
        h = ops.conv1d_nlc(h, f[2], f[3], pad=self.kt // 2, w_tap_major=True, bn_stats=True)
        h = ops.batch_norm_act(h, self.tcn[3], slope=1.0, chan_map=cmap)
        if isinstance(self.residual, nn.Module):
            r = ops.conv1d_nlc(x, f[4], f[5], w_tap_major=True, bn_stats=True)
            r = ops.batch_norm_act(r, self.residual[1], slope=1.0, chan_map=cmap)
            return ops.add_act(h, r, self.slope)
        return ops.add_act(h, None, self.slope)

    def forward(self, x, A):
        n, c, t, v = x.shape
        in_col, out_col = default_cols(v, c), default_cols(v, self.out_channels)
        y = self.forward_nlc(x.permute(0, 2, 3, 1).reshape(n, t, v * c), A, in_col, out_col)
        return y.view(n, t, v, self.out_channels).permute(0, 3, 1, 2).contiguous(), A
