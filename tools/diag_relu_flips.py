"""Is a large TCN weight-gradient mismatch at H = 300 a kernel bug or ReLU sign flips at ~0?  One causal dilated conv
(300 -> 300, 2 taps) + ReLU at M = B*34 rows in the three product modes: output error, number of elements whose ReLU mask
differs from the CPU fp32 conv, and the largest |pre-activation| (CPU fp64) among those elements."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from speech2affective_gestures_amd import ops, _lib as L

lib = L.load()
for B in (88, 128):
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 34, 300, generator=g)
    w = torch.randn(300, 2, 300, generator=g) / 24.0          # tap-major (Cout, k, Cin)
    b = torch.randn(300, generator=g) * 0.1
    xc = x.transpose(1, 2).double()
    wc = w.permute(0, 2, 1).double()
    pre = F.conv1d(F.pad(xc, (2, 0)), wc, b.double(), dilation=2)          # causal, dilation 2
    pre32 = F.conv1d(F.pad(xc.float(), (2, 0)), wc.float(), b, dilation=2)
    for pieces in (0, 2, 3):
        lib.s2ag_gru_coop_set_split_pieces(pieces)
        wd = w.cuda()
        if hasattr(wd, '_s2ag_wp'):
            del wd._s2ag_wp
        y = ops.conv1d_nlc(x.cuda(), wd, b.cuda(), pad=2, dil=2, lout=34, act=L.ACT_LEAKY, slope=0.0, w_tap_major=True)
        y = y.cpu().transpose(1, 2).double()
        ref = pre.clamp_min(0)
        err = float((y - ref).abs().max() / ref.abs().max())
        flips_vs_f64 = ((y > 0) != (pre > 0))
        flips_vs_f32 = ((y > 0) != (pre32 > 0))
        mx = float(pre[flips_vs_f64].abs().max()) if flips_vs_f64.any() else 0.0
        print(f'B={B} pieces={pieces}: out err {err:.2e}; mask flips vs fp64 {int(flips_vs_f64.sum())}, vs CPU fp32 '
              f'{int(flips_vs_f32.sum())} of {y.numel()}; largest |pre| among flips {mx:.2e}; '
              f'CPU fp32 itself flips vs fp64: {int(((pre32 > 0) != (pre > 0)).sum())}')
lib.s2ag_gru_coop_set_split_pieces(-1)
