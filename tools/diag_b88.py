import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from speech2affective_gestures_amd import ops, noise
lib = ops._lib()
for (M, K, N) in ((2992, 1800, 600), (2992, 600, 1800), (4352, 1800, 600)):
    g = torch.Generator().manual_seed(M + K + N)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    ref = a.double() @ w.double().t()
    for pieces in (3, 2):
        lib.s2ag_gru_coop_set_split_pieces(pieces)
        ap, wp = ops.split_planes_raw(a.cuda()), ops.split_planes_raw(w.cuda())
        y = torch.empty(M, N, device='cuda')
        ops.gemm_split_raw(ap, wp, None, y, K)
        d = (y.double().cpu() - ref).abs() / ref.abs().max()
        bad = (d > 1e-4).nonzero()
        print(f'gemm_split M={M} K={K} N={N} pieces={pieces}: max err {float(d.max()):.2e}; bad elements {len(bad)}'
              + (f' rows {bad[:, 0].min().item()}..{bad[:, 0].max().item()} cols {bad[:, 1].min().item()}..{bad[:, 1].max().item()}' if len(bad) else ''), flush=True)
lib.s2ag_gru_coop_set_split_pieces(-1)
# GRU fwd/bwd vs fp64 torch at B = 88 with / without the split GEMMs
from test_gpu_ops import _gru_sd, _flat
for B in (88, 96, 128):
    I, H, L_, T = 600, 300, 1, 34
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
    for pieces in (3, 2):
        for split_gemm in (True, False):
            lib.s2ag_gru_coop_set_split_pieces(pieces)
            ops.SPLIT_GEMM = split_gemm
            wg = [w.cuda().requires_grad_(True) for w in _flat(sd, L_)]
            xg = x.cuda().requires_grad_(True)
            yg = ops.gru(xg, wg, H, L_, True, 0.0, noise.begin_pass('cuda'), 300, False)
            yg.backward(dy.cuda())
            err = lambda a_, b_: float((a_.detach().double().cpu() - b_.detach()).abs().max() / b_.detach().abs().max())
            print(f'GRU B={B} pieces={pieces} split_gemm={split_gemm}: y {err(yg, yr):.2e} dx {err(xg.grad, xr.grad):.2e} '
                  f'dWhh {err(wg[1].grad, ref.weight_hh_l0.grad):.2e} dWih {err(wg[0].grad, ref.weight_ih_l0.grad):.2e}', flush=True)
lib.s2ag_gru_coop_set_split_pieces(-1)
