import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from oracle import s2ag_oracle as O
from speech2affective_gestures_amd import ops, noise
from test_gpu_ops import _gru_sd, _flat
lib = ops._lib()
def run(B, T, I, H, L, sum_dirs, p, pieces):
    lib.s2ag_gru_coop_set_split_pieces(pieces)
    sd = _gru_sd(I, H, L, B * 100 + H)
    g = torch.Generator().manual_seed(B + T)
    x = torch.randn(B, T, I, generator=g)
    noise.manual_seed(5)
    nz = noise.begin_pass('cuda')
    site0 = 300
    wg = [w.cuda().requires_grad_(True) for w in _flat(sd, L)]
    xg = x.cuda().requires_grad_(True)
    yg = ops.gru(xg, wg, H, L, True, p, nz, site0, sum_dirs)
    pinned = {f'gru.drop{l}': ops.dropout_mask(nz, site0 + l, p, (B, T, 2 * H)).cpu() for l in range(L - 1)} if p > 0 else 'off'
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = O.gru(sdr, 'gru.', xr, True, p, O.Noise(pinned), 'gru')
    if sum_dirs:
        yr = yr[..., :H] + yr[..., H:]
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    yg.backward(dy.cuda())
    e = lambda a, b: float((a.detach().cpu().double() - b.detach().double()).abs().max() / b.detach().abs().max())
    d = (xg.grad.cpu() - xr.grad).abs().amax(dim=(1, 2)) / xr.grad.abs().max()
    badclips = (d > 1e-4).nonzero().flatten().tolist()
    print(f'B={B} L={L} p={p} sum={sum_dirs} pieces={pieces}: y {e(yg, yr):.2e} dx {e(xg.grad, xr.grad):.2e} bad clips {badclips[:12]}{"..." if len(badclips) > 12 else ""}', flush=True)
for pieces in (2, 3):
    run(88, 34, 88, 300, 4, True, 0.3, pieces)
run(88, 34, 88, 300, 4, True, 0.0, 2)
run(88, 34, 88, 300, 2, True, 0.3, 2)
run(88, 34, 88, 300, 1, True, 0.0, 2)
run(80, 34, 88, 300, 4, True, 0.3, 2)
run(96, 34, 88, 300, 4, True, 0.3, 2)
run(40, 34, 88, 300, 4, True, 0.3, 2)
run(128, 34, 88, 300, 4, True, 0.3, 2)
lib.s2ag_gru_coop_set_split_pieces(-1)
