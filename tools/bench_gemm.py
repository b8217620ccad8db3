"""Micro-benchmark of the implicit-GEMM conv family at the step's shapes (HIP events on the launch stream)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech2affective_gestures_amd import ops

SHAPES = [  # name, N, Lin, Cin, Cout, ks, stride, pad, dil, causal
    ('gru_ih  M4352 N900 K600', 4352, 1, 600, 900, 1, 1, 0, 1, False),
    ('gru_ih0 M4352 N900 K88 ', 4352, 1, 88, 900, 1, 1, 0, 1, False),
    ('tcn     M4352 N300 K600', 128, 34, 300, 300, 2, 1, 4, 4, True),
    ('lin300  M4352 N300 K600', 4352, 1, 600, 300, 1, 1, 0, 1, False),
    ('lin144  M4352 N144 K1296', 4352, 1, 1296, 144, 1, 1, 0, 1, False),
    ('stgcn   M4352 N144 K1296', 128, 34, 144, 144, 9, 1, 4, 1, False),
    ('stgcn   M4352 N144 K243 ', 128, 34, 27, 144, 9, 1, 4, 1, False),
    ('mfcc    M4736 N64  K320 ', 128, 37, 64, 64, 5, 1, 2, 1, False),
    ('out0    M4352 N150 K300 ', 4352, 1, 300, 150, 1, 1, 0, 1, False),
    ('wav1    M1.01M N16 K15  ', 128, 36267, 1, 16, 15, 5, 1600, 1, False),
    ('wav2    M168k N32 K240  ', 128, 7891, 16, 32, 15, 6, 0, 1, False),
]

def timeit(fn, iters=30):
    for _ in range(5): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3

for name, N, Lin, Cin, Cout, ks, stride, pad, dil, causal in SHAPES:
    x = torch.randn(N, Lin, Cin, device='cuda'); w = torch.randn(Cout, Cin, ks, device='cuda') * 0.05; b = torch.randn(Cout, device='cuda')
    Lout = Lin if causal else (Lin + 2 * pad - dil * (ks - 1) - 1) // stride + 1
    y = torch.empty(N * Lout, Cout, device='cuda'); g = torch.randn(N * Lout, Cout, device='cuda')
    dx = torch.empty(N * Lin, Cin, device='cuda'); dw = torch.empty_like(w)
    fl = 2.0 * N * Lout * Cout * Cin * ks
    f = timeit(lambda: ops.conv_fwd_raw(x, w, b, y, N, Lin, Lout, Cin, Cout, ks, stride, pad, dil))
    bd = timeit(lambda: ops.conv_bwd_data_raw(g, w, dx, N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, False))
    bw = timeit(lambda: ops.conv_bwd_weight_raw(g, x, dw, N, Lin, Lout, Cin, Cout, ks, stride, pad, dil, False))
    print(f'{name}: {fl/1e9:6.2f} GFLOP | fwd {f:7.1f}us {fl/f/1e6:6.1f} TF | bwd_data {bd:7.1f}us {fl/bd/1e6:6.1f} TF | bwd_weight {bw:7.1f}us {fl/bw/1e6:6.1f} TF', flush=True)
