"""Split-operand GEMM (gemm_sp.hip) vs the f32-MFMA straight-line GEMM at the GRU projection shape, and at smaller M
(fewer blocks than CUs) to separate per-block time from chip-wide effects."""
import math
import sys

import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from speech2affective_gestures_amd import ops  # noqa: E402


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr):
            for _ in range(10):
                fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n // 10):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n // 10 * 10) * 1e3


K, N = 600, 1800
for M in (4352, 2176, 1088, 512):
    a = torch.randn(M, K, device='cuda')
    w = torch.randn(N, K, device='cuda') / math.sqrt(K)
    b = torch.randn(N, device='cuda')
    y = torch.empty(M, N, device='cuda')
    ap, wp = ops.split_planes_raw(a), ops.split_planes_raw(w)
    ts = t(lambda: ops.split_planes_raw(a))
    tg = t(lambda: ops.gemm_split_raw(ap, wp, b, y, K))
    tl = t(lambda: ops.conv_fwd_raw(a, w, b, y, M, 1, 1, K, N, 1, 1, 0, 1))
    gf = 2.0 * M * N * K / 1e6
    print(f'M={M:5d}: split A {ts:6.1f} us | gemm_sp {tg:6.1f} us ({gf / tg:6.1f} TF fp32-equivalent) | gemm_lin {tl:6.1f} us '
          f'({gf / tl:6.1f} TF)', flush=True)
