"""GRU forward+backward micro-benchmark: cooperative vs streaming kernels (HIP events)."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech2affective_gestures_amd import ops, noise
def run(B, T, I, H, L, coop_min):
    ops.COOP_GRU_MIN_H = coop_min
    g = torch.Generator().manual_seed(0); k = 1 / math.sqrt(H); ws = []
    for l in range(L):
        for d in range(2):
            In = I if l == 0 else 2 * H
            ws += [(torch.rand(3*H, In, generator=g)*2-1)*k, (torch.rand(3*H, H, generator=g)*2-1)*k, (torch.rand(3*H, generator=g)*2-1)*k, (torch.rand(3*H, generator=g)*2-1)*k]
    wg = [w.cuda().requires_grad_(True) for w in ws]
    x = torch.randn(B, T, I, generator=g).cuda().requires_grad_(True)
    nz = noise.begin_pass('cuda')
    def fwd(): return ops.gru(x, wg, H, L, True, 0.3, nz, 700, True)
    def t(fn, n=20):
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): fn()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr): fn()
        gr.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): gr.replay()
        b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
    def fwd_ng():
        with torch.no_grad(): return fwd()
    tf = t(fwd_ng)
    def fb():
        y = fwd(); y.sum().backward()
    tfb = t(fb)
    return tf, tfb
for (B, T, I, H, L) in ((128, 34, 8, 64, 4), (128, 34, 88, 300, 4)):
    for cm in (64, 10000):
        tf, tfb = run(B, T, I, H, L, cm)
        print(f'H={H} L={L} B={B} coop_min_H={cm}: fwd {tf:8.1f} us  fwd+bwd {tfb:8.1f} us', flush=True)
