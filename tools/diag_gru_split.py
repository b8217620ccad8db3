"""Cooperative GRU (H = 300): f32-MFMA vs split-bf16 products (S2AG_GRU_SPLIT = 0 / 2 / 3) -- error against an fp64
torch reference and forward / forward+backward time (hipGraph replay), one sub-process per mode."""
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch
    sys.path.insert(0, ROOT)
    from speech2affective_gestures_amd import noise, ops
    B, T, I, H, L = 128, 34, 88, 300, 4
    g = torch.Generator().manual_seed(0)
    k = 1 / math.sqrt(H)
    ws = []
    for l in range(L):
        for d in range(2):
            In = I if l == 0 else 2 * H
            ws += [(torch.rand(3 * H, In, generator=g) * 2 - 1) * k, (torch.rand(3 * H, H, generator=g) * 2 - 1) * k,
                   (torch.rand(3 * H, generator=g) * 2 - 1) * k, (torch.rand(3 * H, generator=g) * 2 - 1) * k]
    x = torch.randn(B, T, I, generator=g)
    ref = torch.nn.GRU(I, H, L, batch_first=True, bidirectional=True).double()
    with torch.no_grad():
        i = 0
        for l in range(L):
            for suf in ('', '_reverse'):
                for kind in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
                    getattr(ref, f'{kind}_l{l}{suf}').copy_(ws[i].double())
                    i += 1
    xr = x.double().requires_grad_(True)
    yr, _ = ref(xr)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    wg = [w.cuda().requires_grad_(True) for w in ws]
    xg = x.cuda().requires_grad_(True)
    nz = noise.begin_pass('cuda')
    yg = ops.gru(xg, wg, H, L, True, 0.0, nz, 700, False)
    yg.backward(dy.cuda())
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
    e = [rel(yg, yr.detach()), rel(xg.grad, xr.grad), rel(wg[1].grad, ref.weight_hh_l0.grad),
         rel(wg[0].grad, ref.weight_ih_l0.grad)]

    print(f"S2AG_GRU_SPLIT={os.environ.get('S2AG_GRU_SPLIT')}: max-rel error vs fp64  y {e[0]:.2e}  dx {e[1]:.2e}  "
          f"dW_hh(l0) {e[2]:.2e}  dW_ih(l0) {e[3]:.2e}  timeouts {ops.coop_gru_timeouts()}", flush=True)

    def t(fn, n=30):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        gr.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            gr.replay()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3

    def fwd():
        return ops.gru(xg, wg, H, L, True, 0.3, nz, 700, True)

    def fwd_ng():
        with torch.no_grad():
            return fwd()

    print(f"S2AG_GRU_SPLIT={os.environ.get('S2AG_GRU_SPLIT')}: 4-layer forward (B=128, T=34, dropout 0.3) "
          f"{t(fwd_ng):7.1f} us", flush=True)
else:
    for mode in sys.argv[1:] or ('0', '2', '3'):
        env = dict(os.environ, S2AG_GRU_SPLIT=mode)
        subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, check=False, timeout=300)
