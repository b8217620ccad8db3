import sys, os, time, math, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech2affective_gestures_amd import ops, noise, _lib as L
lib = L.load()
B, T, I, H, Lyr = 128, 34, 88, 300, 4
g = torch.Generator().manual_seed(0)
k = 1 / math.sqrt(H)
ws_ = []
for l in range(Lyr):
    for d in range(2):
        In = I if l == 0 else 2 * H
        ws_ += [((torch.rand(3 * H, In, generator=g) * 2 - 1) * k), ((torch.rand(3 * H, H, generator=g) * 2 - 1) * k),
                ((torch.rand(3 * H, generator=g) * 2 - 1) * k), ((torch.rand(3 * H, generator=g) * 2 - 1) * k)]
wg = [w.cuda().requires_grad_(True) for w in ws_]
x = torch.randn(B, T, I, generator=g).cuda().requires_grad_(True)
noise.manual_seed(1)

def report(tag):
    torch.cuda.synchronize()
    bad = []
    for i, (ws, b, t, h, bwd) in enumerate(list(ops._COOP_WS)):
        off = C.c_longlong(0); lib.s2ag_gru_coop_error_word_offset(b, t, h, bwd, C.byref(off))
        if int(ws[off.value:off.value + 4].view(torch.int32).item()) != 0:
            bad.append((i, 'bwd' if bwd else 'fwd'))
    print(tag, 'bad launches:', bad, flush=True)

def step():
    nz = noise.begin_pass('cuda')
    y = ops.gru(x, wg, H, Lyr, True, 0.3, nz, 500, True)
    y.square().mean().backward()

for mode in sys.argv[1:] or ['eager', 'graph']:
    ops._COOP_WS.clear()
    if mode == 'eager':
        for n in (1, 5, 20):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): step()
            torch.cuda.synchronize(); print('eager', n, 'ms/step', round((time.perf_counter() - t0) / n * 1e3, 2), flush=True)
        report('eager')
    else:
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): step()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        ops._COOP_WS.clear()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            step()
        for n in (1, 1, 5, 20, 20):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): gr.replay()
            torch.cuda.synchronize(); print('graph', n, 'ms/step', round((time.perf_counter() - t0) / n * 1e3, 2), flush=True)
            report('graph')
