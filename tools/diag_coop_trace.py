"""Per-phase timeline of one cooperative-GRU workgroup (forward, H = 300): builds a traced copy of the library
(-DS2AG_COOP_TRACE) next to the real one, runs one launch and prints wall-clock stamps (100 MHz) per time step."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech2affective_gestures_amd import _lib as L  # noqa: E402

src = os.path.join(ROOT, 'speech2affective_gestures_amd', 'csrc')
out = '/tmp/libs2ag_trace.so'
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-DS2AG_COOP_TRACE', *os.environ.get('EXTRA_DEFS', '').split(),
       '-I' + os.path.join(ROOT, 'include'), '-I' + src] + [os.path.join(src, f) for f in sorted(os.listdir(src))
                                                            if f.endswith('.hip')] + ['-o', out]
subprocess.check_call(cmd)
lib = C.CDLL(out)
B, T, H = int(os.environ.get('B', 128)), 34, 300
dev = 'cuda'
gi = torch.randn(B * T, 6 * H, device=dev) * 0.5
whh = torch.randn(2, 3 * H, H, device=dev) * 0.05
bhh = torch.randn(2, 3 * H, device=dev) * 0.05
y = torch.empty(B * T, 2 * H, device=dev)
yd = torch.empty_like(y)
gates = torch.empty(2, B * T, 4 * H, device=dev)
rng = torch.tensor([1, 0], dtype=torch.int64, device=dev)
e = L.Epilogue(0, 1.0, 0.3, C.c_void_p(rng.data_ptr()), 1)
lib.s2ag_gru_coop_workspace_bytes.restype = C.c_longlong
ws = torch.empty(lib.s2ag_gru_coop_workspace_bytes(B, T, H, 0), dtype=torch.uint8, device=dev)
vp = C.c_void_p
for _ in range(3):
    rc = lib.s2ag_gru_coop_fwd(vp(gi.data_ptr()), vp(whh.data_ptr()), vp(bhh.data_ptr()), vp(y.data_ptr()),
                               vp(yd.data_ptr()), vp(gates.data_ptr()), B, T, H, C.byref(e), vp(ws.data_ptr()), vp(0))
    assert rc == 0, rc
    torch.cuda.synchronize()
tr = np.zeros(64 * 8, dtype=np.uint64)
assert lib.s2ag_gru_coop_trace_read(tr.ctypes.data_as(C.c_void_p)) == 0
tr = tr.reshape(64, 8)[:T].astype(np.int64)
names = ['gather', 'barrier', 'mfma+red', 'gate math', 'stores']
print('step  rounds ' + ' '.join(f'{n:>12s}' for n in names) + '   step total (us)')
for s in range(1, T - 1):
    d = [(tr[s, i + 1] - tr[s, i]) / 100.0 for i in range(5)]
    print(f'{s:4d} {tr[s, 7]:7d} ' + ' '.join(f'{v:12.2f}' for v in d) + f'   {(tr[s + 1, 0] - tr[s, 0]) / 100.0:8.2f}'
          f'   mfma-phase shader cycles (wave 0, before the barrier): {tr[s, 6]}')


# ---- backward
dyt = torch.randn(B * T, 2 * H, device=dev)
dgi = torch.empty(B * T, 6 * H, device=dev)
dgh = torch.empty(2, B * T, 3 * H, device=dev)
ws2 = torch.empty(lib.s2ag_gru_coop_workspace_bytes(B, T, H, 1), dtype=torch.uint8, device=dev)
for _ in range(3):
    rc = lib.s2ag_gru_coop_bwd(vp(dyt.data_ptr()), 2 * H, H, vp(whh.data_ptr()), vp(y.data_ptr()), vp(gates.data_ptr()),
                               vp(dgi.data_ptr()), vp(dgh.data_ptr()), B, T, H, C.byref(e), vp(ws2.data_ptr()), vp(0))
    assert rc == 0, rc
    torch.cuda.synchronize()
tr = np.zeros(64 * 8, dtype=np.uint64)
assert lib.s2ag_gru_coop_trace_read(tr.ctypes.data_as(C.c_void_p)) == 0
tr = tr.reshape(64, 8)[:T].astype(np.int64)
names = ['gate grads+stores', 'barrier', 'mfma', 'publish', 'gather']
print('BACKWARD\nstep  rounds ' + ' '.join(f'{n:>14s}' for n in names) + '   step total (us)')
for s in range(1, T - 2):
    d = [(tr[s, i + 1] - tr[s, i]) / 100.0 for i in range(5)]
    print(f'{s:4d} {tr[s, 7]:7d} ' + ' '.join(f'{v:14.2f}' for v in d) + f'   {(tr[s + 1, 0] - tr[s, 0]) / 100.0:8.2f}')


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


sp = vp(torch.cuda.current_stream().cuda_stream)
tf = timed(lambda: lib.s2ag_gru_coop_fwd(vp(gi.data_ptr()), vp(whh.data_ptr()), vp(bhh.data_ptr()), vp(y.data_ptr()),
                                         vp(yd.data_ptr()), vp(gates.data_ptr()), B, T, H, C.byref(e),
                                         vp(ws.data_ptr()), sp))
tb = timed(lambda: lib.s2ag_gru_coop_bwd(vp(dyt.data_ptr()), 2 * H, H, vp(whh.data_ptr()), vp(y.data_ptr()),
                                         vp(gates.data_ptr()), vp(dgi.data_ptr()), vp(dgh.data_ptr()), B, T, H,
                                         C.byref(e), vp(ws2.data_ptr()), sp))
print(f'launch (incl. the clear kernel): forward {tf:.1f} us, backward {tb:.1f} us   [S2AG_COOP_L2={os.environ.get("S2AG_COOP_L2", "1")}]')


# ---- small-H GRU (H = 64) forward
Hs = 64
gis = torch.randn(B * T, 6 * Hs, device=dev) * 0.5
whs = torch.randn(2, 3 * Hs, Hs, device=dev) * 0.1
bhs = torch.randn(2, 3 * Hs, device=dev) * 0.1
ys = torch.empty(B * T, 2 * Hs, device=dev)
yds = torch.empty_like(ys)
gs = torch.empty(2, B * T, 4 * Hs, device=dev)
for _ in range(3):
    rc = lib.s2ag_gru_seq_fwd(vp(gis.data_ptr()), vp(whs.data_ptr()), None, vp(bhs.data_ptr()), vp(ys.data_ptr()),
                              vp(yds.data_ptr()), vp(gs.data_ptr()), B, T, Hs, C.byref(e), vp(0))
    assert rc == 0, rc
    torch.cuda.synchronize()
tr = np.zeros(64 * 8, dtype=np.uint64)
assert lib.s2ag_gru_small_trace_read(tr.ctypes.data_as(C.c_void_p)) == 0
tr = tr.reshape(64, 8)[:T].astype(np.int64)
names = ['gi prefetch issue', 'lds + fma', 'merge + gates', 'lds write + stores', 'barrier']
print('SMALL-H FORWARD\nstep ' + ' '.join(f'{n:>18s}' for n in names) + '   step total (us)')
for s_ in range(1, T - 1, 4):
    d = [(tr[s_, i + 1] - tr[s_, i]) / 100.0 for i in range(5)]
    print(f'{s_:4d} ' + ' '.join(f'{v:18.2f}' for v in d) + f'   {(tr[s_ + 1, 0] - tr[s_, 0]) / 100.0:8.2f}')
