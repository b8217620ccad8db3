"""Stand-alone timing of the fp32-operand transpose-read weight-gradient launch at a GRU layer's shapes."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech2affective_gestures_amd import _lib as L  # noqa: E402

lib = L.load()
B, T, H, In = int(os.environ.get('B', 128)), 34, 300, int(os.environ.get('IN', 600))
H3 = 3 * H
dev = 'cuda'
dgi = torch.randn(B * T, 2 * H3, device=dev)
dgh = torch.randn(2, B * T, H3, device=dev)
inp = torch.randn(B * T, In, device=dev)
y = torch.randn(B * T, 2 * H, device=dev)
dw_ih = torch.zeros(2 * H3, In, device=dev)
db_ih = torch.zeros(2 * H3, device=dev)
dw_hh = [torch.zeros(H3, H, device=dev) for _ in range(2)]
db_hh = [torch.zeros(H3, device=dev) for _ in range(2)]
p = lambda t: C.c_void_p(t.data_ptr())
jobs = (L.BF16Wgrad * 3)()
jobs[0] = L.BF16Wgrad(p(dgi), p(inp), p(dw_ih), p(db_ih), B, T, T, T * In, In, 2 * H3, 1, 0, 0, 1, In, In, 2 * H3, In, In, 0, 1, 0, 1)
for d in range(2):
    jobs[1 + d] = L.BF16Wgrad(C.c_void_p(dgh[d].data_ptr()), C.c_void_p(y.data_ptr() + 4 * d * H), p(dw_hh[d]), p(db_hh[d]), B, T, T,
                              T * 2 * H, 2 * H, H3, 1, -1 if d == 0 else 1, 0, 1, H, H, H3, H, H, 0, 1, 0, 1)
nj = int(os.environ.get('NJ', 3))
need = int(lib.s2ag_f32_wgrad_tr_scratch_floats(jobs, nj))
sc = torch.empty(need, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    L.check(lib.s2ag_f32_wgrad_tr(jobs, nj, p(sc), need, st), 'tr32')
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
N = 20
for _ in range(N):
    lib.s2ag_f32_wgrad_tr(jobs, nj, p(sc), need, st)
b.record()
torch.cuda.synchronize()
flops = 2.0 * B * T * (2 * H3 * In + (2 * H3 * H if nj == 3 else 0))
ms = a.elapsed_time(b) / N
print(f'B={B} In={In} jobs={nj}: {ms * 1e3:.1f} us per launch (+reduce), {flops / ms / 1e9:.1f} TFLOP/s fp32-equivalent, scratch {need * 4 / 1e6:.1f} MB')
# accuracy vs fp64
ref = (dgi.double().T @ inp.double())
dw_ih.zero_()
lib.s2ag_f32_wgrad_tr(jobs, nj, p(sc), need, st)
torch.cuda.synchronize()
print('dW_ih rel err', float((dw_ih.double() - ref).abs().max() / ref.abs().max()))

buf = torch.zeros(256, dtype=torch.int64, device=dev)
lib.s2ag_wgrad_tr_set_trace(C.c_void_p(buf.data_ptr()))
lib.s2ag_f32_wgrad_tr(jobs, nj, p(sc), need, st)
torch.cuda.synchronize()
lib.s2ag_wgrad_tr_set_trace(None)
t = [v for v in buf.cpu().tolist() if v]
d = [b - a for a, b in zip(t, t[1:])]
print('stamps (cycles): per step [stash, fetch+barrier, mma]:')
for i in range(0, min(len(d), 36), 3):
    print('  ', d[i:i + 3])
