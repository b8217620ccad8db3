"""Phase stamps of s2ag_wave12_bwd's workgroup 0 at the configs[3] shape (B = 256): where a step's time goes.
MODE=bf16|fp32"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech2affective_gestures_amd import _lib as L
from speech2affective_gestures_amd import wave12

lib = L.load()
bf = os.environ.get('MODE', 'bf16') == 'bf16'
N, Lin = int(os.environ.get('B', 256)), 36267
dev = 'cuda'
g = torch.Generator().manual_seed(0)
x = (torch.randn(N, Lin, generator=g) * 0.05).to(dev)
w1, b1 = torch.randn(16, 1, 15, generator=g).to(dev) * 0.25, torch.randn(16, generator=g).to(dev) * 0.1
g1, e1 = (torch.rand(16, generator=g) + 0.5).to(dev), torch.randn(16, generator=g).to(dev) * 0.3
w2 = torch.randn(32, 16, 15, generator=g).to(dev) / 15.5


class BN:
    def __init__(self, c):
        self.running_mean, self.running_var = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        self.num_batches_tracked = torch.zeros((), dtype=torch.int64, device=dev)
        self.eps, self.momentum = 1e-5, 0.1


pk = wave12.packed_weights(w1, w2)
coef1 = wave12.stats(x, pk, b1, BN(16), g1, e1, bf)
L1, L2 = wave12.lengths(Lin)
slots = {k: torch.zeros(s, device=dev) for k, s in (('w1', (16, 1, 15)), ('g1', (16,)), ('e1', (16,)), ('w2', (32, 16, 15)))}
if bf:
    dz = torch.randn(N, L2, 32, generator=g).to(torch.bfloat16).to(dev)
    z2 = torch.randn(N, L2, 32, generator=g).to(torch.bfloat16).to(dev)
    cabc = torch.stack([torch.ones(32), torch.zeros(32), torch.zeros(32)]).to(dev)
    args = (dz, z2, cabc)
else:
    args = (torch.randn(N, L2, 32, generator=g).to(dev), None, None)
for _ in range(3):
    wave12.backward(x, pk, b1, coef1, g1, 0.3, *args, slots)
torch.cuda.synchronize()
tr = torch.zeros(128, dtype=torch.int64, device=dev)
lib.s2ag_wave12_set_trace(C.c_void_p(tr.data_ptr()))
wave12.backward(x, pk, b1, coef1, g1, 0.3, *args, slots)
torch.cuda.synchronize()
lib.s2ag_wave12_set_trace(None)
t = tr.cpu().tolist()
t = [v for v in t if v]
# s_memtime runs at 100 MHz: 10 ns per tick
names = ['stash+fetch issue', 'barrier 1', 'phase 2', 'barrier 2', 'phase 3']
per = [[] for _ in names]
i = 0
while i + 5 < len(t):
    for k in range(5):
        per[k].append((t[i + k + 1] - t[i + k]) * 10)
    i += 5
print(f'mode={"bf16" if bf else "fp32"} B={N}: {len(per[0])} steps of workgroup 0 (ns, median / mean)')
for nm, v in zip(names, per):
    v2 = sorted(v)
    print(f'  {nm:20s} {v2[len(v2) // 2]:7.0f} {sum(v) / len(v):7.0f}')
print('  step total (mean)   ', sum(sum(v) for v in per) / len(per[0]))
