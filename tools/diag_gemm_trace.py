"""Where one block of the implicit-GEMM kernel spends its cycles (block (0,0), thread 0): builds a traced copy of the
library (-DS2AG_GEMM_TRACE) and runs single launches at the step's shapes."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, 'speech2affective_gestures_amd', 'csrc')
out = '/tmp/libs2ag_gemm_trace.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
                       '-DS2AG_GEMM_TRACE', '-I' + os.path.join(ROOT, 'include'), '-I' + src] +
                      [os.path.join(src, f) for f in sorted(os.listdir(src)) if f.endswith('.hip')] + ['-o', out])
os.environ['S2AG_HIP_LIB'] = out
from speech2affective_gestures_amd import _lib as L, ops  # noqa: E402

lib = L.load()
raw = C.CDLL(out)
SHAPES = [  # name, M(N clips x Lin), Lin, Cin, Cout, ks, pad, dil, bwd
    ('gru_ih fwd   M4352 N1800 K600', 4352, 1, 600, 1800, 1, 0, 1, False),
    ('gru dx bwd   M4352 N600 K1800', 4352, 1, 600, 1800, 1, 0, 1, True),
    ('tcn fwd      M4352 N300 K600 ', 128, 34, 300, 300, 2, 4, 4, False),
    ('lin300 fwd   M4352 N300 K600 ', 4352, 1, 600, 300, 1, 0, 1, False),
]
names = ['prologue', 'load issue', 'lds+mfma', 'wait+stash', 'barrier', 'epilogue']
for name, N, Lin, Cin, Cout, ks, pad, dil, bwd in SHAPES:
    x = torch.randn(N, Lin, Cin, device='cuda')
    w = torch.randn(Cout, Cin, ks, device='cuda') * 0.05
    b = torch.randn(Cout, device='cuda')
    Lout = Lin
    y = torch.empty(N * Lout, Cout, device='cuda')
    gy = torch.randn(N * Lout, Cout, device='cuda')
    dx = torch.empty(N * Lin, Cin, device='cuda')

    def run():
        if bwd:
            ops.conv_bwd_data_raw(gy, w, dx, N, Lin, Lout, Cin, Cout, ks, 1, pad, dil, False)
        else:
            ops.conv_fwd_raw(x, w, b, y, N, Lin, Lout, Cin, Cout, ks, 1, pad, dil)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    tr = np.zeros(8, dtype=np.uint64)
    raw.s2ag_gemm_trace_read(tr.ctypes.data_as(C.c_void_p), 1)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    run()
    e.record()
    torch.cuda.synchronize()
    raw.s2ag_gemm_trace_read(tr.ctypes.data_as(C.c_void_p), 1)
    tot = float(tr[:6].sum())
    print(f'{name}: launch {a.elapsed_time(e) * 1e3:7.1f} us; block (0,0): {tot:9.0f} cycles = ' +
          ', '.join(f'{n} {100 * float(v) / tot:4.1f}%' for n, v in zip(names, tr[:6])))
