"""CPU dry run of the wave encoder in both precision modes (bf16._WaveFused16; fp32: fused head + layer-by-layer tail): the real ctypes signatures and the
library's own argument validation, with every launch failing for want of a device (its hipError_t is recorded, not raised).
Catches host-side slips -- argument counts / types, null or misaligned pointers, geometry the entry points reject, shapes,
attribute names -- before a GPU sees the code.  Prints one JSON line; run by tests/test_host_logic.py."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech2affective_gestures_amd import _lib as L, bf16, ops, wave12    # noqa: E402
from speech2affective_gestures_amd.net.multimodal_context_net_v2 import WavEncoder  # noqa: E402

rcs = []
L.check = lambda rc, what='': rcs.append((what, int(rc)))
wave12._s = bf16._s = lambda: None
wave12._check_wav = lambda wav: None
ops.run_wgrad = lambda launch, keep=(), flops=0.0: launch()
_tk = torch.zeros(256, dtype=torch.int32)
ops._tickets = lambda dev, n: C.c_void_p(_tk.data_ptr())

enc = WavEncoder().train()
fe = enc.feat_extractor
mode = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
wav = torch.randn(2, 36267) * 0.05
if mode == 'bf16':                       # the default bf16-mode encoder (bf16._WaveFused16)
    with bf16.precision('bf16'):
        out = enc(wav)
    n_fwd = n_sg = len(rcs)
else:                     # the default fp32-mode encoder: fused head (wave12) + layer-by-layer tail (ops)
    ops._stream = lambda: None
    ops._need_cuda = lambda *a: None
    ops.join_side_streams = lambda *a, **k: None
    ops._ticket = lambda dev: C.c_void_p(_tk.data_ptr())
    ops._barrier = lambda dev: C.c_void_p(_tk.data_ptr())
    torch.cuda.is_current_stream_capturing = lambda: False
    enc.forward.__func__.__globals__['torch'].Tensor.is_cuda = property(lambda self: True)
    out = enc(wav)
    n_fwd = n_sg = len(rcs)
(out * torch.randn_like(out)).sum().backward()
print(json.dumps({
    'out': list(out.shape), 'forward': [w for w, _ in rcs[:n_fwd]], 
    'backward': [w for w, _ in rcs[n_sg:]], 'codes': sorted(set(rc for _, rc in rcs)),
    'grads': {k: (None if p.grad is None else list(p.grad.shape)) for k, p in enc.named_parameters()}}))
