"""CPU dry run of TextEncoderTCN (argv[1]: bf16 mode, fp32 mode, or fp32 lockstep passes): real ctypes signatures and the
library's argument validation, every launch failing for want of a device (codes recorded, not raised).  The guards that make the product refuse CPU tensors are patched out HERE only.
Prints one JSON line; run by tests/test_host_logic.py."""
import json
import os
import sys
import types
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech2affective_gestures_amd import _lib as L, bf16, noise, ops                     # noqa: E402
from speech2affective_gestures_amd.net.multimodal_context_net_v2 import TextEncoderTCN   # noqa: E402

warnings.simplefilter('ignore')
rcs = []
L.check = lambda rc, what='': rcs.append((what, int(rc)))
bf16._s = lambda: None
bf16._rows16 = lambda t: ((t if t.is_contiguous() else t.contiguous()), t.numel() // t.shape[-1], t.shape[-1])
ops.run_wgrad = lambda launch, keep=(), flops=0.0: launch()
ops._stream = lambda: None
ops._need_cuda = lambda *a: None
ops.join_side_streams = lambda *a, **k: None
torch.cuda.is_current_stream_capturing = lambda: False
noise.begin_pass = lambda device: torch.zeros(2, dtype=torch.int64)
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'          # bf16 | fp32 | fp32_passes

args = types.SimpleNamespace(hidden_size=300, n_layers=4, freeze_wordembed=False)
enc = TextEncoderTCN(args, 1000).train()
ids = torch.randint(0, 1000, (4, 34))
if mode == 'bf16':
    with bf16.precision('bf16'):
        y, _ = enc(ids)
elif mode == 'fp32':
    y, _ = enc(ids)
else:                                                        # three passes in lockstep, the first with autograd
    noises = [torch.zeros(2, dtype=torch.int64) for _ in range(3)]
    outs = enc.forward_passes(ids, noises)
    assert len(outs) == 3 and all(o.shape == (4, 34, 32) for o in outs) and not outs[1].requires_grad
    y = outs[0]
n = len(rcs)
(y * torch.randn_like(y)).sum().backward()
print(json.dumps({'out': list(y.shape), 'forward': [w for w, _ in rcs[:n]], 'backward': [w for w, _ in rcs[n:]],
                  'refused': sorted(set(w for w, rc in rcs if rc == -1)),
                  'grads': {k: (None if p.grad is None else list(p.grad.shape)) for k, p in enc.named_parameters()}}))
