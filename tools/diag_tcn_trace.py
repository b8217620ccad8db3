"""Phase time line of the clip-resident TCN kernels (csrc/tcn_fused.hip): s_memtime stamps of workgroup 0, wave 0."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech2affective_gestures_amd import _lib as L, bf16, noise, ops  # noqa: E402
from speech2affective_gestures_amd.net.multimodal_context_net_v2 import TextEncoderTCN  # noqa: E402

B = int(os.environ.get('B', 256))
cfg = types.SimpleNamespace(hidden_size=300, n_layers=4, dropout_prob=0.3, freeze_wordembed=False)
txt = TextEncoderTCN(cfg, 20000, 300, dropout=0.3).cuda().train()
ids = torch.randint(0, 20000, (B, 34)).cuda()
buf = torch.zeros(256, dtype=torch.int64, device='cuda')
lib = L.load()
noise.manual_seed(1)
with bf16.precision('bf16'):
    for it in range(3):
        ops.begin_step()
        if it == 2:
            lib.s2ag_bf16_tcn_set_trace(buf.data_ptr())
        txt(ids)[0].sum().backward()
        torch.cuda.synchronize()
lib.s2ag_bf16_tcn_set_trace(None)
t = buf.cpu().tolist()
for name, lo in (('forward', 0), ('backward', 128)):
    st = [v for v in t[lo:lo + 128] if v]
    d = [(b - a) / 100.0 for a, b in zip(st, st[1:])]         # s_memtime ticks at 100 MHz -> us
    print(name, f'{(st[-1] - st[0]) / 100.0:.1f} us:', ' '.join(f'{x:.1f}' for x in d))
