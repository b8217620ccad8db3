"""Launch one implicit-GEMM shape a few times (for `rocprofv3 --pmc ...` passes): SHAPE=tcn|lin300|gru_ih|gru_dx."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech2affective_gestures_amd import ops  # noqa: E402

SHAPES = {  # N, Lin, Cin, Cout, ks, pad, dil, bwd
    'tcn': (128, 34, 300, 300, 2, 4, 4, False),
    'lin300': (4352, 1, 600, 300, 1, 0, 1, False),
    'gru_ih': (4352, 1, 600, 1800, 1, 0, 1, False),
    'gru_dx': (4352, 1, 600, 1800, 1, 0, 1, True),
    'gru_wg': (4352, 1, 600, 1800, 1, 0, 1, 'w'),
    'tcn_wg': (128, 34, 300, 300, 2, 4, 4, 'w'),
}
N, Lin, Cin, Cout, ks, pad, dil, bwd = SHAPES[os.environ.get('SHAPE', 'tcn')]
x = torch.randn(N, Lin, Cin, device='cuda')
w = torch.randn(Cout, Cin, ks, device='cuda') * 0.05
b = torch.randn(Cout, device='cuda')
y = torch.empty(N * Lin, Cout, device='cuda')
gy = torch.randn(N * Lin, Cout, device='cuda')
dx = torch.empty(N * Lin, Cin, device='cuda')
dw = torch.zeros(Cout, Cin, ks, device='cuda')
for _ in range(10):
    if bwd == 'w':
        ops.conv_bwd_weight_raw(gy, x, dw, N, Lin, Lin, Cin, Cout, ks, 1, pad, dil, True)
    elif bwd:
        ops.conv_bwd_data_raw(gy, w, dx, N, Lin, Lin, Cin, Cout, ks, 1, pad, dil, False)
    else:
        ops.conv_fwd_raw(x, w, b, y, N, Lin, Lin, Cin, Cout, ks, 1, pad, dil)
torch.cuda.synchronize()
