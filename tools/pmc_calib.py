"""Known-byte-count launches for calibrating FETCH_SIZE / WRITE_SIZE (run under rocprofv3 --pmc ...): each pattern of
s2ag_calib_traffic touches 1 GiB once.  tools/pmc_traffic.py reads the counters of these four kernels back."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech2affective_gestures_amd import _lib as L

lib = L.load()
n = 1 << 30
buf = torch.zeros(n, dtype=torch.uint8, device='cuda')
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for rep in range(2):
    for pattern in (0, 1, 2, 3):
        L.check(lib.s2ag_calib_traffic(C.c_void_p(buf.data_ptr()), n, pattern, s), 'calib')
        torch.cuda.synchronize()
print('calibration launches done:', n, 'bytes per pattern')
