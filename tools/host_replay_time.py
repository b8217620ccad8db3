"""Host-side cost of replaying the step's three hipGraph segments (time inside CUDAGraph.replay(), no device sync)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from speech2affective_gestures_amd import noise  # noqa: E402

B = int(os.environ.get('B', 128))
pr = bench.build_processor(B, True)
noise.manual_seed(1234)
text, audio, mfcc, target, vid = bench.synthetic_batch(B, 0, pr.device)
for _ in range(8):
    pr.train_step(text, audio, mfcc, target, vid, sync=False)
torch.cuda.synchronize()
segs = pr._graphed['segs']
tot = [0.0] * len(segs.graphs)
N = 20
for _ in range(N):
    torch.cuda.synchronize()
    for i, g in enumerate(segs.graphs):
        t0 = time.perf_counter()
        g.replay()
        tot[i] += time.perf_counter() - t0
torch.cuda.synchronize()
print('host ms per replay() call, per segment:', [round(1e3 * t / N, 3) for t in tot], 'sum', round(1e3 * sum(tot) / N, 3))
t0 = time.perf_counter()
for _ in range(N):
    pr.train_step(text, audio, mfcc, target, vid, sync=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host time to ISSUE a step: {1e3 * (t1 - t0) / N:.3f} ms; wall per step incl. drain: {1e3 * (t2 - t0) / N:.3f} ms')
