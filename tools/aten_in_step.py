"""Which ATen operators still launch kernels inside one eager training step (and from where)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pr = bench.build_processor(128, False, 34, bench.AUDIO_LEN)
batch = bench.synthetic_batch(128, 0, pr.device, 34, bench.AUDIO_LEN)
for _ in range(3):
    pr.train_step(*batch, sync=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    pr.train_step(*batch, sync=False)
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.name.startswith('aten::') and e.device_time_total > 0 and not any(c.name.startswith('aten::') and c.device_time_total > 0 for c in e.cpu_children):
        st = [s for s in e.stack if 'speech2affective' in s or 'bench' in s]
        rows.append((e.name, st[0] if st else (e.stack[0] if e.stack else '?'), e.device_time_total))
agg = {}
for n, s, t in rows:
    k = (n, s)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += t
print(f'{len(rows)} kernel-launching ATen calls in one step, {sum(t for _, _, t in rows):.0f} us')
for (n, s), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{c:4d} x {n:28s} {t:8.1f} us  {s}')
