import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from speech2affective_gestures_amd import ops, noise
B = 128
pr = bench.build_processor(B, True)
noise.manual_seed(1234)
text, audio, mfcc, target, vid = bench.synthetic_batch(B, 0, pr.device)
ts = []
for i in range(26):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr.train_step(text, audio, mfcc, target, vid, sync=False)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(['%.1f' % t for t in ts], flush=True)
print('timeouts', ops.coop_gru_timeouts(), 'losses', pr._finish(pr._graphed['out']['comps'], pr._graphed['out']['dis']), pr.last_losses)
