import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from speech2affective_gestures_amd import ops, noise
B = 128
pr = bench.build_processor(B, True)
noise.manual_seed(1234)
text, audio, mfcc, target, vid = bench.synthetic_batch(B, 0, pr.device)
pr.train_step(text, audio, mfcc, target, vid, sync=False)
torch.cuda.synchronize()
print('captured; ws kept', len(ops._COOP_WS), 'timeouts so far', ops.coop_gru_timeouts(), flush=True)
for n in (3, 8, 20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        pr.train_step(text, audio, mfcc, target, vid, sync=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    print(n, 'back-to-back: ms/step', round(dt, 1), 'timeouts', ops.coop_gru_timeouts(), flush=True)
