import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from speech2affective_gestures_amd import ops, noise
B = 128
for graph in (False, True):
    pr = bench.build_processor(B, graph)
    batch = bench.synthetic_batch(B, 0, pr.device)
    text, audio, mfcc, target, vid = batch
    for i in range(2):
        pr.train_step(text, audio, mfcc, target, vid, sync=False)
    torch.cuda.synchronize()
    ops._COOP_WS.clear()
    t0 = time.perf_counter()
    for i in range(3):
        pr.train_step(text, audio, mfcc, target, vid, sync=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print('graph' if graph else 'eager', 'ms/step', dt * 1e3, 'timeouts among recent ws:', ops.coop_gru_timeouts(), len(ops._COOP_WS), flush=True)
    if graph:
        # time each segment
        segs = pr._graphed['segs']
        for i, g in enumerate(segs.graphs):
            torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
            print(' segment', i, (time.perf_counter() - t0) * 1e3, 'ms', flush=True)
