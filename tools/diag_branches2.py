import sys, os, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from speech2affective_gestures_amd import ops, noise
br, ov = sys.argv[1] == '1', sys.argv[2] == '1'
ops.PARALLEL_BRANCHES = br
B = 16
pr = bench.build_processor(B, True)
pr.overlap_passes = ov
text, audio, mfcc, target, vid = bench.synthetic_batch(B, 0, pr.device)
for i in range(3):
    pr.train_step(text, audio, mfcc, target, vid)
torch.cuda.synchronize(); print('branches', br, 'overlap', ov, 'OK', pr.last_losses['total'], flush=True)
