import sys, os, torch, types, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from speech2affective_gestures_amd import ops, noise
mode = sys.argv[1]
B = 16
ops.PARALLEL_BRANCHES = mode != 'off'
pr = bench.build_processor(B, False)
text, audio, mfcc, target, vid = bench.synthetic_batch(B, 0, pr.device)
pre = pr._make_pre_seq(target)
G = pr.s2ag_generator
def fwd_only():
    with torch.no_grad():
        return G(pre, text, mfcc, vid)[0]
def fwd_bwd():
    pr.s2ag_gen_optimizer.zero_grad()
    out = G(pre, text, mfcc, vid)[0]
    out.square().mean().backward()
    ops.join_side_streams()
fn = fwd_only if mode in ('fwd', 'off') else fwd_bwd
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): fn()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
print('warmup ok', flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fn()
print('capture ok', flush=True)
for _ in range(3): g.replay()
torch.cuda.synchronize(); print(mode, 'replay ok', flush=True)
