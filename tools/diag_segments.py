import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from speech2affective_gestures_amd import ops, noise
pr = bench.build_processor(128, True)
text, audio, mfcc, target, vid = bench.synthetic_batch(128, 0, pr.device)
for _ in range(3): pr.train_step(text, audio, mfcc, target, vid)
segs = pr._graphed['segs']
for rep in range(2):
    for i, g in enumerate(segs.graphs):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); print('segment', i, round((time.perf_counter() - t0) / 5 * 1e3, 2), 'ms', flush=True)
# individual passes, eager-free: capture each pass separately
def timeit(fn, n=10):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
pre = pr._make_pre_seq(target)
G, D, T3 = pr.s2ag_generator, pr.s2ag_discriminator, pr.trimodal_generator
def g_fwd():
    with torch.no_grad(): G(pre, text, mfcc, vid)
def d_fwd():
    with torch.no_grad(): D(target)
def t_fwd():
    with torch.no_grad(): T3(pre, text, audio, vid)
def g_fb():
    pr.s2ag_gen_optimizer.zero_grad(); G(pre, text, mfcc, vid)[0].square().mean().backward()
def d_fb():
    pr.s2ag_dis_optimizer.zero_grad(); D(target).log().mean().backward()
def enc():
    with torch.no_grad():
        G.audio_encoder(mfcc); G.text_encoder(text); G.aff_encoder(target)
for name, fn in (('G fwd', g_fwd), ('D fwd', d_fwd), ('PGT fwd', t_fwd), ('G fwd+bwd', g_fb), ('D fwd+bwd', d_fb), ('G encoders fwd', enc),
                 ('G adam', lambda: pr.s2ag_gen_optimizer.step())):
    print(name, round(timeit(fn), 3), 'ms', flush=True)
