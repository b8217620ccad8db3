"""Launches per iteration of BASELINE configs[3] (WavEncoder + TextEncoderTCN forward + backward) and per GAN step, counted
on the CPU device model (tests/emu; the launch sequence does not depend on the batch size for these paths) -- a number that
needs no GPU.   python tools/count_launches_emu.py [cfg3|step] [fp32|bf16]   (switches: export S2AG_<NAME>=1 as usual)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests')); sys.path.insert(0, os.path.join(%(root)r, 'tests', 'emu'))
import harness; harness.install()
import torch
import bench
from speech2affective_gestures_amd import bf16, ops, noise
what, mode = %(what)r, %(mode)r
if what == 'cfg3':
    from speech2affective_gestures_amd.net.multimodal_context_net_v2 import TextEncoderTCN, WavEncoder
    from speech2affective_gestures_amd.optim import ParamArena
    cfg = bench.make_cfg()
    wav, txt = WavEncoder().train(), TextEncoderTCN(cfg, 500, 300, dropout=cfg.dropout_prob).train()
    arena = ParamArena(list(wav.parameters()) + list(txt.parameters()))
    B = 4
    text = torch.randint(0, 500, (B, 34)); audio = torch.randn(B, 36267) * 0.05
    g = torch.ones(B, 34, 32)
    def fn():
        ops.begin_step(); arena.zero_grad()
        wav(audio).backward(g); txt(text)[0].backward(g)
    with bf16.precision(mode):
        fn()
        sys.stderr.write('=== ITERATION ===\n'); sys.stderr.flush()
        fn()
        sys.stderr.write('=== END ===\n'); sys.stderr.flush()
else:
    sys.path.insert(0, os.path.join(%(root)r, 'tests'))
    from oracle import s2ag_oracle as O
    from s2ag_testing import to_cuda
    from test_gpu_step import make_processor
    pr, _ = make_processor(300, 64, 12, 6, 9300, 0.3, hip_graph=False)
    with bf16.precision(mode):
        for s in range(2):
            b = to_cuda(O.recipe_inputs(6, 34, 9400 + s, 64, 12))
            if s == 1:
                sys.stderr.write('=== ITERATION ===\n'); sys.stderr.flush()
            pr.forward_pass_s2ag(b['in_text'], b['in_audio'], b['in_mfcc'], b['target'], b['vid'], True)
        sys.stderr.write('=== END ===\n'); sys.stderr.flush()
'''


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
    mode = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
    env = dict(os.environ, S2AG_EMU='1', S2AG_EMU_TRACE='1')
    p = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT, what=what, mode=mode)], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if p.returncode:
        sys.stderr.write(p.stderr[-3000:])
        return p.returncode
    seg = p.stderr.split('=== ITERATION ===')[1].split('=== END ===')[0]
    names = [re.sub(r'^\(|\)$', '', m) for m in re.findall(r'launch (.+?) grid', seg)]
    cnt = collections.Counter(names)
    print(f'{what} {mode}: {len(names)} launches per ' + ('iteration' if what == 'cfg3' else 'step (eager, H = 300, B = 6)'))
    for k, v in cnt.most_common():
        print(f'  {v:4d}  {k}')
    return 0


if __name__ == '__main__':
    sys.exit(main())
