"""Sliding-window synthesis (Processor.synthesize_clip) at full model size: ms per 34-frame window for a
60 s clip (30 windows; tri-modal baseline + s2ag generator per window, seed hand-off and cross-fade on the device)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pr = bench.build_processor(128, False)
pr.use_hip_graph = os.environ.get("S2AG_SYNTH_GRAPH", "1") != "0"
pr.s2ag_config_args.motion_resampling_framerate = 15
pr.lang_model = type('L', (), {'get_word_index': staticmethod(lambda w: 4 + (hash(w) % 1000))})()
rs = np.random.RandomState(0)
SR, SEC = 16000, float(os.environ.get('SEC', 60))
audio = (rs.randn(int(SEC * SR)) * 0.05).astype(np.float32)
words = [['w%d' % i, 0.4 * i, 0.4 * i + 0.3] for i in range(int(SEC / 0.4))]
seed = (rs.randn(4, 27) * 0.2).astype(np.float32)
mfcc_fn = lambda win: np.zeros((bench.NUM_MFCC, bench.MFCC_LEN), dtype=np.float32) + 0.01  # noqa: E731
for it in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out_t, out_g = pr.synthesize_clip(seed, audio, SR, words, mfcc_fn=mfcc_fn, speaker_vid_idx=5)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'call {it}: {dt * 1e3:.1f} ms, graph cached: {pr._synth["graph"] is not None}, key tail {pr._synth["key"][-2:]}', flush=True)
W = (out_g.shape[0] - 4) // 30
print(f'{SEC:.0f} s clip: {W} windows, {out_g.shape[0]} frames in {dt * 1e3:.1f} ms = {dt * 1e3 / W:.2f} ms per window '
      f'(both generators), {out_g.shape[0] / 15 / dt:.0f}x real time', flush=True)
