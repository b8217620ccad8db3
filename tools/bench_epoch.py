"""PCIe-inclusive rate: `Processor.per_train_epoch`-style loop (yield_batch -> train_step) over a synthetic TED-shaped
numpy dataset held in HOST memory, with the prefetching feeder (data.BatchFeeder) and with the reference-shaped host
path (synchronous fancy-indexing, float64 decode on the host, blocking copies)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

N = int(os.environ.get('N', 4096))
B = 128
rs = np.random.RandomState(0)
text = np.zeros((N, bench.T), dtype=np.int64)
for i in range(N):
    k = rs.randint(2, 9)
    text[i, rs.permutation(bench.T)[:k]] = rs.randint(4, bench.N_WORDS, k)
samples = dict(extended_word_seq=text, vec_seq=rs.randn(N, bench.T, bench.POSE_DIM) * 0.2,
               audio=np.clip(rs.randn(N, bench.AUDIO_LEN) * 0.05 * 32767, -32767, 32767).astype(np.int16),
               audio_max=np.ones(N), mfcc_features=(rs.randn(N, bench.NUM_MFCC, bench.MFCC_LEN) * 0.1).astype(np.float16),
               vid_indices=rs.randint(0, bench.N_SPK, N))
pr = bench.build_processor(B, True)
pr.train_samples, pr.num_train_samples = samples, N
for prefetch in (True, False):
    pr.args.prefetch_batches = prefetch
    for ep in range(2):                                  # epoch 0 warms up (graph capture, pinned allocations)
        np.random.seed(ep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for text_b, vec, audio, mfcc, vids in pr.yield_batch(train=True):
            pr.train_step(text_b, audio, mfcc, vec, vids, sync=False)
            n += B
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f'prefetch={prefetch}: {n} clips in {dt * 1e3:.1f} ms = {n / dt:.0f} clips/s (host-resident dataset, '
          f'PCIe + decode inside the timed loop)', flush=True)
