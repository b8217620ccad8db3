"""BASELINE configs[3] (the "Conv1d roofline run": WavEncoder + TextEncoderTCN fwd+bwd, B=256) on its own, for profiling."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

r = bench.conv1d_roofline_run('cuda', B=int(os.environ.get('B', 256)), iters=int(os.environ.get('ITERS', 30)), cpu=False,
                              mode=os.environ.get('MODE', 'fp32'))
print(r)
