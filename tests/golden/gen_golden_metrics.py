"""Golden fixture for forward_pass_s2ag(calculate_metrics=True): runs the REFERENCE's own ``Processor.push_samples``
(processor_v2.py:738-774: F.l1_loss, convert_dir_vec_to_pose on dir + mean_dir_vec, joint MAE behind the seed poses,
acceleration difference) with its ``AverageMeter`` (utils/average_meter.py) over three batches of generated / target
direction vectors.  Build container only; see gen_golden.py for the import recipe.

    python tests/golden/gen_golden_metrics.py      # rewrites tests/golden/metrics.npz

Inputs are not stored: ``metrics_inputs`` below regenerates them from legacy RandomState seeds."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from metrics_recipe import MEAN_DIR_VEC, N_BATCHES, N_PRE, metrics_inputs  # noqa: E402


def main():
    import gen_golden as gg  # noqa: E402  (sets up the stubs and imports the reference)
    from utils.average_meter import AverageMeter
    P, torch = gg.P, gg.torch
    meters = [AverageMeter('loss'), AverageMeter('mae_joint'), AverageMeter('accel')]
    vals = []
    for b in range(N_BATCHES):
        out, tgt = metrics_inputs(b)
        # on CPU tensors upstream's `.cpu().numpy(); += mean` would write through into the caller's tensors: hand it copies
        _, *meters = P.Processor.push_samples(None, torch.from_numpy(tgt.copy()), torch.from_numpy(out.copy()), None, None,
                                              *meters, MEAN_DIR_VEC, out.shape[1], N_PRE)
        vals.append([m.val for m in meters])
    res = dict(vals=np.array(vals, dtype=np.float64), avgs=np.array([m.avg for m in meters], dtype=np.float64),
               counts=np.array([m.count for m in meters], dtype=np.int64))
    path = os.path.join(HERE, 'metrics.npz')
    np.savez_compressed(path, **res)
    print('wrote', path, res['vals'], res['avgs'])


if __name__ == '__main__':
    main()
