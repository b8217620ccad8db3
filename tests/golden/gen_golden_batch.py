"""Golden fixture for the batch draw: runs the REFERENCE's own ``Processor.yield_batch`` (processor_v2.py:589-638) on
CPU under a fixed ``np.random.seed`` over the tiny sample dict of ``batch_recipe.py`` and records, per batch, the
tensors it yields (words, poses, decoded audio, MFCCs, "other speaker" ids) for one train and one val pseudo-epoch.
Build container only (needs /root/reference); see gen_golden.py for the import recipe.

    python tests/golden/gen_golden_batch.py          # rewrites tests/golden/batch_small.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (sets up the stubs and imports the reference)

torch, P = gg.torch, gg.P

import batch_recipe as R  # noqa: E402


def run(train):
    pr = object.__new__(P.Processor)
    s = R.samples()
    pr.device = torch.device('cpu')
    pr.args = types.SimpleNamespace(batch_size=R.BATCH)
    pr.data_loader = {'train_data_s2ag': None, 'val_data_s2ag': None}
    pr.train_samples = pr.val_samples = s
    pr.num_train_samples = pr.num_val_samples = R.N_DATA
    pr.train_speaker_model = pr.val_speaker_model = R.Vocab(R.N_SPK)
    np.random.seed(R.SEED + int(train))
    out = []
    for batch in pr.yield_batch(train):
        out.append([t.numpy().copy() for t in batch])
    return out


def main():
    rec = {}
    for train in (True, False):
        tag = 'train' if train else 'val'
        batches = run(train)
        rec[tag + '.n'] = np.int64(len(batches))
        for name, k in (('text', 0), ('vec', 1), ('audio', 2), ('mfcc', 3), ('vids', 4)):
            rec[f'{tag}.{name}'] = np.stack([b[k] for b in batches])
    path = os.path.join(HERE, 'batch_small.npz')
    np.savez_compressed(path, **rec)
    print('wrote', path, {k: (v.shape, v.dtype) for k, v in rec.items()})


if __name__ == '__main__':
    main()
