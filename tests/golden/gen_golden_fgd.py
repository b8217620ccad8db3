"""Golden fixture for the evaluation metrics: runs the REFERENCE's ``EmbeddingNet`` in 'pose' mode
(net/embedding_net.py:262-308) with recipe weights in eval and train mode, and its ``EmbeddingSpaceEvaluator``
(net/embedding_space_evaluator.py:46-103: push_samples / get_scores) over three batches of real / "generated" poses.
Build container only; see gen_golden.py for the import recipe.

    python tests/golden/gen_golden_fgd.py          # rewrites tests/golden/fgd.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (sets up the stubs and imports the reference)

torch, O, en = gg.torch, gg.O, gg.en
from net.embedding_space_evaluator import EmbeddingSpaceEvaluator  # noqa: E402

SEED, B, NB = 6100, 24, 3


def poses(seed, n):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.standard_normal((n, 34, 27)) * 0.2).astype(np.float32))


def main():
    out = {}
    sd = O.recipe_state_dict(O.embedding_net_shapes(), SEED, scale=3.0, tcn_aliases=False)
    for mode in ('eval', 'train'):
        net = en.EmbeddingNet(None, 27, 34, 10, 300, None, 'pose')
        assert set(net.state_dict()) == set(sd), set(net.state_dict()) ^ set(sd)
        net.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
        net.train(mode == 'train')
        x = poses(SEED + 1, B)
        with torch.no_grad():
            _, _, _, feat, mu, lv, rec = net(None, None, None, x, 'pose', variational_encoding=False)
        out[f'{mode}.feat'], out[f'{mode}.log_var'], out[f'{mode}.recon'] = gg.npy(feat), gg.npy(lv), gg.npy(rec)
        if mode == 'train':
            out['train.rv'] = gg.npy(net.state_dict()['pose_encoder.out_net.1.running_var'])
    # the evaluator: bypass __init__ (it loads outputs/embedding_net.pth.tar)
    ev = object.__new__(EmbeddingSpaceEvaluator)
    ev.n_pre_poses, ev.pose_dim = 4, 27
    ev.net = en.EmbeddingNet(None, 27, 34, 10, 300, None, 'pose')
    ev.net.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    ev.net.train(False)
    ev.reset()
    with torch.no_grad():
        for b in range(NB):
            real = poses(SEED + 10 + b, B)
            gen = real * 0.5 + poses(SEED + 20 + b, B) * 1.5 + 0.2
            ev.push_samples(None, None, gen, real)
    fd, feat_dist = ev.get_scores()
    out['scores'] = np.array([fd, feat_dist], dtype=np.float64)
    out['recon_err_diff'] = np.array(ev.recon_err_diff, dtype=np.float64)
    path = os.path.join(HERE, 'fgd.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, out['scores'], out['recon_err_diff'])


if __name__ == '__main__':
    main()
