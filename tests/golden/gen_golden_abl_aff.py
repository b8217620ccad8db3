"""Golden fixture for the affective-encoder ablation: runs the REFERENCE's ``net.multimodal_context_net_v2_abl_aff``
(``PoseGenerator`` :285-392 trained against ``ConvDiscriminator`` :394-439) on CPU with recipe weights: eval- and
train-mode (dropout off) forwards at two widths, and the gradients of ``(out * d_out).sum() + D(out).log().mean()``
with respect to a few generator tensors.  Build container only; see gen_golden.py for the import recipe.

    python tests/golden/gen_golden_abl_aff.py          # rewrites tests/golden/abl_aff.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (sets up the stubs and imports the reference)

torch, O, s2ag_rng = gg.torch, gg.O, gg.s2ag_rng
import net.multimodal_context_net_v2_abl_aff as m2f  # noqa: E402

GRAD_KEYS = ('out.2.weight', 'gru.weight_ih_l0', 'gru.weight_hh_l2_reverse', 'text_encoder.tcn.network.1.conv1.weight_v',
             'audio_encoder.conv2.weight', 'speaker_mu.weight')


GRAD_KEYS_FULL = ('out.2.weight', 'gru.bias_hh_l0', 'audio_encoder.conv2.weight', 'speaker_mu.weight',
                  'text_encoder.tcn.network.1.conv1.weight_g')


def build(hidden, n_words, n_spk, seed0):
    cfg = gg.make_cfg(hidden, 0.0)
    spk = gg.speakers(n_spk)
    G = m2f.PoseGenerator(cfg, 27, n_words, 300, None, 71, 37, 34, z_obj=spk)
    D = m2f.ConvDiscriminator(27)
    oc = gg.ocfg(hidden, 0.0)
    sds = dict(GF=O.recipe_state_dict(O.generator_shapes(oc, n_words, n_spk, aff=False), seed0 + 6),
               CD=O.recipe_state_dict(O.conv_discriminator_shapes(), seed0 + 3))
    for mod, key in ((G, 'GF'), (D, 'CD')):
        ref_sd = mod.state_dict()
        assert set(ref_sd) == set(sds[key]), (key, set(ref_sd) ^ set(sds[key]))
        for k in ref_sd:
            assert tuple(ref_sd[k].shape) == tuple(sds[key][k].shape), (key, k)
        mod.load_state_dict({k: v.clone() for k, v in sds[key].items()}, strict=True)
    return G, D


def main():
    out = {}
    for tag, hidden, n_words, n_spk, B, seed0 in (('small', 32, 64, 12, 2, 1000), ('full', 300, 2000, 1371, 4, 2000)):
        inp = O.recipe_inputs(B, 34, seed0 + 10, n_words, n_spk)
        eps = torch.from_numpy(s2ag_rng.normal(gg.STEP_SEED, 0, gg.G_Z_SITE, B * 16).reshape(B, 16))
        pre_seq = O.make_pre_seq(inp['target'], 4)
        d_out = torch.from_numpy(np.random.RandomState(seed0 + 77).standard_normal((B, 34, 27)).astype(np.float32))
        out[f'{tag}.d_out'] = gg.npy(d_out)
        for mode in ('eval', 'train'):
            G, D = build(hidden, n_words, n_spk, seed0)
            for m in (G, D):
                m.train(mode == 'train')
                for sub in m.modules():
                    if isinstance(sub, torch.nn.Dropout):
                        sub.p = 0.0
                    if isinstance(sub, torch.nn.GRU):
                        sub.dropout = 0.0
            gg.pin_eps([eps])
            o, z, mu, lv = G(pre_seq, inp['in_text'], inp['in_mfcc'], inp['vid'])
            d = D(o)
            out[f'{tag}.{mode}.out'], out[f'{tag}.{mode}.z'] = gg.npy(o), gg.npy(z)
            out[f'{tag}.{mode}.mu'], out[f'{tag}.{mode}.d'] = gg.npy(mu), gg.npy(d)
            if mode == 'train':
                ((o * d_out).sum() + d.log().mean()).backward()
                named = dict(G.named_parameters())
                for k in (GRAD_KEYS if tag == 'small' else GRAD_KEYS_FULL):      # keep the fixture small
                    out[f'{tag}.grad.{k}'] = gg.npy(named[k].grad)
                out[f'{tag}.train.bn_rv'] = gg.npy(G.state_dict()['audio_encoder.batch_norm2.running_var'])
    path = os.path.join(HERE, 'abl_aff.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, sorted(out)[:6], len(out))


if __name__ == '__main__':
    main()
