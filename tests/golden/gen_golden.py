"""Generate the golden fixtures in this directory by running the REFERENCE itself on CPU.

Runs ONLY in the build container (needs /root/reference).  Nothing here travels to the GPU box except the
``*.npz`` outputs.  Recipe: SURVEY.md Appendix A (stub absent third-party modules, neutralise the
hard-coded ``.cuda()`` calls, bypass ``Processor.__init__``).

Weights and the large inputs are NOT stored: they are regenerated from the frozen legacy-RandomState
recipe in ``oracle/s2ag_oracle.py`` (``recipe_state_dict`` / ``recipe_inputs``), which only draws
numbers -- it restates nothing of the reference.  Loading the recipe dict with ``strict=True`` into the
reference modules also proves that the oracle's key/shape tables equal the reference's state_dict.

    python tests/golden/gen_golden.py          # rewrites tests/golden/*.npz
"""
import importlib.machinery
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
sys.path.insert(1, ROOT)


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith('__'):
            raise AttributeError(k)
        m = _Stub(self.__name__ + '.' + k)
        m.__path__ = []
        m.__spec__ = importlib.machinery.ModuleSpec(m.__name__, None)
        sys.modules[m.__name__] = m
        setattr(self, k, m)
        return m

    def __call__(self, *a, **k):
        return _Stub('call')


for name in ['librosa', 'librosa.feature', 'librosa.display', 'fasttext', 'lmdb', 'python_speech_features',
             'h5py', 'transforms3d', 'umap', 'soundfile', 'pyttsx3']:
    m = _Stub(name)
    m.__path__ = []
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules[name] = m

import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self            # AffEncoder.__init__ hard-codes .cuda()
torch.set_num_threads(8)

import net.multimodal_context_net_v2 as m2                 # noqa: E402
import net.multimodal_context_net_v2_abl_audio as m2a      # noqa: E402
import net.embedding_net as en                             # noqa: E402
import processor_v2 as P                                   # noqa: E402
from utils.vocab import Vocab                              # noqa: E402

from oracle import s2ag_oracle as O                        # noqa: E402  (recipes only)
sys.path.insert(2, HERE)
import s2ag_rng                                            # noqa: E402

# the product's step consumes one noise pass per forward, in this order (processor_v2.py of the product):
# 0 G(dis) 1 D(real) 2 D(fake) 3 PGT 4 G(main) 5 D(gen) 6 G(rand); the tests pin these ids on the modules
STEP_SEED, G_Z_SITE, PGT_Z_SITE, PASSES_PER_STEP = 1234, 9001, 9002, 7


class Cfg:
    pass


def make_cfg(hidden, drop):
    c = Cfg()
    c.n_pre_poses = 4
    c.n_poses = 34
    c.input_context = 'both'
    c.hidden_size = hidden
    c.hidden_size_s2eg = hidden
    c.n_layers = 4
    c.dropout_prob = drop
    c.freeze_wordembed = False
    c.loss_warmup = 0
    c.loss_gan_weight = 5.0
    c.z_type = 'speaker'
    c.loss_reg_weight = 0.05
    c.loss_regression_weight = 500
    c.loss_kld_weight = 0.1
    return c


def speakers(n):
    spk = Vocab('vid', insert_default_tokens=False)
    for i in range(n - 1):
        spk.index_word('v%d' % i)
    assert spk.n_words == n
    return spk


def ocfg(hidden, drop):
    return O.ModelCfg(hidden_size=hidden, hidden_size_s2eg=hidden, dropout_prob=drop)


def build(hidden, n_words, n_spk, drop, seed0):
    """Reference modules with recipe weights (strict load = key/shape contract check)."""
    cfg = make_cfg(hidden, drop)
    spk = speakers(n_spk)
    G = m2.PoseGenerator(cfg, 27, n_words, 300, None, 71, 37, 34, z_obj=spk)
    D = m2.AffDiscriminator(27)
    CD = m2.ConvDiscriminatorTriModal(27)
    T3 = m2.PoseGeneratorTriModal(cfg, 27, n_words, 300, None, z_obj=spk)
    GA = m2a.PoseGenerator(cfg, 27, n_words, 300, None, 71, 37, 34, z_obj=spk)
    oc = ocfg(hidden, drop)
    sds = dict(G=O.recipe_state_dict(O.generator_shapes(oc, n_words, n_spk), seed0 + 1),
               D=O.recipe_state_dict(O.aff_discriminator_shapes(), seed0 + 2),
               CD=O.recipe_state_dict(O.conv_discriminator_shapes(), seed0 + 3),
               T3=O.recipe_state_dict(O.trimodal_shapes(oc, n_words, n_spk), seed0 + 4),
               GA=O.recipe_state_dict(O.generator_shapes(oc, n_words, n_spk, audio='wav'), seed0 + 5))
    for mod, key in ((G, 'G'), (D, 'D'), (CD, 'CD'), (T3, 'T3'), (GA, 'GA')):
        ref_sd = mod.state_dict()
        assert set(ref_sd) == set(sds[key]), (key, set(ref_sd) ^ set(sds[key]))
        for k in ref_sd:
            assert tuple(ref_sd[k].shape) == tuple(sds[key][k].shape), (key, k)
        mod.load_state_dict({k: v.clone() for k, v in sds[key].items()}, strict=True)
    return cfg, dict(G=G, D=D, CD=CD, T3=T3, GA=GA)


def pin_eps(eps_list):
    """en.re_parametrize draws randn_like (embedding_net.py:10-13); replace by a queue of fixtures."""
    q = list(eps_list)

    def f(mu, log_var):
        e = q.pop(0)
        return mu + e * torch.exp(0.5 * log_var)
    en.re_parametrize = f
    return q


def npy(t):
    return t.detach().cpu().numpy().copy()


def module_goldens(tag, hidden, n_words, n_spk, B, seed0):
    """Eval-mode and train-mode(dropout off) forwards of every module on the path."""
    out = {}
    cfg, mods = build(hidden, n_words, n_spk, 0.0, seed0)
    inp = O.recipe_inputs(B, 34, seed0 + 10, n_words, n_spk)
    # eps as libs2ag_hip.so draws it for (seed, pass counter 0, site G_Z_SITE): the GPU tests reset the noise
    # state before every module forward and pin z_site, so product and reference see the same deviates
    eps = torch.from_numpy(s2ag_rng.normal(STEP_SEED, 0, G_Z_SITE, B * 16).reshape(B, 16))
    pre_seq = O.make_pre_seq(inp['target'], 4)
    out['eps'] = npy(eps)
    for mode in ('eval', 'train'):
        _, mods = build(hidden, n_words, n_spk, 0.0, seed0)     # fresh running stats per mode
        for m in mods.values():
            m.train(mode == 'train')
            if mode == 'train':
                for sub in m.modules():
                    if isinstance(sub, torch.nn.Dropout):
                        sub.p = 0.0
                    if isinstance(sub, torch.nn.GRU):
                        sub.dropout = 0.0
        G, D, CD, T3, GA = (mods[k] for k in ('G', 'D', 'CD', 'T3', 'GA'))
        with torch.no_grad():
            out[f'{mode}.wav_encoder'] = npy(T3.audio_encoder(inp['in_audio']))
            out[f'{mode}.mfcc_encoder'] = npy(G.audio_encoder(inp['in_mfcc']))
            out[f'{mode}.text_encoder'] = npy(G.text_encoder(inp['in_text'])[0])
            out[f'{mode}.aff_encoder'] = npy(G.aff_encoder(inp['target']))
        # the full nets see fresh copies so BN running stats are those of exactly one forward
        _, mods = build(hidden, n_words, n_spk, 0.0, seed0)
        for m in mods.values():
            m.train(mode == 'train')
            if mode == 'train':
                for sub in m.modules():
                    if isinstance(sub, torch.nn.Dropout):
                        sub.p = 0.0
                    if isinstance(sub, torch.nn.GRU):
                        sub.dropout = 0.0
        G, D, CD, T3, GA = (mods[k] for k in ('G', 'D', 'CD', 'T3', 'GA'))
        with torch.no_grad():
            pin_eps([eps, eps, eps])
            o, z, mu, lv = G(pre_seq, inp['in_text'], inp['in_mfcc'], inp['vid'])
            out[f'{mode}.G.out'], out[f'{mode}.G.z'] = npy(o), npy(z)
            out[f'{mode}.G.mu'], out[f'{mode}.G.log_var'] = npy(mu), npy(lv)
            out[f'{mode}.T3.out'] = npy(T3(pre_seq, inp['in_text'], inp['in_audio'], inp['vid'])[0])
            out[f'{mode}.GA.out'] = npy(GA(pre_seq, inp['in_text'], inp['in_audio'], inp['vid'])[0])
            out[f'{mode}.D.out'] = npy(D(inp['target']))
            out[f'{mode}.CD.out'] = npy(CD(inp['target']))
        if mode == 'train':
            sdG = G.state_dict()
            for k in ('audio_encoder.batch_norm1.running_mean', 'audio_encoder.batch_norm1.running_var',
                      'aff_encoder.st_gcn1.tcn.0.running_var', 'aff_encoder.batch_norm1.running_mean',
                      'aff_encoder.batch_norm4.running_var', 'aff_encoder.batch_norm2.num_batches_tracked'):
                out['train.G.' + k] = npy(sdG[k])
            out['train.T3.audio_encoder.feat_extractor.1.running_var'] = \
                npy(T3.state_dict()['audio_encoder.feat_extractor.1.running_var'])
    np.savez_compressed(os.path.join(HERE, f'modules_{tag}.npz'), **out)
    print('wrote modules_%s.npz' % tag, {k: v.shape for k, v in list(out.items())[:4]})


def tcn_dropout_golden(seed0):
    """TextEncoderTCN in train mode with PINNED dropout masks (F.dropout replaced by a mask queue)."""
    hidden, n_words, n_spk, B = 32, 64, 12, 2
    cfg, mods = build(hidden, n_words, n_spk, 0.3, seed0)
    G = mods['G']
    G.train()
    inp = O.recipe_inputs(B, 34, seed0 + 10, n_words, n_spk)
    rs = np.random.RandomState(seed0 + 12)
    masks = {}
    order = []

    def fake_dropout(x, p=0.5, training=True, inplace=False):
        if not training or p == 0:
            return x
        name = 'm%d' % len(order)
        m = torch.from_numpy((rs.uniform(size=tuple(x.shape)) >= p).astype(np.float32) / (1 - p))
        masks[name] = m
        order.append((name, p))
        return x * m
    real = torch.nn.functional.dropout
    torch.nn.functional.dropout = fake_dropout
    try:
        with torch.no_grad():
            y = G.text_encoder(inp['in_text'])[0]
    finally:
        torch.nn.functional.dropout = real
    out = {'y': npy(y)}
    for (name, p) in order:
        out[name] = npy(masks[name])
    out['ps'] = np.array([p for _, p in order], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, 'tcn_dropout.npz'), **out)
    print('wrote tcn_dropout.npz', len(order), 'masks')


# the branches of the step the reference's configuration selects (processor_v2.py:793, :899-934, :936): what each variant
# changes in the config, the file it writes, and the noise passes (counter offset inside the step, site) whose eps the
# reference's re_parametrize calls consume, in call order -- the product numbers its passes 0 G(dis) 1 D(real) 2 D(fake) 3 PGT
# 4 G(main) 5 D(gen) 6 G(rand) with the GAN phase on, and 0 PGT 1 G(main) 2 D(gen) 3 G(rand) during warm-up (no D phase)
STEP_VARIANTS = {
    'speaker': (dict(), 'step_small.npz', 7, ((0, G_Z_SITE), (3, PGT_Z_SITE), (4, G_Z_SITE), (6, G_Z_SITE))),
    'znone': (dict(z_type='none'), 'step_small_znone.npz', 7, ((0, G_Z_SITE), (3, PGT_Z_SITE), (4, G_Z_SITE))),
    # (the same branch by the other route: the reference's trace is IDENTICAL to 'znone' array for array -- step_golden checks
    #  that instead of storing it twice)
    'noreg': (dict(loss_reg_weight=0.0), 'step_small_znone.npz', 7, ((0, G_Z_SITE), (3, PGT_Z_SITE), (4, G_Z_SITE))),
    'warmup': (dict(loss_warmup=5), 'step_small_warmup.npz', 4, ((0, PGT_Z_SITE), (1, G_Z_SITE), (3, G_Z_SITE))),
}


def step_golden(seed0, n_steps=3, variant='speaker'):
    """Three reference GAN steps (Processor.forward_pass_s2ag) at reduced width, dropout off, noise pinned."""
    hidden, n_words, n_spk, B = 32, 64, 12, 4
    overrides, fname, passes_per_step, eps_passes = STEP_VARIANTS[variant]
    cfg, mods = build(hidden, n_words, n_spk, 0.0, seed0)
    for k, v in overrides.items():
        assert hasattr(cfg, k), k
        setattr(cfg, k, v)
    G, D, T3 = mods['G'], mods['D'], mods['T3']
    for m in (G, D, T3):
        m.train()
        for sub in m.modules():
            if isinstance(sub, torch.nn.Dropout):
                sub.p = 0.0
            if isinstance(sub, torch.nn.GRU):
                sub.dropout = 0.0
    pr = object.__new__(P.Processor)
    pr.s2ag_config_args = cfg
    pr.meta_info = dict(epoch=1, iter=0)
    pr.use_mfcc = True
    pr.s2ag_generator, pr.s2ag_discriminator, pr.trimodal_generator = G, D, T3
    pr.s2ag_gen_optimizer = torch.optim.Adam(G.parameters(), lr=5e-4, betas=(0.5, 0.999))
    pr.s2ag_dis_optimizer = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.999))

    rs = np.random.RandomState(seed0 + 13)
    out = {}
    recorded = []
    real_backward = torch.Tensor.backward

    def rec_backward(self, *a, **k):
        recorded.append(float(self))
        return real_backward(self, *a, **k)
    torch.Tensor.backward = rec_backward
    real_randperm = torch.randperm
    try:
        for s in range(n_steps):
            inp = O.recipe_inputs(B, 34, seed0 + 100 + s, n_words, n_spk)
            # eps exactly as libs2ag_hip.so will draw them: (seed, pass counter, site) -> normal deviates
            eps = [torch.from_numpy(s2ag_rng.normal(STEP_SEED, passes_per_step * s + k, site, B * 16).reshape(B, 16))
                   for k, site in eps_passes]
            perm = torch.from_numpy(rs.permutation(B))
            out[f's{s}.eps'] = np.stack([npy(e) for e in eps])     # order: G(dis), PGT, G(main), G(rand) -- those that run
            out[f's{s}.perm'] = npy(perm)
            queue = pin_eps(eps)
            torch.randperm = lambda n, *a, **k: perm
            recorded.clear()
            ret = pr.forward_pass_s2ag(inp['in_text'], inp['in_audio'], inp['in_mfcc'], inp['target'],
                                       inp['vid'], train=True)
            assert not queue, (variant, 'eps left over', len(queue))   # every pinned deviate was consumed: the pass list is right
            out[f's{s}.metric'] = np.float64(ret[0])
            if variant == 'warmup':                                   # no D phase: one backward() per step
                assert len(recorded) == 1
                out[f's{s}.loss'] = np.float64(recorded[0])
            else:
                assert len(recorded) == 2
                out[f's{s}.dis_error'] = np.float64(recorded[0])
                out[f's{s}.loss'] = np.float64(recorded[1])
            if s == 0:
                for k, p in G.named_parameters():
                    if k in ('out.2.weight', 'gru.weight_hh_l3_reverse', 'aff_encoder.st_gcn1.gcn.conv.weight',
                             'text_encoder.tcn.network.0.conv1.weight_v', 'audio_encoder.conv1.weight',
                             'speaker_log_var.weight', 'text_encoder.tcn.network.3.conv2.weight_g'):
                        out['s0.grad.G.' + k] = npy(p.grad)
            for tagm, mod in (('G', G), ('D', D)):
                sd = mod.state_dict()
                groups = {}
                for k, v in sd.items():
                    if '.net.' in k or k.endswith('num_batches_tracked'):
                        continue
                    top = k.split('.')[0]
                    groups[top] = groups.get(top, 0.0) + float(v.double().abs().sum())
                for top, val in groups.items():
                    out[f's{s}.abs.{tagm}.{top}'] = np.float64(val)
    finally:
        torch.Tensor.backward = real_backward
        torch.randperm = real_randperm
    sdG, sdD = G.state_dict(), D.state_dict()
    for k in ('out.2.weight', 'gru.bias_hh_l0', 'aff_encoder.batch_norm1.running_mean',
              'aff_encoder.conv4.weight', 'speaker_mu.weight', 'audio_encoder.batch_norm4.running_var'):
        out['final.G.' + k] = npy(sdG[k])
    for k in ('out2.weight', 'gru.weight_ih_l0', 'aff_encoder.st_gcn2.tcn.2.weight'):
        out['final.D.' + k] = npy(sdD[k])
    if variant == 'noreg':          # must reproduce the 'znone' fixture exactly (same code path of processor_v2.py:933-934)
        ref = dict(np.load(os.path.join(HERE, fname)))
        assert set(ref) == set(out) and all(np.array_equal(np.asarray(out[k]), ref[k]) for k in ref), 'noreg != znone'
        print('noreg: identical to', fname)
        return
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print('wrote', fname, [(k, float(out[k])) for k in out if k.endswith('.loss') or k.endswith('.dis_error')])


def misc_goldens():
    """Graph adjacency, get_epoch_and_loss parse results (processor_v2.py:53-83)."""
    import tempfile
    out = {}
    enc = m2.AffEncoder()
    out['A1'], out['A2'] = npy(enc.A1), npy(enc.A2)
    with tempfile.TemporaryDirectory() as d:
        names = ['epoch_000010_loss_0.5000_model.pth.tar', 'epoch_000020_loss_0.2500_model.pth.tar',
                 'epoch_000030_loss_0.1250_model.pth.tar', 'epoch_000040_loss_0.3000_model.pth.tar']
        for n in names:
            open(os.path.join(d, n), 'w').close()
        # os.listdir order is filesystem dependent; pin it so the fixture is reproducible
        real = os.listdir
        P.os.listdir = lambda p: list(names)
        try:
            best = P.get_epoch_and_loss(d, 'best')
            at20 = P.get_epoch_and_loss(d, 20)
            missing = P.get_epoch_and_loss(d, 7)
        finally:
            P.os.listdir = real
    out['ckpt_names'] = np.array(names)
    out['best'] = np.array([best[0], str(best[1]), repr(best[2])])
    out['at20'] = np.array([at20[0], str(at20[1]), repr(at20[2])])
    out['missing'] = np.array([missing[0], str(missing[1]), repr(missing[2])])
    np.savez_compressed(os.path.join(HERE, 'misc.npz'), **out)
    print('wrote misc.npz', best, at20, missing)


if __name__ == '__main__':
    misc_goldens()
    module_goldens('small', hidden=32, n_words=64, n_spk=12, B=2, seed0=1000)
    module_goldens('full', hidden=300, n_words=2000, n_spk=1371, B=4, seed0=2000)
    tcn_dropout_golden(3000)
    for v in STEP_VARIANTS:
        step_golden(4000, variant=v)
