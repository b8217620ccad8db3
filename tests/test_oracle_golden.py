"""Pins oracle/s2ag_oracle.py against the fixtures generated from the reference itself
(tests/golden/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import s2ag_oracle as O

torch.set_num_threads(8)


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name), allow_pickle=False))


def _close(a, b, tol=2e-5):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b.detach().numpy() if torch.is_tensor(b) else b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() / max(1e-6, np.abs(a).max())
    assert err < tol, err


def test_adjacency_matches_reference(golden_dir):
    g = _load(golden_dir, 'misc.npz')
    A1, A2 = O.aff_adjacencies()
    np.testing.assert_allclose(A1.numpy(), g['A1'], atol=1e-7)
    np.testing.assert_allclose(A2.numpy(), g['A2'], atol=1e-7)
    assert A1.shape == (5, 9, 9) and A2.shape == (5, 3, 3)
    assert float(A2[4].abs().sum()) == 0.0          # 5th slice all zero


CASES = {'small': dict(hidden=32, n_words=64, n_spk=12, B=2, seed0=1000),
         'full': dict(hidden=300, n_words=2000, n_spk=1371, B=4, seed0=2000)}


def _models(c):
    oc = O.ModelCfg(hidden_size=c['hidden'], hidden_size_s2eg=c['hidden'], dropout_prob=0.0)
    s0 = c['seed0']
    return oc, dict(G=O.recipe_state_dict(O.generator_shapes(oc, c['n_words'], c['n_spk']), s0 + 1),
                    D=O.recipe_state_dict(O.aff_discriminator_shapes(), s0 + 2),
                    CD=O.recipe_state_dict(O.conv_discriminator_shapes(), s0 + 3),
                    T3=O.recipe_state_dict(O.trimodal_shapes(oc, c['n_words'], c['n_spk']), s0 + 4),
                    GA=O.recipe_state_dict(O.generator_shapes(oc, c['n_words'], c['n_spk'], audio='wav'), s0 + 5))


@pytest.mark.parametrize('tag', ['small', 'full'])
@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_modules_match_reference(golden_dir, tag, mode):
    c = CASES[tag]
    g = _load(golden_dir, f'modules_{tag}.npz')
    inp = O.recipe_inputs(c['B'], 34, c['seed0'] + 10, c['n_words'], c['n_spk'])
    pre_seq = O.make_pre_seq(inp['target'], 4)
    eps = torch.from_numpy(g['eps'])
    tr = mode == 'train'
    with torch.no_grad():
        oc, m = _models(c)
        _close(g[f'{mode}.wav_encoder'], O.wav_encoder(m['T3'], 'audio_encoder.', inp['in_audio'], tr))
        _close(g[f'{mode}.mfcc_encoder'], O.mfcc_encoder(m['G'], 'audio_encoder.', inp['in_mfcc'], tr))
        _close(g[f'{mode}.text_encoder'],
               O.text_encoder_tcn(m['G'], 'text_encoder.', inp['in_text'], tr, 0.0, O.Noise('off'), 0.0))
        _close(g[f'{mode}.aff_encoder'], O.aff_encoder(m['G'], 'aff_encoder.', inp['target'], tr))
        oc, m = _models(c)

        def nz():
            return O.Noise({'eps': eps}) if not tr else _TrainNoise(eps)
        o, z, mu, lv = O.pose_generator(m['G'], oc, pre_seq, inp['in_text'], inp['in_mfcc'], inp['vid'], tr, nz())
        _close(g[f'{mode}.G.out'], o)
        _close(g[f'{mode}.G.z'], z)
        _close(g[f'{mode}.G.mu'], mu)
        _close(g[f'{mode}.G.log_var'], lv)
        _close(g[f'{mode}.T3.out'], O.pose_generator_trimodal(m['T3'], oc, pre_seq, inp['in_text'], inp['in_audio'],
                                                              inp['vid'], tr, nz())[0])
        _close(g[f'{mode}.GA.out'], O.pose_generator_abl_audio(m['GA'], oc, pre_seq, inp['in_text'], inp['in_audio'],
                                                               inp['vid'], tr, nz())[0])
        _close(g[f'{mode}.D.out'], O.aff_discriminator(m['D'], inp['target'], tr, O.Noise('off')))
        _close(g[f'{mode}.CD.out'], O.conv_discriminator(m['CD'], inp['target'], tr, O.Noise('off')))
        if tr:
            for k in g:
                if k.startswith('train.G.') and k != 'train.G.out' and k[8:] in m['G']:
                    _close(g[k], m['G'][k[8:]].float())
            _close(g['train.T3.audio_encoder.feat_extractor.1.running_var'],
                   m['T3']['audio_encoder.feat_extractor.1.running_var'])


class _TrainNoise(O.Noise):
    """dropout disabled (the golden run sets every p to 0) but eps pinned."""

    def __init__(self, eps):
        super().__init__({'eps': eps})

    def dropout(self, name, x, p):
        return x


def test_tcn_pinned_dropout_matches_reference(golden_dir):
    g = _load(golden_dir, 'tcn_dropout.npz')
    c = CASES['small']
    oc = O.ModelCfg(hidden_size=32, hidden_size_s2eg=32, dropout_prob=0.3)
    G = O.recipe_state_dict(O.generator_shapes(oc, c['n_words'], c['n_spk']), 3000 + 1)
    inp = O.recipe_inputs(2, 34, 3000 + 10, c['n_words'], c['n_spk'])
    names = ['text_encoder.emb_drop'] + [f'text_encoder.tcn.{i}.drop{j}' for i in range(4) for j in (1, 2)]
    pinned = {n: torch.from_numpy(g['m%d' % i]) for i, n in enumerate(names)}
    assert list(g['ps']) == pytest.approx([0.1] + [0.3] * 8)
    with torch.no_grad():
        y = O.text_encoder_tcn(G, 'text_encoder.', inp['in_text'], True, 0.3, O.Noise(pinned), 0.1)
    _close(g['y'], y)


def test_gru_fast_equals_cellwise():
    torch.manual_seed(0)
    sd = {k: torch.randn(s) * 0.2 for k, s in O._gru_shapes('gru.', 11, 24, 3).items()}
    x = torch.randn(3, 7, 11)
    a = O.gru(sd, 'gru.', x, False, 0.0, O.Noise('off'), 'gru')
    b = O.gru_fast(sd, 'gru.', x, False, 0.0)
    assert (a - b).abs().max() < 1e-5


# the branches of the step (tests/golden/gen_golden.py STEP_VARIANTS: traces recorded from the reference's own
# forward_pass_s2ag with the config changed as named): file, StepCfg overrides, eps rows -> passes
_STEP_VARIANTS = {
    'speaker': ('step_small.npz', {}, ('g_dis', 'pgt', 'g_main', 'g_rand')),
    'znone': ('step_small_znone.npz', {'z_type': 'none'}, ('g_dis', 'pgt', 'g_main')),
    'noreg': ('step_small_znone.npz', {'loss_reg_weight': 0.0}, ('g_dis', 'pgt', 'g_main')),
    'warmup': ('step_small_warmup.npz', {'loss_warmup': 5}, ('pgt', 'g_main', 'g_rand')),
}


class _NoPass:
    """A pass the branch must not run: any noise request fails the test."""
    def __getattr__(self, k):
        raise AssertionError('this pass does not exist in this branch of the step (processor_v2.py:793, :899-934)')


@pytest.mark.parametrize('variant', list(_STEP_VARIANTS))
def test_three_step_trace_matches_reference(golden_dir, variant):
    fname, overrides, eps_passes = _STEP_VARIANTS[variant]
    g = _load(golden_dir, fname)
    hidden, n_words, n_spk, B, s0 = 32, 64, 12, 4, 4000
    oc = O.ModelCfg(hidden_size=hidden, hidden_size_s2eg=hidden, dropout_prob=0.0)
    G = O.recipe_state_dict(O.generator_shapes(oc, n_words, n_spk), s0 + 1)
    D = O.recipe_state_dict(O.aff_discriminator_shapes(), s0 + 2)
    T3 = O.recipe_state_dict(O.trimodal_shapes(oc, n_words, n_spk), s0 + 4)
    gopt, dopt = O.AdamState(), O.AdamState()
    scfg = O.StepCfg(**overrides)
    for s in range(3):
        inp = O.recipe_inputs(B, 34, s0 + 100 + s, n_words, n_spk)
        eps = torch.from_numpy(g[f's{s}.eps'])
        assert eps.shape[0] == len(eps_passes)
        by_pass = {name: _TrainNoise(eps[i]) for i, name in enumerate(eps_passes)}
        gen_pass = lambda name: by_pass.get(name, _NoPass())                       # noqa: E731
        dis_pass = O.Noise('off') if variant != 'warmup' else _NoPass()
        nz = O.StepNoise(g_dis=gen_pass('g_dis'), d_real=dis_pass, d_fake=dis_pass, pgt=gen_pass('pgt'),
                         g_main=gen_pass('g_main'), d_gen=O.Noise('off'), g_rand=gen_pass('g_rand'),
                         perm=torch.from_numpy(g[f's{s}.perm']))
        metric, losses, grads = O.gan_step(G, D, T3, gopt, dopt, oc, scfg, inp['in_text'], inp['in_audio'],
                                           inp['in_mfcc'], inp['target'], inp['vid'], epoch=1, noise=nz)
        if variant == 'warmup':
            assert 'dis' not in losses and 'gen' not in losses and 'D' not in grads
        else:
            assert losses['dis'] == pytest.approx(float(g[f's{s}.dis_error']), rel=2e-5)
        assert ('KLD' in losses) == ('DIV_REG' in losses) == (variant in ('speaker', 'warmup'))
        assert losses['total'] == pytest.approx(float(g[f's{s}.loss']), rel=2e-5)
        assert metric == pytest.approx(float(g[f's{s}.metric']), rel=1e-3, abs=1e-6)
        if s == 0:
            for k in g:
                if k.startswith('s0.grad.G.'):
                    _close(g[k], grads['G'][k[10:]], tol=2e-4)
        for tagm, sd in (('G', G), ('D', D)):
            groups = {}
            for k, v in sd.items():
                if '.net.' in k or k.endswith('num_batches_tracked'):
                    continue
                top = k.split('.')[0]
                groups[top] = groups.get(top, 0.0) + float(v.double().abs().sum())
            for top, val in groups.items():
                assert val == pytest.approx(float(g[f's{s}.abs.{tagm}.{top}']), rel=1e-5), (s, tagm, top)
    for k in g:
        if k.startswith('final.G.'):
            _close(g[k], G[k[8:]], tol=1e-4)
        if k.startswith('final.D.'):
            _close(g[k], D[k[8:]], tol=1e-4)


def test_z_type_random_with_the_regulariser_fails_like_the_reference():
    """processor_v2.py:906-910: z_type 'random' hands vid_indices = None to a generator built with the speaker model as z_obj
    (the only kind Processor builds, :140) -> `assert vid_indices is not None` (net/multimodal_context_net_v2.py:511)."""
    hidden, n_words, n_spk, B, s0 = 32, 64, 12, 2, 4000
    oc = O.ModelCfg(hidden_size=hidden, hidden_size_s2eg=hidden, dropout_prob=0.0)
    G = O.recipe_state_dict(O.generator_shapes(oc, n_words, n_spk), s0 + 1)
    D = O.recipe_state_dict(O.aff_discriminator_shapes(), s0 + 2)
    T3 = O.recipe_state_dict(O.trimodal_shapes(oc, n_words, n_spk), s0 + 4)
    inp = O.recipe_inputs(B, 34, s0 + 100, n_words, n_spk)
    with pytest.raises(AssertionError):
        O.gan_step(G, D, T3, O.AdamState(), O.AdamState(), oc, O.StepCfg(z_type='random'), inp['in_text'], inp['in_audio'],
                   inp['in_mfcc'], inp['target'], inp['vid'], epoch=1, noise=O.StepNoise.fresh())


def test_checkpoint_name_protocol(golden_dir):
    """get_epoch_and_loss fixture (processor_v2.py:53-83) -- checked again against the product in
    test_host_logic.py; here we only make sure the fixture is self-consistent."""
    g = _load(golden_dir, 'misc.npz')
    assert str(g['best'][0]) == 'epoch_000020_loss_0.2500_model.pth.tar'     # 2nd smallest, argpartition(…,2)[1]
    assert str(g['at20'][1]) == '20' and str(g['missing'][0]) == ''


def test_synthesis_loop_matches_reference_render_clip(golden_dir):
    """The oracle's sliding-window synthesis against a run of the reference's own ``Processor.render_clip`` (3 windows,
    seed hand-off, cross-fade, zero-padded last window, word -> frame mapping): tests/golden/gen_golden_synth.py."""
    import sys
    sys.path.insert(0, golden_dir)
    import synth_recipe as R
    g = _load(golden_dir, 'synth_small.npz')
    audio, words, mfcc, poses, eps = R.clip_fixture()
    oc = O.ModelCfg(hidden_size=R.HIDDEN, hidden_size_s2eg=R.HIDDEN, dropout_prob=0.0)
    sdG = O.recipe_state_dict(O.generator_shapes(oc, R.N_WORDS, R.N_SPK), R.SEED0 + 1)
    sdT = O.recipe_state_dict(O.trimodal_shapes(oc, R.N_WORDS, R.N_SPK), R.SEED0 + 4)
    index = {w: 4 + i for i, w in enumerate(R.VOCAB)}
    with torch.no_grad():
        out_t, out_g = O.synthesize_clip(sdG, sdT, oc, g['seed_seq'], audio, R.SR, words, mfcc, R.SPEAKER, eps,
                                         lambda w: index.get(w, 3), fps=R.FPS)
    assert out_t.shape == out_g.shape == (94, 27)
    _close(g['out_trimodal'], out_t)
    _close(g['out_s2ag'], out_g)
    plan, alen = O.synthesis_windows(len(audio), R.SR, 34, 4, R.FPS)
    assert alen == 36266 and [p[2] for p in plan] == [0, 32000, 64000]


@pytest.mark.parametrize('tag', ['small', 'full'])
def test_abl_aff_generator_and_conv_discriminator_match_reference(golden_dir, tag):
    """tests/golden/abl_aff.npz was recorded from the reference's net.multimodal_context_net_v2_abl_aff (PoseGenerator
    without the affective encoder, trained against ConvDiscriminator): forwards in both modes and parameter gradients."""
    import sys
    sys.path.insert(0, golden_dir)
    import s2ag_rng
    c = CASES[tag]
    g = _load(golden_dir, 'abl_aff.npz')
    oc = O.ModelCfg(hidden_size=c['hidden'], hidden_size_s2eg=c['hidden'], dropout_prob=0.0)
    inp = O.recipe_inputs(c['B'], 34, c['seed0'] + 10, c['n_words'], c['n_spk'])
    pre_seq = O.make_pre_seq(inp['target'], 4)
    eps = torch.from_numpy(s2ag_rng.normal(1234, 0, 9001, c['B'] * 16).reshape(c['B'], 16))
    for mode in ('eval', 'train'):
        sdG = O.recipe_state_dict(O.generator_shapes(oc, c['n_words'], c['n_spk'], aff=False), c['seed0'] + 6)
        sdD = O.recipe_state_dict(O.conv_discriminator_shapes(), c['seed0'] + 3)
        tr = mode == 'train'
        if tr:
            sdG = {k: (v.clone().requires_grad_(True) if O.is_param(k) and '.net.' not in k else v) for k, v in sdG.items()}
            for k in list(sdG):
                if '.net.0.' in k or '.net.4.' in k:
                    sdG[k] = sdG[k.replace('.net.0.', '.conv1.').replace('.net.4.', '.conv2.')]
        with torch.set_grad_enabled(tr):
            class PinnedEpsNoDropout(O.Noise):            # the fixture was recorded with every dropout at p = 0
                def dropout(self, name, x, p):
                    return x
            o, z, mu, lv = O.pose_generator_abl_aff(sdG, oc, pre_seq, inp['in_text'], inp['in_mfcc'], inp['vid'], tr,
                                                    PinnedEpsNoDropout({'eps': eps}))
            d = O.conv_discriminator(sdD, o, tr, O.Noise('off'))
        _close(g[f'{tag}.{mode}.out'], o)
        _close(g[f'{tag}.{mode}.z'], z)
        _close(g[f'{tag}.{mode}.mu'], mu)
        _close(g[f'{tag}.{mode}.d'], d)
        if tr:
            ((o * torch.from_numpy(g[f'{tag}.d_out'])).sum() + d.log().mean()).backward()
            for k in g:
                if k.startswith(f'{tag}.grad.'):
                    _close(g[k], sdG[k[len(tag) + 6:]].grad, tol=1e-4)
            _close(g[f'{tag}.train.bn_rv'], sdG['audio_encoder.batch_norm2.running_var'])


def test_embedding_net_and_fgd_match_reference(golden_dir):
    """tests/golden/fgd.npz: the reference's EmbeddingNet ('pose' mode) forwards in eval / train mode and its
    EmbeddingSpaceEvaluator scores (Frechet distance, feature L1 distance, reconstruction-error differences)."""
    g = _load(golden_dir, 'fgd.npz')
    SEED, B, NB = 6100, 24, 3

    def poses(seed, n):
        return torch.from_numpy((np.random.RandomState(seed).standard_normal((n, 34, 27)) * 0.2).astype(np.float32))
    for mode in ('eval', 'train'):
        sd = O.recipe_state_dict(O.embedding_net_shapes(), SEED, scale=3.0, tcn_aliases=False)
        with torch.no_grad():
            feat, mu, lv, rec = O.embedding_net_pose(sd, poses(SEED + 1, B), mode == 'train')
        _close(g[f'{mode}.feat'], feat)
        _close(g[f'{mode}.log_var'], lv)
        _close(g[f'{mode}.recon'], rec)
        if mode == 'train':
            _close(g['train.rv'], sd['pose_encoder.out_net.1.running_var'])
    sd = O.recipe_state_dict(O.embedding_net_shapes(), SEED, scale=3.0, tcn_aliases=False)
    reals, gens, diffs = [], [], []
    with torch.no_grad():
        for b in range(NB):
            real = poses(SEED + 10 + b, B)
            gen = real * 0.5 + poses(SEED + 20 + b, B) * 1.5 + 0.2
            fr, _, _, rr = O.embedding_net_pose(sd, real, False)
            fg, _, _, rg = O.embedding_net_pose(sd, gen, False)
            reals.append(fr.numpy()), gens.append(fg.numpy())
            diffs.append(float((gen - rg).abs().mean()) - float((real - rr).abs().mean()))
    fd, dist = O.fgd_scores(np.vstack(gens), np.vstack(reals))
    assert fd == pytest.approx(float(g['scores'][0]), rel=1e-4) and dist == pytest.approx(float(g['scores'][1]), rel=1e-5)
    np.testing.assert_allclose(diffs, g['recon_err_diff'], rtol=1e-4, atol=1e-7)


def test_push_samples_metrics_match_reference(golden_dir):
    """tests/golden/metrics.npz: the reference's own Processor.push_samples (processor_v2.py:738-774) with its AverageMeter
    over three batches -- per-batch values of the three meters, their running averages and counts."""
    import sys
    sys.path.insert(0, golden_dir)
    from metrics_recipe import BATCH_SIZES, MEAN_DIR_VEC, N_BATCHES, N_PRE, metrics_inputs
    g = _load(golden_dir, 'metrics.npz')
    sums, count = np.zeros(3), 0
    for b in range(N_BATCHES):
        out, tgt = metrics_inputs(b)
        o, t = torch.from_numpy(out), torch.from_numpy(tgt)
        vals = O.push_samples_metrics(o, t, MEAN_DIR_VEC, N_PRE)
        assert torch.equal(o, torch.from_numpy(metrics_inputs(b)[0]))       # inputs untouched
        np.testing.assert_allclose(vals, g['vals'][b], rtol=1e-6)
        sums += np.array(vals) * BATCH_SIZES[b]
        count += BATCH_SIZES[b]
    np.testing.assert_allclose(sums / count, g['avgs'], rtol=1e-6)
    assert count == int(g['counts'][0])


def test_sign_replay_is_audited_and_a_wrong_branch_fails_the_audit():
    """replaying the product's branch decisions in the oracle (use_signs) must not be able to hide a
    product that took the WRONG branch.  The replay files, per site, how many live elements it overrode and how far from
    the kink the oracle's own pre-activation is there; audit_benign accepts a few elements within rounding distance of
    zero and nothing else.  Dropped-out elements of the TCN's ReLU sites (whose recorded decision, read off a post-dropout
    output, is meaningless) are not counted."""
    torch.manual_seed(0)
    x = torch.randn(4, 6, 50)
    x[0, 0, 0] = 1e-7                                   # within rounding distance of the kink
    own = x > 0
    near = own.clone()
    near[0, 0, 0] = False                               # the 'product' landed on the other side there: benign
    with O.use_signs({'s': near}) as u:
        y = O._act(x, 0.3, 's')
    assert torch.equal(y, torch.where(near, x, 0.3 * x)) and u.used == ['s']
    assert u.audit['s']['flipped'] == 1 and u.audit['s']['worst_rel'] < 1e-6
    u.assert_benign(8, 5e-6, 1e-5)
    wrong = own.clone()
    i = int(x.abs().flatten().argmax())
    wrong.view(-1)[i] = not bool(wrong.view(-1)[i])     # the wrong side of the LARGEST pre-activation
    with O.use_signs({'s': wrong}) as u:
        O._act(x, 0.3, 's')
    assert u.audit['s']['flipped'] == 1 and u.audit['s']['worst_rel'] == pytest.approx(1.0)
    with pytest.raises(AssertionError, match='NOT a few elements within rounding distance'):
        u.assert_benign(8, 5e-6, 1e-5)
    many = own.clone()
    many.view(-1)[:100] = ~many.view(-1)[:100]          # many flips: refused whatever their size
    with O.use_signs({'s': many}) as u:
        O._act(x, 0.3, 's')
    with pytest.raises(AssertionError):
        u.assert_benign(8, 5e-6, 1.0)
    # dropped-out elements are not live: a ReLU site whose recorded sign is False wherever the dropout dropped
    keep = torch.rand(4, 6, 50) > 0.3
    rec = own & keep
    with O.use_signs({'s': rec}) as u:
        O._act(x, 0.0, 's', live=keep)
    assert u.audit['s']['flipped'] == 0 and u.audit['s']['live'] == int(keep.sum())
    with O.use_signs({'s': rec}) as u:
        O._act(x, 0.0, 's')
    assert u.audit['s']['flipped'] == int((own & ~keep).sum())
    assert O.Noise({'d': keep.float() / 0.7}).keep('d', 0.3).equal(keep) and O.Noise('off').keep('d', 0.3) is None
