"""BASELINE-size checks (B = 128 per GPU, H = 300, n_words = 20000, 1371 speakers) through size-independent
properties -- the oracle would need minutes at this size, these run in seconds:
  * clip independence: an eval-mode forward of 128 clips equals the concatenation of two 64-clip forwards,
  * time reversal: reversing the frames of the GRU input swaps its two directions,
  * linearity of the conv / linear kernel family,
  * the training step at full size: finite losses, no cooperative-GRU peer time-outs, D phase leaves G untouched,
    graph replay keeps drawing fresh noise, checkpoint round trip restores outputs exactly."""
import os
import tempfile
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

import bench  # noqa: E402  (synthetic TED-shaped batches + processor builder at BASELINE sizes)


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / max(1e-6, float(b.abs().max())))


@pytest.fixture(scope='module')
def pr():
    from speech2affective_gestures_amd import noise
    p = bench.build_processor(128, True)
    noise.manual_seed(7)
    return p


def test_clip_independence_of_the_generator_and_discriminator(pr):
    from speech2affective_gestures_amd import noise
    text, audio, mfcc, target, vid = bench.synthetic_batch(128, 3, pr.device)
    pre = pr._make_pre_seq(target)
    G, D, T3 = pr.s2ag_generator, pr.s2ag_discriminator, pr.trimodal_generator
    for m in (G, D, T3):
        m.eval()
    try:
        with torch.no_grad():
            outs = []
            for sl in (slice(0, 128), slice(0, 64), slice(64, 128)):
                noise.manual_seed(11)            # same eps stream start; eps is per element index, so compare mu-only
                o, z, mu, lv = G(pre[sl], text[sl], mfcc[sl], vid[sl])
                outs.append((mu, lv, D(target[sl]), T3.audio_encoder(audio[sl]), G.text_encoder(text[sl])[0],
                             G.aff_encoder(target[sl])))
            full, a, b = outs
            for i in range(len(full)):
                assert rel(torch.cat((a[i], b[i])), full[i]) < 1e-5, i
    finally:
        for m in (G, D, T3):
            m.train()


def test_gru_time_reversal_swaps_directions():
    import math
    from speech2affective_gestures_amd import ops
    B, T, I, H, L = 128, 34, 88, 300, 1
    g = torch.Generator().manual_seed(0)
    k = 1 / math.sqrt(H)
    w = [((torch.rand(s, generator=g) * 2 - 1) * k).cuda() for s in ((3 * H, I), (3 * H, H), (3 * H,), (3 * H,)) * 2]
    x = torch.randn(B, T, I, generator=g).cuda()
    with torch.no_grad():
        y = ops.gru(x, w, H, L, False, 0.0, None, 0, False)
        y_rev = ops.gru(x.flip(1), w[4:] + w[:4], H, L, False, 0.0, None, 0, False)     # directions' weights swapped
    assert rel(y_rev.flip(1)[..., :H], y[..., H:]) < 1e-5 and rel(y_rev.flip(1)[..., H:], y[..., :H]) < 1e-5
    assert ops.coop_gru_timeouts() == 0


def test_conv_family_is_linear_at_full_size():
    from speech2affective_gestures_amd import ops
    g = torch.Generator().manual_seed(1)
    x1, x2 = torch.randn(128, 34, 300, generator=g).cuda(), torch.randn(128, 34, 300, generator=g).cuda()
    w = (torch.randn(300, 2, 300, generator=g) * 0.05).cuda()
    f = lambda t: ops.conv1d_nlc(t, w, None, pad=8, dil=8, lout=34, w_tap_major=True)   # noqa: E731
    assert rel(f(2.5 * x1 - 0.75 * x2), 2.5 * f(x1) - 0.75 * f(x2)) < 1e-5
    a = torch.randn(128, 36267, 1, generator=g).cuda()
    w1 = (torch.randn(16, 1, 15, generator=g) * 0.2).cuda()
    f1 = lambda t: ops.conv1d_nlc(t, w1, None, stride=5, pad=1600)                       # noqa: E731
    y = f1(a)
    assert y.shape == (128, 7891, 16)
    assert rel(f1(-3.0 * a), -3.0 * y) < 1e-5
    assert float(y[:, :300].abs().max()) == 0.0            # the first 1600/5 - 2 outputs see only zero padding


def test_full_size_training_steps(pr):
    from speech2affective_gestures_amd import ops
    text, audio, mfcc, target, vid = bench.synthetic_batch(128, 4, pr.device)
    before_g = pr.gen_arena.data.clone()
    # D phase alone must not move the generator
    pre = pr._make_pre_seq(target)
    pr._dis_phase(text, mfcc, target, vid, pre, True)
    pr.s2ag_dis_optimizer.step()
    assert torch.equal(pr.gen_arena.data, before_g)
    losses = []
    for _ in range(4):
        pr.train_step(text, audio, mfcc, target, vid)
        L = dict(pr.last_losses)
        assert all(torch.isfinite(torch.tensor(v)) for v in L.values()), L
        losses.append(L)
    assert not torch.equal(pr.gen_arena.data, before_g)
    assert ops.coop_gru_timeouts() == 0
    # same batch, same weights apart from one Adam step, but the dropout/eps noise must differ from replay to replay
    assert losses[1]['total'] != losses[2]['total'] and losses[2]['total'] != losses[3]['total']
    assert 0.0 < losses[-1]['dis'] < 5.0 and losses[-1]['loss'] > 0.0


def test_checkpoint_round_trip(pr):
    from speech2affective_gestures_amd import noise
    text, audio, mfcc, target, vid = bench.synthetic_batch(16, 5, pr.device)
    pre = pr._make_pre_seq(target)
    G = pr.s2ag_generator
    with tempfile.TemporaryDirectory() as d:
        pr.args.work_dir_s2ag = d
        path = pr.save_model(21, 0.1234)
        assert os.path.basename(path) == 'epoch_000021_loss_0.1234_model.pth.tar'
        ck = torch.load(path, map_location='cpu')
        assert set(ck) == {'gen_model_dict', 'dis_model_dict'} and len(ck['gen_model_dict']) == 190
        G.eval()
        with torch.no_grad():
            noise.manual_seed(5)
            ref = G(pre, text, mfcc, vid)[0].clone()
            pr.gen_arena.data.mul_(1.5)                      # wreck the weights, then restore from the file
            open(os.path.join(d, 'epoch_000022_loss_9.0000_model.pth.tar'), 'w').close()
            assert pr.load_model_at_epoch(21)
            noise.manual_seed(5)
            assert torch.equal(G(pre, text, mfcc, vid)[0], ref)
        G.train()
        pr.args.work_dir_s2ag = None


def test_long_context_full_size_steps():
    """BASELINE configs[4] at full size (B = 64, T = 136, H = 300, audio 146 000, n_words 20 000) through the
    size-independent checks: the cooperative GRU at T = 136 loses no peer (error word), graph-replayed steps stay finite
    and keep drawing fresh noise, the D phase leaves G untouched, and an eval forward of 64 clips equals two forwards of 32
    (clip independence).  Oracle parity of the same step at hidden 32: test_gpu_step.py::test_long_clip_steps_136_frames_...."""
    from speech2affective_gestures_amd import noise, ops
    cfgl = bench.CONFIGS['long']
    B, T, AL = cfgl['batch'], cfgl['frames'], cfgl['audio_len']
    p = bench.build_processor(B, True, T, AL)
    noise.manual_seed(17)
    text, audio, mfcc, target, vid = bench.synthetic_batch(B, 6, p.device, T, AL)
    assert mfcc.shape == (B, 37, 286) and p.s2ag_discriminator.out2.weight.shape == (1, T)
    before_g = p.gen_arena.data.clone()
    pre = p._make_pre_seq(target)
    p._dis_phase(text, mfcc, target, vid, pre, True)
    p.s2ag_dis_optimizer.step()
    assert torch.equal(p.gen_arena.data, before_g)
    losses = []
    for _ in range(3):
        p.train_step(text, audio, mfcc, target, vid)
        L = dict(p.last_losses)
        assert all(torch.isfinite(torch.tensor(v)) for v in L.values()), L
        losses.append(L)
    assert not torch.equal(p.gen_arena.data, before_g)
    assert ops.coop_gru_timeouts() == 0
    assert losses[0]['total'] != losses[1]['total'] != losses[2]['total']
    assert 0.0 < losses[-1]['dis'] < 5.0 and losses[-1]['loss'] > 0.0
    G, T3 = p.s2ag_generator, p.trimodal_generator
    G.eval()
    T3.eval()
    try:
        with torch.no_grad():
            outs = []
            for sl in (slice(0, B), slice(0, B // 2), slice(B // 2, B)):
                noise.manual_seed(11)
                outs.append((G(pre[sl], text[sl], mfcc[sl], vid[sl])[2], T3.audio_encoder(audio[sl]),
                             G.text_encoder(text[sl])[0]))
            for i in range(3):
                assert rel(torch.cat((outs[1][i], outs[2][i])), outs[0][i]) < 1e-5, i
            assert outs[0][1].shape == (B, T, 32)
    finally:
        G.train()
        T3.train()
    assert ops.coop_gru_timeouts() == 0


def _grad_report(named_params, ref_grads, skip=lambda k: '.net.' in k):
    """(relative L2 error over all parameters together, {key: (max-norm error, relative L2 error)})."""
    from s2ag_testing import grad_err
    per, num, den = {}, 0.0, 0.0
    for k, p in named_params:
        if skip(k):
            continue
        assert p.grad is not None, k
        a, b = p.grad.detach().cpu().double(), ref_grads[k].double()
        e = grad_err(a, b, k)
        per[k] = (e, 0.0 if e == 0.0 else float((a - b).norm() / max(1e-12, float(b.norm()))))
        if e != 0.0:
            num += float((a - b).square().sum())
            den += float(b.square().sum())
    return (num / max(den, 1e-300)) ** 0.5, per


@pytest.mark.parametrize('config', ['step', 'long'])
def test_full_size_step_matches_the_oracle(monkeypatch, config):
    """ONE training step at the full BASELINE size against the oracle's gan_step fed the product's materialised masks:
    configs[1] (B = 128, T = 34) and configs[4] (B = 64, T = 136, audio 146 000), both at H = 300, n_words 20 000, 1 371
    speakers, dropout on -- the sizes bench.py times, through the kernels it times (three-pass lockstep cooperative GRU,
    two-slice recurrence, clip-resident TCN, split-operand GEMMs).  Losses, metric, BatchNorm running statistics and --
    since r04, with the product's branch decisions of all seven passes replayed in the oracle -- EVERY generator and
    discriminator gradient strictly (1e-3 max-norm)."""
    from oracle import s2ag_oracle as O
    from s2ag_testing import PASSES_PER_STEP, STEP_SEED, oracle_cfg, to_cuda
    from test_gpu_step import _materialise_step_noise, make_processor
    from speech2affective_gestures_amd import noise, ops
    from speech2affective_gestures_amd import processor_v2 as P
    wl = bench.CONFIGS[config]
    B, T, AL = wl['batch'], wl['frames'], wl['audio_len']
    hidden, n_words, n_spk, s0 = 300, bench.N_WORDS, bench.N_SPK, 8800
    mfcc_len = -(-AL // 512)
    pr, sds = make_processor(hidden, n_words, n_spk, B, s0, 0.3, T=T, audio_len=AL)
    G, D, T3 = ({k: v.clone() for k, v in sds[n].items()} for n in ('G', 'D', 'T3'))
    noise.manual_seed(STEP_SEED)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0))
    monkeypatch.setattr(P.torch, 'randperm', lambda n, *a, **k: perm.cuda())
    inp = O.recipe_inputs(B, T, s0 + 100, n_words, n_spk, audio_len=AL, mfcc_len=mfcc_len)
    gi = to_cuda(inp)
    nz = _materialise_step_noise(pr, 0, B, T, hidden)
    nz.perm = perm
    from s2ag_testing import StepSignTap, grad_err
    with StepSignTap(pr, 0) as tap:        # every ReLU / LeakyReLU decision of the seven module passes, filed by (module, pass)
        ret = pr.forward_pass_s2ag(gi['in_text'], gi['in_audio'], gi['in_mfcc'], gi['target'], gi['vid'], True)
    monkeypatch.undo()
    signs = tap.signs_per_pass()
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    # a copy of D's gradient: the generator phase never touches D's arena (frozen there), so it is still the D step's
    metric, losses, grads = O.gan_step(G, D, T3, O.AdamState(), O.AdamState(), oracle_cfg(hidden, 0.3, T), O.StepCfg(),
                                       inp['in_text'], inp['in_audio'], inp['in_mfcc'], inp['target'], inp['vid'], epoch=1,
                                       noise=nz, signs=signs)
    for name in StepSignTap.PASSES:
        assert set(O.gan_step.signs_used[name]) == set(signs[name]), name
    assert len(signs['g_main']) == 5 + 12 + 6 + 1 and len(signs['d_gen']) == 6
    for k in ('dis', 'total', 'loss', 'KLD', 'DIV_REG', 'gen'):
        assert pr.last_losses[k] == pytest.approx(losses[k], rel=3e-4, abs=1e-6), k
    assert ret[0] == pytest.approx(metric, rel=5e-3, abs=2e-6)
    # the replay must not hide a wrong branch: what it overrode are a few live elements within rounding
    # distance of the kink, in every pass -- and the oracle WITHOUT any replay gives the same losses
    from s2ag_testing import REPLAY_LIMITS
    info = O.audit_benign(O.gan_step.signs_audit, *REPLAY_LIMITS, what=f'full-size step {config}')
    print(f'[replay audit full-size step {config}] {info}')
    G0, D0, T0 = ({k: v.clone() for k, v in sds[n].items()} for n in ('G', 'D', 'T3'))
    nz0 = _materialise_step_noise(pr, 0, B, T, hidden)
    nz0.perm = perm
    _, own, _ = O.gan_step(G0, D0, T0, O.AdamState(), O.AdamState(), oracle_cfg(hidden, 0.3, T), O.StepCfg(), inp['in_text'],
                           inp['in_audio'], inp['in_mfcc'], inp['target'], inp['vid'], epoch=1, noise=nz0)
    for k in ('dis', 'total', 'loss', 'KLD', 'DIV_REG', 'gen'):
        assert pr.last_losses[k] == pytest.approx(own[k], rel=3e-4, abs=1e-6), ('without replay', k)
    # STRICT since r04: the oracle differentiates the same piecewise-linear function as the product (branch decisions of all
    # seven passes replayed), so no kink-tolerant criterion is needed -- every gradient tensor of G and of D within 1e-3 of
    # its largest element (r03: 5 % per tensor, 0.2 max-norm).  The small-batch twin of this test
    # (tests/test_gpu_step.py::test_one_step_strictly_with_the_products_branch_decisions) measures 4e-5 on the device model.
    for tag, mod in (('G', pr.s2ag_generator), ('D', pr.s2ag_discriminator)):
        total, per = _grad_report(mod.named_parameters(), grads[tag])
        worst = sorted(per.items(), key=lambda kv: -kv[1][0])[:3]
        print(f'[full-size step {config}, strict] {tag}: relative L2 error over all parameters {total:.2e}; worst tensors: ' +
              ', '.join(f'{k} max {e:.1e} L2 {l:.1e}' for k, (e, l) in worst))
        for k, (e, l) in per.items():
            assert e < 1e-3, (tag, k, e, l)
    for k, v in pr.s2ag_generator.state_dict().items():
        if 'running_var' in k:
            assert rel(v.cpu(), G[k]) < 3e-4, k
    for k, v in pr.s2ag_discriminator.state_dict().items():
        if 'running_var' in k:
            assert rel(v.cpu(), D[k]) < 3e-4, k
    assert ops.coop_gru_timeouts() == 0


@pytest.mark.parametrize('B', [4, 256])
def test_conv1d_roofline_run_gradients_match_the_oracle_strictly(B):
    """BASELINE configs[3] at its own size (B = 256; B = 4 is what the CPU device model of tests/emu can run): WavEncoder +
    TextEncoderTCN forward + backward, train mode, dropout on, against the oracle fed the product's masks AND the product's
    branch decisions (SignTap / oracle.use_signs: the three BatchNorm + LeakyReLU pairs of the wave encoder -- the first
    recomputed by s2ag_wave12_act_signs --, the twelve ReLU sites of the TCN): outputs 3e-4, EVERY parameter gradient within
    1e-3 of its largest element.  (r03 accepted 5 % per tensor here; fp32 mode -- bf16 mode has its own tests.)"""
    from oracle import s2ag_oracle as O
    from s2ag_testing import SignTap, grad_err
    from speech2affective_gestures_amd import noise, ops
    from speech2affective_gestures_amd.net.multimodal_context_net_v2 import TextEncoderTCN, WavEncoder
    T = 34
    cfg = bench.make_cfg()
    oc = O.ModelCfg()
    n_words = bench.N_WORDS if B > 16 else 500
    sd = O.recipe_state_dict({**O._wav_encoder_shapes('wav.'),
                              **O._text_encoder_shapes('txt.', n_words, 300, oc.hidden_size, oc.n_layers)}, 11)
    both = torch.nn.ModuleDict(dict(wav=WavEncoder(), txt=TextEncoderTCN(cfg, n_words, 300, dropout=cfg.dropout_prob))).cuda().train()
    wav, txt = both['wav'], both['txt']
    wav.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith('wav.')}, strict=True)
    txt.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith('txt.')}, strict=True)
    inp = O.recipe_inputs(B, T, 5, n_words, bench.N_SPK)
    noise.manual_seed(31)
    nz = torch.tensor([31, 0], dtype=torch.int64, device='cuda')
    with noise.noise_pass('cuda'), SignTap(both, tcn_prefix='txt.') as tap:
        yw = wav(inp['in_audio'].cuda())
        yt = txt(inp['in_text'].cuda())[0]
    signs = tap.signs()
    pin = {'txt.emb_drop': ops.dropout_mask(nz, txt.site, txt.drop.p, (B, T, 300)).cpu()}
    for i, blk in enumerate(txt.tcn.network):
        for j in (0, 1):
            pin[f'txt.tcn.{i}.drop{j + 1}'] = ops.dropout_mask(nz, blk.sites[j], blk.p, (B, T, 300)).cpu().transpose(1, 2)
    leaf = {k: (v.detach().clone().requires_grad_(True) if O.is_param(k) and '.net.' not in k else v.clone())
            for k, v in sd.items()}
    for k in list(leaf):
        if '.net.0.' in k or '.net.4.' in k:
            leaf[k] = leaf[k.replace('.net.0.', '.conv1.').replace('.net.4.', '.conv2.')]
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    with O.use_signs(signs) as used:
        rw = O.wav_encoder(leaf, 'wav.', inp['in_audio'], True)
        rt = O.text_encoder_tcn(leaf, 'txt.', inp['in_text'], True, oc.dropout_prob, O.Noise(pin))
    assert set(used.used) == set(signs) and len(signs) == 3 + 12, (sorted(signs), sorted(used.used))
    assert rel(yw.cpu(), rw) < 3e-4 and rel(yt.cpu(), rt) < 3e-4
    # (the replay must not hide a wrong branch: see test_full_size_step_matches_the_oracle)
    from s2ag_testing import REPLAY_LIMITS
    info = used.assert_benign(*REPLAY_LIMITS, what=f'configs[3] B={B}')
    print(f'[replay audit configs[3] B={B}] {info}')
    with torch.no_grad():
        own = {k: v.detach().clone() for k, v in sd.items()}
        assert rel(yw.cpu(), O.wav_encoder(own, 'wav.', inp['in_audio'], True)) < 3e-4
        assert rel(yt.cpu(), O.text_encoder_tcn(own, 'txt.', inp['in_text'], True, oc.dropout_prob, O.Noise(pin))) < 3e-4
    g = torch.Generator().manual_seed(2)
    dw, dt = torch.randn(rw.shape, generator=g), torch.randn(rt.shape, generator=g)
    ((rw * dw).sum() + (rt * dt).sum()).backward()
    ((yw * dw.cuda()).sum() + (yt * dt.cuda()).sum()).backward()
    dead = ('feat_extractor.0.bias', 'feat_extractor.3.bias', 'feat_extractor.6.bias')      # a BatchNorm cancels them: true gradient 0
    named = [('wav.' + k, p) for k, p in wav.named_parameters() if k not in dead] + \
        [('txt.' + k, p) for k, p in txt.named_parameters() if '.net.' not in k]
    errs = {k: grad_err(p.grad, leaf[k].grad, k) for k, p in named}
    print(f'[configs[3] strict, B={B}] out wav {rel(yw.cpu(), rw):.1e} txt {rel(yt.cpu(), rt):.1e}; worst gradients (max-norm): ' +
          ', '.join(f'{k} {e:.1e}' for k, e in sorted(errs.items(), key=lambda kv: -kv[1])[:4]))
    for k, e in errs.items():
        assert e < 1e-3, (k, e)
