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
