"""Shared helpers of the GPU parity tests: reference-shaped configs, recipe weights, product module builders."""
import types

import torch

from oracle import s2ag_oracle as O

STEP_SEED, G_Z_SITE, PGT_Z_SITE, PASSES_PER_STEP = 1234, 9001, 9002, 7     # must equal tests/golden/gen_golden.py


class Vocab:                       # duck-typed like utils/vocab.py: class name 'Vocab', .n_words, .word2index
    def __init__(self, n):
        self.n_words = n
        self.word2index = {'v%d' % i: i for i in range(n)}
        self.word_embedding_weights = None


def make_cfg(hidden, drop, n_poses=34):
    return types.SimpleNamespace(n_pre_poses=4, n_poses=n_poses, input_context='both', hidden_size=hidden,
                                 hidden_size_s2eg=hidden, n_layers=4, dropout_prob=drop, freeze_wordembed=False,
                                 loss_warmup=0, loss_gan_weight=5.0, z_type='speaker', loss_reg_weight=0.05,
                                 loss_regression_weight=500, loss_kld_weight=0.1, wordembed_dim=300,
                                 learning_rate=5e-4, discriminator_lr_weight=0.2)


def oracle_cfg(hidden, drop, n_poses=34):
    return O.ModelCfg(n_poses=n_poses, hidden_size=hidden, hidden_size_s2eg=hidden, dropout_prob=drop)


def recipe_sds(hidden, n_words, n_spk, seed0, n_poses=34, mfcc_length=71):
    oc = oracle_cfg(hidden, 0.0, n_poses)
    return dict(G=O.recipe_state_dict(O.generator_shapes(oc, n_words, n_spk, mfcc_length=mfcc_length), seed0 + 1),
                D=O.recipe_state_dict(O.aff_discriminator_shapes(n_poses), seed0 + 2),
                CD=O.recipe_state_dict(O.conv_discriminator_shapes(n_poses=n_poses), seed0 + 3),
                T3=O.recipe_state_dict(O.trimodal_shapes(oc, n_words, n_spk), seed0 + 4),
                GA=O.recipe_state_dict(O.generator_shapes(oc, n_words, n_spk, audio='wav'), seed0 + 5))


def build_product(hidden, n_words, n_spk, drop, seed0, which=('G', 'D', 'CD', 'T3', 'GA')):
    from speech2affective_gestures_amd.net import multimodal_context_net_v2 as m2
    from speech2affective_gestures_amd.net import multimodal_context_net_v2_abl_audio as m2a
    cfg, spk = make_cfg(hidden, drop), Vocab(n_spk)
    sds = recipe_sds(hidden, n_words, n_spk, seed0)
    mk = dict(G=lambda: m2.PoseGenerator(cfg, 27, n_words, 300, None, 71, 37, 34, z_obj=spk),
              D=lambda: m2.AffDiscriminator(27), CD=lambda: m2.ConvDiscriminatorTriModal(27),
              T3=lambda: m2.PoseGeneratorTriModal(cfg, 27, n_words, 300, None, z_obj=spk),
              GA=lambda: m2a.PoseGenerator(cfg, 27, n_words, 300, None, 71, 37, 34, z_obj=spk))
    mods = {}
    for k in which:
        m = mk[k]()
        m.load_state_dict(sds[k], strict=True)          # same keys / shapes as the reference, by construction
        mods[k] = m.cuda()
        if hasattr(m, 'z_site'):
            m.z_site = PGT_Z_SITE if k == 'T3' else G_Z_SITE
    return cfg, mods, sds


def set_dropout(module, p_tcn=None, p_emb=None, p_gru=None):
    """Override the dropout probabilities of a product module tree (None = leave)."""
    from speech2affective_gestures_amd.net.multimodal_context_net_v2 import GRU, TextEncoderTCN
    from speech2affective_gestures_amd.net.tcn import TemporalBlock
    for sub in module.modules():
        if isinstance(sub, TemporalBlock) and p_tcn is not None:
            sub.p = p_tcn
        if isinstance(sub, TextEncoderTCN) and p_emb is not None:
            sub.drop.p = p_emb
        if isinstance(sub, GRU) and p_gru is not None:
            sub.dropout = p_gru


def to_cuda(d):
    return {k: v.cuda() for k, v in d.items()}


class StepSignTap:
    """SignTap for one whole GAN step (Processor.forward_pass_s2ag): the seven module passes of the step -- G(dis), D(real),
    D(fake), the frozen baseline, G(main), D(gen), G(rand); processor_v2.py:798-937 -- run inside ``noise.use_pass`` scopes
    whose snapshot carries the pass counter, so every recorded activation is filed under (module, pass) whatever the order
    or the stream the trainer issues the passes in.  The generator's dropout-free encoders run ONCE per step for its three
    passes (PoseGenerator.share_passes): their decisions count for all three.

        with StepSignTap(pr, counter0) as tap:
            pr.forward_pass_s2ag(...)
        O.gan_step(..., signs=tap.signs_per_pass())"""
    PASSES = ('g_dis', 'd_real', 'd_fake', 'pgt', 'g_main', 'd_gen', 'g_rand')
    OWNER = ('G', 'D', 'D', 'T3', 'G', 'D', 'G')

    def __init__(self, pr, counter0):
        from speech2affective_gestures_amd import noise, ops, wave12
        self.ops, self.noise, self.w12, self.c0 = ops, noise, wave12, int(counter0)
        self.mods = {'G': pr.s2ag_generator, 'D': pr.s2ag_discriminator, 'T3': pr.trimodal_generator}
        self.owner = {}
        for tag, m in self.mods.items():
            for _, sub in m.named_modules():
                self.owner[id(sub)] = tag
            for _, q in m.named_parameters():
                self.owner[id(q)] = tag
        self.children = {}

    def _pass(self):
        stack = getattr(self.noise._tls, 'stack', None)
        if not stack:
            return None
        k = int(stack[-1][1]) - self.c0
        return k if 0 <= k < 7 else None

    def _child(self, tag, k):
        key = (tag, k)
        if key not in self.children:
            c = SignTap(self.mods[tag])
            c._w12, c.pre = self.w12, {}
            self.children[key] = c
        return self.children[key]

    def __enter__(self):
        ops = self.ops
        self._orig = (ops.batch_norm_act, ops.add_act, ops.linear, ops.tcn_fused32, self.w12.head_f32)
        o_bn, o_add, o_lin, o_tcn, o_head = self._orig

        def by_pass(k):                      # modules without an object to identify them by: the pass says whose they are
            return self.OWNER[k] if k is not None else 'G'

        def bn_act(x, bn, slope=1.0, chan_map=None, training=None):
            y = o_bn(x, bn, slope=slope, chan_map=chan_map, training=training)
            if slope != 1.0 and id(bn) in self.owner:
                c = self._child(self.owner[id(bn)], self._pass())
                c.bn.append((c.names[id(bn)], y.detach()))
            return y

        def add_act(a, b, slope):
            y = o_add(a, b, slope)
            if slope == 0.0:                 # a TemporalBlock's ReLU(out + x), layer by layer (autograd passes only)
                if y.requires_grad:
                    k = self._pass()
                    self._child(by_pass(k), k).tcn_adds.append(y.detach())
            elif slope != 1.0:
                k = self._pass()
                self._child(by_pass(k), k).adds.append(y.detach())
            return y

        def linear(x, w, bias, act=0, slope=1.0):
            y = o_lin(x, w, bias, act=act, slope=slope)
            if act == 1 and slope != 1.0 and id(w) in self.owner:
                c = self._child(self.owner[id(w)], self._pass())
                c.lin.append((c.params[id(w)], y.detach()))
            return y

        # (saved tensors are gone once the step has back-propagated: what SignTap.signs() reads from grad_fn is read here)
        def tcn(*a, **kw):
            r = o_tcn(*a, **kw)
            k = self._pass()
            t = r[0] if isinstance(r, tuple) else r
            if t.grad_fn is not None:            # (a no_grad pass saves nothing and back-propagates nothing: its decisions
                c = self._child(by_pass(k), k)   # only enter forward values, where a flipped 1e-7 input changes 1e-7)
                c.tcn.append(t)
                c.pre.update(c.signs())
                c.tcn.clear()
            return r

        def head(wav, fe):
            z2 = o_head(wav, fe)
            if id(fe[1]) in self.owner and z2.grad_fn is not None:
                c = self._child(self.owner[id(fe[1])], self._pass())
                c.heads.append((c.names[id(fe[1])], z2, fe[0].bias))
                c.pre.update(c.signs())
                c.heads.clear()
            return z2

        # the layer-by-layer TemporalConvNet (T beyond the clip-resident kernels, e.g. configs[4]'s 136 frames): conv epilogues
        # by their dropout site id, ReLU(out + x) through add_act (slope 0) -- like the fused form, autograd passes only
        site_owner = {}
        for tag, m in self.mods.items():
            for sid in tcn_site_names(m, 'text_encoder.'):
                site_owner[sid] = tag
        self._o_conv = o_conv = ops.conv1d_nlc

        def conv1d_nlc(*a, **kw):
            y = o_conv(*a, **kw)
            sid = kw.get('site')
            if sid in site_owner and kw.get('slope', 1.0) == 0.0 and y.requires_grad:
                c = self._child(site_owner[sid], self._pass())
                c.tcn_conv.append((c.tcn_sites[sid], y.detach()))
            return y
        ops.conv1d_nlc = conv1d_nlc
        ops.batch_norm_act, ops.add_act, ops.linear, ops.tcn_fused32 = bn_act, add_act, linear, tcn
        self.w12.head_f32 = head
        return self

    def __exit__(self, *a):
        ops = self.ops
        ops.batch_norm_act, ops.add_act, ops.linear, ops.tcn_fused32, self.w12.head_f32 = self._orig
        ops.conv1d_nlc = self._o_conv

    def signs_per_pass(self):
        """{pass name: {site: bool tensor}} for oracle.gan_step(signs=...)."""
        per = {key: {**c.signs(), **c.pre} for key, c in self.children.items()}
        out = {}
        for k, (name, tag) in enumerate(zip(self.PASSES, self.OWNER)):
            d = dict(per.get((tag, None), {}))            # ran outside every pass scope: the generator's shared encoders
            d.update(per.get((tag, k), {}))
            for (t2, k2), sg in per.items():              # shared encoders that ran inside the FIRST pass's scope
                if t2 == tag and k2 is not None and k2 != k:
                    for site, v in sg.items():
                        if site not in d and ('aff_encoder.' in site or 'audio_encoder.' in site):
                            d[site] = v
            out[name] = d
        return out


import re as _re

_DEAD = _re.compile(r'(audio_encoder\.conv[1-4]\.bias|audio_encoder\.feat_extractor\.[036]\.bias|'
                    r'st_gcn[12]\.tcn\.2\.bias|st_gcn[12]\.residual\.0\.bias|aff_encoder\.conv[34]\.bias|'
                    r'pre_conv\.[013]\.bias)$')


def is_dead_bias(key: str) -> bool:
    """Conv biases that feed a BatchNorm directly: the batch mean removes them, so their true gradient is exactly
    zero (both implementations hold ~1e-8 rounding noise there) and Adam then moves them by +-lr per step in a
    noise-determined direction -- in the reference too.  They cannot influence any output."""
    return _DEAD.search(key) is not None


def is_noise_driven_after_adam(key: str) -> bool:
    """State that legitimately random-walks under Adam (update = lr * m / sqrt(v) turns ~1e-8 rounding noise into
    +-lr steps): dead biases, the st_gcn2 graph-conv bias (its k = 0 slice only adds a per-channel constant, which the
    following BatchNorm2d removes) and the running means that track such biases.  Excluded from after-step WEIGHT
    comparisons only; their gradients and every loss/metric are still compared."""
    return is_dead_bias(key) or key.endswith('running_mean') or key.endswith('st_gcn2.gcn.conv.bias')


def grad_err(a, b, key=''):
    """max |a-b| / max |b|; for dead biases (see is_dead_bias) only require both sides to be ~0."""
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    if is_dead_bias(key):
        assert float(a.abs().max()) < 1e-3 and float(b.abs().max()) < 1e-3, key
        return 0.0
    return float((a - b).abs().max() / max(1e-6, float(b.abs().max())))


# What a sign replay (oracle.use_signs) may override, per site: at most 8 + 5e-6 x (live elements of the site) elements decided
# against the oracle's own x > 0, each with |x| <= 5e-5 of the site's largest |x| (oracle/s2ag_oracle.py `audit_benign`).
# A decision can only differ where |x| is below the product's own forward error at that site, which is ~1e-6 of the largest
# element behind fp32 arithmetic and up to ~2e-5 behind the two-piece (16 mantissa bit) products of the default mode.
# Measured on the device model: modules at H = 300, B = 88 .. 128 (1.1e7 .. 2.6e7 live elements in 22 .. 24 sites): 20 .. 23
# flips in all, |x| <= 3.2e-6; configs[3] at B = 256: 52 of 7.2e7, 2.2e-6; the whole step at B = 128: worst site
# d_gen/aff_encoder.st_gcn1.out -- the discriminator fed the GENERATOR'S OUTPUT, i.e. behind all of its two-piece products --
# 1 flip at 1.98e-5.  A wrong branch at a pre-activation of ordinary size (1e-2 .. 1 of the largest) fails this by orders.
REPLAY_LIMITS = (8, 5e-6, 5e-5)


def adam_close(v, ref, lr, steps, frac=5e-3):
    """Weights after a few Adam steps.  Adam turns a gradient element g into a step lr*m/sqrt(v), i.e. ~lr*sign(g)
    early on, so elements whose true gradient is at rounding-noise level may legitimately differ by a fraction of
    lr per step.  Require: almost every element within 0.1*lr, and none beyond what sign flips can produce."""
    d = (torch.as_tensor(v).detach().cpu().double() - torch.as_tensor(ref).detach().cpu().double()).abs()
    frac_off = float((d > 0.1 * lr).double().mean())
    return frac_off < frac and float(d.max()) <= 2.05 * lr * steps, (frac_off, float(d.max()))


class SignTap:
    """Records the branch decisions of the product's piecewise-linear activations during one forward pass of ``module``
    (a generator or discriminator), in the site names and tensor layouts of the oracle (oracle.use_signs):

        with SignTap(G) as tap:
            out = G(...)
        with O.use_signs(tap.signs()):
            ref = O.pose_generator(...)

    What is recorded are the products's own OUTPUTS (y > 0: with a positive slope the sign of the output is the sign of
    the input; a dropped-out element is 0 and its branch is immaterial), so nothing in the product changes.  Covers the
    BatchNorm + activation pairs, the residual add + LeakyReLU of the ST-GCN blocks, the LeakyReLU Linears and the
    clip-resident TemporalConvNet (h1, h2, y of every block)."""

    def __init__(self, module, tcn_prefix='text_encoder.'):
        from speech2affective_gestures_amd import ops
        self.ops, self.module, self.tcn_prefix = ops, module, tcn_prefix
        self.names = {id(m): n for n, m in module.named_modules()}
        self.params = {id(p): n for n, p in module.named_parameters()}
        self.bn, self.adds, self.lin, self.tcn, self.heads = [], [], [], [], []
        # the layer-by-layer TemporalConvNet (clips longer than the clip-resident kernels take, e.g. T = 136): its ReLU sites
        # are conv epilogues (identified by their dropout site id) and the residual add + ReLU of every block
        self.tcn_sites = tcn_site_names(module, tcn_prefix)
        self.tcn_conv, self.tcn_adds = [], []

    def __enter__(self):
        ops = self.ops
        self._orig = (ops.batch_norm_act, ops.add_act, ops.linear, ops.tcn_fused32)
        o_bn, o_add, o_lin, o_tcn = self._orig
        self._o_conv = o_conv = ops.conv1d_nlc

        def conv1d_nlc(*a, **k):
            y = o_conv(*a, **k)
            if k.get('site') in self.tcn_sites and k.get('slope', 1.0) == 0.0:
                self.tcn_conv.append((self.tcn_sites[k['site']], y.detach()))
            return y
        ops.conv1d_nlc = conv1d_nlc

        def bn_act(x, bn, slope=1.0, chan_map=None, training=None):
            y = o_bn(x, bn, slope=slope, chan_map=chan_map, training=training)
            if slope != 1.0 and id(bn) in self.names:
                self.bn.append((self.names[id(bn)], y.detach()))
            return y

        def add_act(a, b, slope):
            y = o_add(a, b, slope)
            if slope == 0.0:                 # ReLU(out + x) of a TemporalBlock run layer by layer, in block order
                self.tcn_adds.append(y.detach())
            elif slope != 1.0:
                self.adds.append(y.detach())
            return y

        def linear(x, w, bias, act=0, slope=1.0):
            y = o_lin(x, w, bias, act=act, slope=slope)
            if act == 1 and slope != 1.0 and id(w) in self.params:
                self.lin.append((self.params[id(w)], y.detach()))
            return y

        def tcn(*a, **k):
            r = o_tcn(*a, **k)
            self.tcn.append(r[0] if isinstance(r, tuple) else r)
            return r
        ops.batch_norm_act, ops.add_act, ops.linear, ops.tcn_fused32 = bn_act, add_act, linear, tcn
        # the fused head of the wave encoder (wave12.py): BatchNorm 1 + LeakyReLU happen inside its launches; the branch
        # decisions come from s2ag_wave12_act_signs, which forms conv1's output exactly as they do
        from speech2affective_gestures_amd import wave12
        self._w12, self._o_head = wave12, wave12.head_f32

        def head(wav, fe):
            z2 = self._o_head(wav, fe)
            if id(fe[1]) in self.names:
                self.heads.append((self.names[id(fe[1])], z2, fe[0].bias))
            return z2
        wave12.head_f32 = head
        return self

    def __exit__(self, *a):
        ops = self.ops
        ops.batch_norm_act, ops.add_act, ops.linear, ops.tcn_fused32 = self._orig
        ops.conv1d_nlc = self._o_conv
        self._w12.head_f32 = self._o_head

    def signs(self):
        import numpy as np
        out = {}
        aff = getattr(self.module, 'aff_encoder', None)
        cols = {}
        if aff is not None:
            cols = {'aff_encoder.st_gcn1.': torch.from_numpy(np.asarray(aff.out1)), 'aff_encoder.st_gcn2.': torch.from_numpy(np.asarray(aff.out2))}

        def vertex_layout(y, oc):           # (n, t, V * C) in column order oc[w, c] -> (n, c, t, w)
            idx = oc.to(y.device).t().reshape(-1)                         # [c][w] -> column
            V, Cc = oc.shape
            return y[:, :, idx].view(y.shape[0], y.shape[1], Cc, V).permute(0, 2, 1, 3)
        for name, y in self.bn:
            pre = name[:name.rfind('tcn.0')] if name.endswith('tcn.0') else None
            if pre is not None and pre in cols:                           # BatchNorm2d + ReLU inside an ST-GCN block
                out[name + '.'] = (vertex_layout(y, cols[pre]) > 0).cpu()
            else:
                out[name + '.'] = (y > 0).permute(0, 2, 1).cpu()
        for name, z2, b1 in self.heads:
            wav, pk, coef1 = z2.grad_fn.saved_tensors
            out[name + '.'] = self._w12.act_signs(wav, pk, b1, coef1, False, z2.grad_fn.pad).permute(0, 2, 1).cpu()
        for key, y in zip(sorted(cols), self.adds):                      # st_gcn1 runs before st_gcn2
            out[key + 'out'] = (vertex_layout(y, cols[key]) > 0).cpu()
        for pname, y in self.lin:
            mod = pname[:pname.rfind('.') + 1]                            # 'audio_encoder.linear1.' / 'out.0.'
            out['out.1.' if mod == 'out.0.' else mod] = (y > 0).cpu()
        for t_out in self.tcn:
            x2, saved, y_last = t_out.grad_fn.saved_tensors
            B, T, Cch = t_out.shape
            nb = (saved.shape[0] + 1) // 3
            prefix = self.tcn_prefix
            for b in range(nb):
                tens = [saved[3 * b], saved[3 * b + 1], saved[3 * b + 2] if b < nb - 1 else y_last[:B * T]]
                for j, t in enumerate(tens):
                    out[f'{prefix}tcn.{b}.relu{j + 1}'] = (t.view(B, T, Cch) > 0).permute(0, 2, 1).cpu()
        for name, y in self.tcn_conv:                                     # layer-by-layer TemporalConvNet
            out[name] = (y > 0).permute(0, 2, 1).cpu()
        for b, y in enumerate(self.tcn_adds):
            out[f'{self.tcn_prefix}tcn.{b}.relu3'] = (y > 0).permute(0, 2, 1).cpu()
        return out


def tcn_site_names(module, prefix):
    """{dropout site id of a TemporalBlock's conv: the oracle's name of the ReLU behind it} for every block under ``module``."""
    out = {}
    for name, m in module.named_modules():
        if hasattr(m, 'sites') and hasattr(m, 'dilation') and '.network.' in '.' + name:
            b = int(name.rsplit('.', 1)[1])
            out[int(m.sites[0])], out[int(m.sites[1])] = f'{prefix}tcn.{b}.relu1', f'{prefix}tcn.{b}.relu2'
    return out
