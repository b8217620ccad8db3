"""CPU oracle for the S2AG generator/discriminator GAN step.

TEST INFRASTRUCTURE ONLY.  This file is a plain PyTorch-CPU (fp32) *restatement* of the algorithm of
the reference hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product package (``speech2affective_gestures_amd``) never
does and has no CPU fallback.

Parity status: PINNED.  Every function below is checked against golden vectors produced by
importing the reference itself in the build container (``tests/golden/gen_golden.py`` is the
generating script; ``tests/test_oracle_golden.py`` is the check).

Style: purely functional.  A model is a flat ``dict[str, Tensor]`` that uses the *reference's*
``state_dict`` key names, so a reference checkpoint can be fed directly.
All randomness (dropout masks, re-parametrisation noise, the speaker permutation) is explicit:
callers pass a ``Noise`` object; ``Noise(None)`` draws from torch's CPU generator.

Reference citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

# ----------------------------------------------------------------------------------------------
# skeleton constants  (utils/ted_db_utils.py:14-19)
# ----------------------------------------------------------------------------------------------
DIR_EDGE_PAIRS = [(0, 1), (1, 2), (0, 3), (3, 4), (4, 5), (0, 6), (6, 7), (7, 8)]
BODY_PARTS_EDGE_IDX = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]
BODY_PARTS_EDGE_PAIRS = [(0, 1), (0, 2)]
N_DIR_VECS = 9
N_BODY_PARTS = 3


# ----------------------------------------------------------------------------------------------
# graph adjacency  (net/utils/graph.py:4-142; strategy 'spatial', centre node 0)
# ----------------------------------------------------------------------------------------------
def hop_distance(num_nodes: int, links: Sequence[Tuple[int, int]], max_hop: int) -> np.ndarray:
    """graph.py:108-120 -- smallest d<=max_hop with (A^d)[i,j] > 0, else inf (self links included)."""
    adj = np.zeros((num_nodes, num_nodes))
    for i in range(num_nodes):
        adj[i, i] = 1
    for a, b in links:
        adj[a, b] = 1
        adj[b, a] = 1
    dis = np.full((num_nodes, num_nodes), np.inf)
    power = np.eye(num_nodes)
    reach = []
    for d in range(max_hop + 1):
        reach.append(power > 0)
        power = power @ adj
    for d in range(max_hop, -1, -1):
        dis[reach[d]] = d
    return dis


def spatial_adjacency(num_nodes: int, links: Sequence[Tuple[int, int]], max_hop: int = 2) -> np.ndarray:
    """graph.py:62-105 -- K x V x V stack, K = 1 + 2*max_hop, column-normalised (A . D^-1)."""
    dis = hop_distance(num_nodes, links, max_hop)
    reach = np.zeros((num_nodes, num_nodes))
    reach[dis <= max_hop] = 1
    col = reach.sum(0)
    norm = reach / np.where(col > 0, col, 1)[None, :]          # graph.py:123-131
    centre = dis[:, 0]
    out = []
    for hop in range(max_hop + 1):
        root = np.zeros_like(reach)
        close = np.zeros_like(reach)
        further = np.zeros_like(reach)
        for i in range(num_nodes):
            for j in range(num_nodes):
                if dis[j, i] != hop:
                    continue
                if centre[j] == centre[i]:
                    root[j, i] = norm[j, i]
                elif centre[j] > centre[i]:
                    close[j, i] = norm[j, i]
                else:
                    further[j, i] = norm[j, i]
        if hop == 0:
            out.append(root)
        else:
            out.append(root + close)
            out.append(further)
    return np.stack(out)


_A_CACHE: Dict[str, Tensor] = {}


def aff_adjacencies() -> Tuple[Tensor, Tensor]:
    """A1 (5,9,9) and A2 (5,3,3) of AffEncoder (net/multimodal_context_net_v2.py:99-115)."""
    if 'A1' not in _A_CACHE:
        _A_CACHE['A1'] = torch.tensor(spatial_adjacency(N_DIR_VECS, DIR_EDGE_PAIRS, 2), dtype=torch.float32)
        _A_CACHE['A2'] = torch.tensor(spatial_adjacency(N_BODY_PARTS, BODY_PARTS_EDGE_PAIRS, 2),
                                      dtype=torch.float32)
    return _A_CACHE['A1'], _A_CACHE['A2']


# ----------------------------------------------------------------------------------------------
# explicit randomness
# ----------------------------------------------------------------------------------------------
class Noise:
    """Source of every random tensor of one forward pass.

    ``Noise(None)``            -> draw from torch's global CPU generator (baseline timing).
    ``Noise({'name': t, ...})``-> return pinned tensors by call-site name (parity tests); a missing
                                  name raises, so a test can never silently use fresh randomness.
    ``Noise('off')``           -> dropout is the identity (keep mask of ones, scale 1) and eps = 0.
    Call-site names: '<prefix>emb_drop', '<prefix>tcn.<i>.drop1|drop2', '<prefix>gru.drop<l>', 'eps'.
    Dropout tensors are *keep masks already multiplied by 1/(1-p)*.
    """

    def __init__(self, pinned=None):
        self.pinned = pinned
        self.drawn: Dict[str, Tensor] = {}

    def dropout(self, name: str, x: Tensor, p: float) -> Tensor:
        if p <= 0.0:
            return x
        if self.pinned == 'off':
            return x
        if self.pinned is None:
            m = torch.bernoulli(torch.full_like(x, 1.0 - p)) / (1.0 - p)
        else:
            m = self.pinned[name]
        self.drawn[name] = m
        return x * m

    def keep(self, name: str, p: float) -> Optional[Tensor]:
        """The elements the dropout at ``name`` will keep (bool), when that is known before it is applied (pinned masks);
        None = all of them / not known (fresh randomness is never combined with a sign replay)."""
        if p <= 0.0 or not isinstance(self.pinned, dict):
            return None
        return self.pinned[name] != 0

    def normal(self, name: str, like: Tensor) -> Tensor:
        if self.pinned == 'off':
            return torch.zeros_like(like)
        if self.pinned is None:
            e = torch.randn_like(like)
        else:
            e = self.pinned[name]
        self.drawn[name] = e
        return e


# ----------------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------------
def _bn(sd: SD, p: str, x: Tensor, training: bool) -> Tensor:
    """nn.BatchNorm{1,2}d semantics (eps 1e-5, momentum 0.1, biased var for normalising,
    unbiased for the running estimate); updates running stats in ``sd`` in place when training."""
    if training and (p + 'num_batches_tracked') in sd:
        sd[p + 'num_batches_tracked'] += 1
    return F.batch_norm(x, sd[p + 'running_mean'], sd[p + 'running_var'], sd[p + 'weight'], sd[p + 'bias'],
                        training, 0.1, 1e-5)


# ---- branch decisions of the piecewise-linear activations --------------------------------------------------------------
# An activation input within rounding distance of its kink may legitimately land on different sides in two correct
# implementations (another summation order is enough), and ONE flipped element moves a gradient by far more than any
# rounding error.  ``with use_signs({site: bool tensor}):`` makes every named ReLU / LeakyReLU site take the branches recorded
# there (True = the positive side) instead of deciding from its own input -- the parity tests record them from the product's
# outputs, so both sides differentiate the SAME piecewise-linear function and gradients can be compared strictly.
# Site names: '<BatchNorm prefix>' for BN + activation pairs (e.g. 'audio_encoder.batch_norm1.'), '<prefix>linear1.',
# '<prefix>tcn.<i>.relu1|relu2|relu3', '<prefix>st_gcn<k>.tcn.0.' / '<prefix>st_gcn<k>.out', 'out.1.'.
# The replay is AUDITED, so that it cannot hide a wrong branch: at every replayed site the oracle also takes its own decision
# (x > 0) and files, in ``use_signs(...).audit[site]``, how many LIVE elements (not zeroed by the dropout that follows the
# site) were replayed against it and how far from the kink the oracle's own pre-activation is at the worst of them, relative
# to the site's largest |x|.  A test asserts that these are a handful of elements within rounding distance of zero
# (``assert_benign``) -- a product that takes the wrong side of a LARGE pre-activation fails there, replayed or not.
_SIGNS: List[Optional['use_signs']] = [None]


class use_signs:
    def __init__(self, signs: Optional[Dict[str, Tensor]]):
        self.signs = signs
        self.used: List[str] = []
        self.audit: Dict[str, Dict[str, float]] = {}

    def __enter__(self):
        self.prev = _SIGNS[0]
        _SIGNS[0] = self
        return self

    def __exit__(self, *a):
        _SIGNS[0] = self.prev

    def assert_benign(self, base: int, density: float, max_rel: float, what: str = '') -> Dict[str, float]:
        """Every replayed site: at most ``base + density * live elements`` live elements decided differently from the oracle's
        own x > 0, each with |x| <= max_rel * max|x| of its site.  Returns the totals (for a test's log line)."""
        return audit_benign(self.audit, base, density, max_rel, what)


def audit_benign(audit: Dict[str, Dict[str, float]], base: int, density: float, max_rel: float,
                 what: str = '') -> Dict[str, float]:
    bad = {k: v for k, v in audit.items() if v['flipped'] > base + density * v['live'] or v['worst_rel'] > max_rel}
    assert not bad, (what, 'replayed branch decisions that are NOT a few elements within rounding distance of the kink', bad)
    return dict(sites=len(audit), elements=sum(v['live'] for v in audit.values()),
                flipped=sum(v['flipped'] for v in audit.values()),
                worst_rel=max([v['worst_rel'] for v in audit.values()] + [0.0]))


def _act(x: Tensor, slope: float, name: str, live: Optional[Tensor] = None) -> Tensor:
    """ReLU (slope 0) / LeakyReLU at the named site.  ``live`` (bool, like x; None = all): elements the dropout behind the
    site keeps -- where it drops, the branch is immaterial (and the product's recorded decision, read off its post-dropout
    output, is meaningless), so the audit of a replay skips them."""
    ctx = _SIGNS[0]
    if ctx is not None and ctx.signs is not None and name in ctx.signs:
        m = ctx.signs[name]
        assert m.shape == x.shape and m.dtype == torch.bool, (name, m.shape, x.shape)
        ctx.used.append(name)
        with torch.no_grad():
            own = x > 0
            diff = own != m
            if live is not None:
                diff = diff & live
            n = int(diff.sum())
            top = float(x.abs().max())
            ctx.audit[name] = dict(live=int(x.numel() if live is None else live.sum()), flipped=n,
                                   worst_rel=(float(x[diff].abs().max()) / max(top, 1e-30)) if n else 0.0)
        return torch.where(m, x, x * slope)
    return F.leaky_relu(x, slope) if slope != 0.0 else F.relu(x)


def wav_encoder(sd: SD, p: str, wav: Tensor, training: bool) -> Tensor:
    """WavEncoder (net/multimodal_context_net_v2.py:14-33): (B, n_samples) -> (B, frames, 32)."""
    fe = p + 'feat_extractor.'
    x = wav.unsqueeze(1)
    x = F.conv1d(x, sd[fe + '0.weight'], sd[fe + '0.bias'], stride=5, padding=1600)
    x = _act(_bn(sd, fe + '1.', x, training), 0.3, fe + '1.')
    x = F.conv1d(x, sd[fe + '3.weight'], sd[fe + '3.bias'], stride=6)
    x = _act(_bn(sd, fe + '4.', x, training), 0.3, fe + '4.')
    x = F.conv1d(x, sd[fe + '6.weight'], sd[fe + '6.bias'], stride=6)
    x = _act(_bn(sd, fe + '7.', x, training), 0.3, fe + '7.')
    x = F.conv1d(x, sd[fe + '9.weight'], sd[fe + '9.bias'], stride=6)
    return x.transpose(1, 2)


def mfcc_encoder(sd: SD, p: str, mfcc: Tensor, training: bool) -> Tensor:
    """MFCCEncoder (:36-58): (B, num_mfcc, mfcc_length) -> (B, time_steps, 32).
    MFCC *frames* are the conv channels; the coefficient index is the conv axis."""
    x = mfcc.permute(0, 2, 1)
    for i, pad in ((1, 2), (2, 2), (3, 1), (4, 1)):
        x = F.conv1d(x, sd[f'{p}conv{i}.weight'], sd[f'{p}conv{i}.bias'], padding=pad)
        x = _act(_bn(sd, f'{p}batch_norm{i}.', x, training), 0.3, f'{p}batch_norm{i}.')
    return _act(F.linear(x, sd[p + 'linear1.weight'], sd[p + 'linear1.bias']), 0.3, p + 'linear1.')


def weight_norm_weight(g: Tensor, v: Tensor) -> Tensor:
    """torch.nn.utils.weight_norm (dim=0): w = g * v / ||v|| with the norm over all dims but 0."""
    return g * v / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))


def temporal_block(sd: SD, p: str, x: Tensor, dilation: int, training: bool, drop_p: float,
                   noise: Noise, name: str) -> Tensor:
    """TemporalBlock (net/tcn.py:16-46), kernel 2: causal dilated conv -> chomp -> ReLU -> dropout, twice,
    then ReLU(out + x).  (No downsample: n_inputs == n_outputs on this path, tcn.py:33.)"""
    out = x
    for j, tag in ((1, 'conv1'), (2, 'conv2')):
        w = weight_norm_weight(sd[f'{p}{tag}.weight_g'], sd[f'{p}{tag}.weight_v'])
        k = w.shape[2]
        pad = (k - 1) * dilation
        out = F.conv1d(out, w, sd[f'{p}{tag}.bias'], padding=pad, dilation=dilation)
        out = _act(out[:, :, :-pad], 0.0, f'{name}.relu{j}',
                   live=noise.keep(f'{name}.drop{j}', drop_p) if training else None)
        if training:
            out = noise.dropout(f'{name}.drop{j}', out, drop_p)
    if (p + 'downsample.weight') in sd:
        x = F.conv1d(x, sd[p + 'downsample.weight'], sd[p + 'downsample.bias'])
    return _act(out + x, 0.0, f'{name}.relu3')


def text_encoder_tcn(sd: SD, p: str, ids: Tensor, training: bool, drop_p: float, noise: Noise,
                     emb_drop_p: float = 0.1) -> Tensor:
    """TextEncoderTCN (net/multimodal_context_net_v2.py:61-91): (B,T) int64 -> (B,T,32)."""
    emb = F.embedding(ids, sd[p + 'embedding.weight'])
    if training:
        emb = noise.dropout(p + 'emb_drop', emb, emb_drop_p)
    y = emb.transpose(1, 2)
    i = 0
    while f'{p}tcn.network.{i}.conv1.weight_g' in sd:
        y = temporal_block(sd, f'{p}tcn.network.{i}.', y, 2 ** i, training, drop_p, noise, f'{p}tcn.{i}')
        i += 1
    y = F.linear(y.transpose(1, 2), sd[p + 'decoder.weight'], sd[p + 'decoder.bias'])
    return y.contiguous()


def st_graph_conv(sd: SD, p: str, x: Tensor, A: Tensor, training: bool) -> Tensor:
    """STGraphConv (net/utils/tgcn.py:133-218) with ConvTemporalGraphical (:15-71).
    x (N,C,T,V).  The residual branch is always conv1x1 + BN (tgcn.py:195 compares a tuple to 1)."""
    res = F.conv2d(x, sd[p + 'residual.0.weight'], sd[p + 'residual.0.bias'])
    res = _bn(sd, p + 'residual.1.', res, training)
    wg = sd[p + 'gcn.conv.weight']
    kt = wg.shape[2]
    g = F.conv2d(x, wg, sd[p + 'gcn.conv.bias'], padding=(kt // 2, 0))
    n, kc, t, v = g.shape
    K = A.shape[0]
    g = torch.einsum('nkctv,kvw->nctw', g.view(n, K, kc // K, t, v), A)
    h = _act(_bn(sd, p + 'tcn.0.', g, training), 0.0, p + 'tcn.0.')
    wt = sd[p + 'tcn.2.weight']
    h = F.conv2d(h, wt, sd[p + 'tcn.2.bias'], padding=(wt.shape[2] // 2, wt.shape[3] // 2))
    h = _bn(sd, p + 'tcn.3.', h, training)
    return _act(h + res, 0.01, p + 'out')


def aff_encoder(sd: SD, p: str, poses: Tensor, training: bool) -> Tensor:
    """AffEncoder (net/multimodal_context_net_v2.py:94-175): (B,T,27) -> (B,T,8)."""
    A1, A2 = aff_adjacencies()
    n, t, _ = poses.shape
    x = poses.reshape(n, t, N_DIR_VECS, 3).permute(0, 3, 1, 2)                      # (n,3,t,9)
    f1 = st_graph_conv(sd, p + 'st_gcn1.', x, A1, training)                         # (n,16,t,9)
    c1 = f1.shape[1]
    f1 = _bn(sd, p + 'batch_norm1.', f1.permute(0, 1, 3, 2).reshape(n, c1 * N_DIR_VECS, t), training)
    f1 = f1.view(n, c1, N_DIR_VECS, t)                                              # (n,c,v,t)
    # regroup the 9 edges into 3 body parts; channel index c*3 + j (:161-167)
    parts = [f1[:, :, idx, :].reshape(n, c1 * len(idx), t) for idx in BODY_PARTS_EDGE_IDX]
    x2 = torch.stack(parts, dim=3)                                                  # (n,48,t,3)
    f2 = st_graph_conv(sd, p + 'st_gcn2.', x2, A2, training)                        # (n,16,t,3)
    c2 = f2.shape[1]
    f2 = _bn(sd, p + 'batch_norm2.', f2.permute(0, 1, 3, 2).reshape(n, c2 * N_BODY_PARTS, t), training)
    x3 = F.conv1d(f2, sd[p + 'conv3.weight'], sd[p + 'conv3.bias'], padding=sd[p + 'conv3.weight'].shape[2] // 2)
    x3 = _act(_bn(sd, p + 'batch_norm3.', x3, training), 0.01, p + 'batch_norm3.')
    x4 = F.conv1d(x3, sd[p + 'conv4.weight'], sd[p + 'conv4.bias'], padding=sd[p + 'conv4.weight'].shape[2] // 2)
    x4 = _act(_bn(sd, p + 'batch_norm4.', x4, training), 0.01, p + 'batch_norm4.')
    return x4.permute(0, 2, 1)


def gru_cell_step(x_proj: Tensor, h: Tensor, w_hh: Tensor, b_hh: Tensor) -> Tensor:
    """One GRU cell step, PyTorch gate order (r, z, n):
    n = tanh(W_in x + b_in + r * (W_hn h + b_hn)); h' = (1 - z) * n + z * h."""
    H = h.shape[1]
    gh = F.linear(h, w_hh, b_hh)
    r = torch.sigmoid(x_proj[:, :H] + gh[:, :H])
    z = torch.sigmoid(x_proj[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(x_proj[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def gru(sd: SD, p: str, x: Tensor, training: bool, drop_p: float, noise: Noise, name: str) -> Tensor:
    """nn.GRU(batch_first, bidirectional) restated cell by cell: (B,T,I) -> (B,T,2H).
    Inter-layer dropout on the concatenated output of every layer but the last (train mode)."""
    B, T, _ = x.shape
    layer = 0
    while f'{p}weight_ih_l{layer}' in sd:
        outs = []
        for suffix, order in (('', range(T)), ('_reverse', range(T - 1, -1, -1))):
            w_ih, w_hh = sd[f'{p}weight_ih_l{layer}{suffix}'], sd[f'{p}weight_hh_l{layer}{suffix}']
            b_ih, b_hh = sd[f'{p}bias_ih_l{layer}{suffix}'], sd[f'{p}bias_hh_l{layer}{suffix}']
            H = w_hh.shape[1]
            proj = F.linear(x, w_ih, b_ih)
            h = x.new_zeros(B, H)
            ys: List[Optional[Tensor]] = [None] * T
            for t in order:
                h = gru_cell_step(proj[:, t], h, w_hh, b_hh)
                ys[t] = h
            outs.append(torch.stack(ys, dim=1))
        x = torch.cat(outs, dim=2)
        layer += 1
        if training and f'{p}weight_ih_l{layer}' in sd:
            x = noise.dropout(f'{name}.drop{layer - 1}', x, drop_p)
    return x


def gru_fast(sd: SD, p: str, x: Tensor, training: bool, drop_p: float) -> Tensor:
    """Same op through ATen's fused ``gru`` (what the reference dispatches, nn.GRU); used for the
    CPU-baseline timing only (dropout drawn by ATen).  Equality with ``gru`` is tested."""
    flat, layer = [], 0
    while f'{p}weight_ih_l{layer}' in sd:
        for suffix in ('', '_reverse'):
            flat += [sd[f'{p}weight_ih_l{layer}{suffix}'], sd[f'{p}weight_hh_l{layer}{suffix}'],
                     sd[f'{p}bias_ih_l{layer}{suffix}'], sd[f'{p}bias_hh_l{layer}{suffix}']]
        layer += 1
    H = flat[1].shape[1]
    h0 = x.new_zeros(2 * layer, x.shape[0], H)
    y, _ = torch._VF.gru(x, h0, flat, True, layer, drop_p if training else 0.0, training, True, True)
    return y


def re_parametrize(mu: Tensor, log_var: Tensor, noise: Noise) -> Tensor:
    """net/embedding_net.py:10-13 -- noise is drawn in eval mode too."""
    std = torch.exp(0.5 * log_var)
    return mu + noise.normal('eps', std) * std


@dataclass
class ModelCfg:
    """The fields of the reference's args namespace the nets read (parse_args.py:39-63)."""
    n_poses: int = 34
    n_pre_poses: int = 4
    hidden_size: int = 300
    hidden_size_s2eg: int = 300
    n_layers: int = 4
    dropout_prob: float = 0.3
    input_context: str = 'both'
    freeze_wordembed: bool = False


def _speaker_z(sd: SD, vid: Tensor, noise: Noise):
    assert vid is not None                 # net/multimodal_context_net_v2.py:511 (what z_type 'random' runs into, processor_v2.py:906)
    z = F.embedding(vid, sd['speaker_embedding.0.weight'])
    z = F.linear(z, sd['speaker_embedding.1.weight'], sd['speaker_embedding.1.bias'])
    mu = F.linear(z, sd['speaker_mu.weight'], sd['speaker_mu.bias'])
    log_var = F.linear(z, sd['speaker_log_var.weight'], sd['speaker_log_var.bias'])
    return re_parametrize(mu, log_var, noise), mu, log_var


def _decode(sd: SD, cfg_drop: float, in_data: Tensor, training: bool, noise: Noise, out_slope: float,
            fast: bool):
    if fast and noise.pinned is None:
        y = gru_fast(sd, 'gru.', in_data, training, cfg_drop)
    else:
        y = gru(sd, 'gru.', in_data, training, cfg_drop, noise, 'gru')
    H = y.shape[2] // 2
    y = y[:, :, :H] + y[:, :, H:]
    y = F.linear(y, sd['out.0.weight'], sd['out.0.bias'])
    y = _act(y, out_slope, 'out.1.') if out_slope != 1.0 else y
    return F.linear(y, sd['out.2.weight'], sd['out.2.bias'])


def pose_generator(sd: SD, cfg: ModelCfg, pre_seq: Tensor, in_text: Tensor, in_mfcc: Tensor, vid: Tensor,
                   training: bool, noise: Noise, fast: bool = False):
    """PoseGenerator.forward (net/multimodal_context_net_v2.py:492-546), input_context 'both', speaker z."""
    audio = mfcc_encoder(sd, 'audio_encoder.', in_mfcc, training)
    text = text_encoder_tcn(sd, 'text_encoder.', in_text, training, cfg.dropout_prob, noise)
    assert audio.shape[1] == text.shape[1], 'Audio and text features must have the same number of time steps.'
    z, mu, log_var = _speaker_z(sd, vid, noise)
    pre = aff_encoder(sd, 'aff_encoder.', pre_seq[..., :-1], training)
    in_data = torch.cat((pre, audio, text, z.unsqueeze(1).expand(-1, pre.shape[1], -1)), dim=2)
    out = _decode(sd, cfg.dropout_prob, in_data, training, noise, 0.01, fast)
    return out, z, mu, log_var


def pose_generator_abl_audio(sd: SD, cfg: ModelCfg, pre_seq, in_text, in_audio, vid, training: bool, noise: Noise,
                             fast: bool = False):
    """_abl_audio.PoseGenerator (net/multimodal_context_net_v2_abl_audio.py:413-521): WavEncoder audio branch."""
    audio = wav_encoder(sd, 'audio_encoder.', in_audio, training)
    text = text_encoder_tcn(sd, 'text_encoder.', in_text, training, cfg.dropout_prob, noise)
    assert audio.shape[1] == text.shape[1]
    z, mu, log_var = _speaker_z(sd, vid, noise)
    pre = aff_encoder(sd, 'aff_encoder.', pre_seq[..., :-1], training)
    in_data = torch.cat((pre, audio, text, z.unsqueeze(1).expand(-1, pre.shape[1], -1)), dim=2)
    out = _decode(sd, cfg.dropout_prob, in_data, training, noise, 0.01, fast)
    return out, z, mu, log_var


def pose_generator_abl_aff(sd: SD, cfg: ModelCfg, pre_seq, in_text, in_mfcc, vid, training: bool, noise: Noise,
                           fast: bool = False):
    """_abl_aff.PoseGenerator.forward (net/multimodal_context_net_v2_abl_aff.py:336-392): no affective encoder, the raw
    (pose_dim + 1)-column seed sequence is the GRU's pose input."""
    audio = mfcc_encoder(sd, 'audio_encoder.', in_mfcc, training)
    text = text_encoder_tcn(sd, 'text_encoder.', in_text, training, cfg.dropout_prob, noise)
    assert audio.shape[1] == text.shape[1], 'Audio and text features must have the same number of time steps.'
    z, mu, log_var = _speaker_z(sd, vid, noise)
    in_data = torch.cat((pre_seq, audio, text, z.unsqueeze(1).expand(-1, pre_seq.shape[1], -1)), dim=2)
    out = _decode(sd, cfg.dropout_prob, in_data, training, noise, 0.01, fast)
    return out, z, mu, log_var


def pose_generator_trimodal(sd: SD, cfg: ModelCfg, pre_seq, in_text, in_audio, vid, training: bool, noise: Noise,
                            fast: bool = False):
    """PoseGeneratorTriModal.forward (:287-343): raw pre_seq into the GRU; ``nn.LeakyReLU(True)`` in
    ``out`` has negative_slope == 1.0, i.e. the identity (:285)."""
    audio = wav_encoder(sd, 'audio_encoder.', in_audio, training)
    text = text_encoder_tcn(sd, 'text_encoder.', in_text, training, cfg.dropout_prob, noise)
    assert audio.shape[1] == text.shape[1]
    z, mu, log_var = _speaker_z(sd, vid, noise)
    in_data = torch.cat((pre_seq, audio, text, z.unsqueeze(1).expand(-1, pre_seq.shape[1], -1)), dim=2)
    out = _decode(sd, cfg.dropout_prob, in_data, training, noise, 1.0, fast)
    return out, z, mu, log_var


# ---------------------------------------------------------------------------------------------------------------------
# sliding-window synthesis (Processor.render_clip, processor_v2.py:1173-1330, without rendering / fade-out)
# ---------------------------------------------------------------------------------------------------------------------
def words_in_time_range(word_list, start_time: float, end_time: float):
    """DataPreprocessor.get_words_in_time_range (utils/data_preprocessor.py): words overlapping [start, end)."""
    words = []
    for word in word_list:
        _, word_s, word_e = word[0], word[1], word[2]
        if word_s >= end_time:
            break
        if word_e <= start_time:
            continue
        words.append(word)
    return words


def synthesis_windows(n_samples: int, sample_rate: int, n_poses: int, n_pre: int, fps: float, unit_time=None):
    """The window plan of render_clip (:1199-1216, :1233-1246): [(start_time, end_time, audio_start)], audio window length."""
    clip_length = n_samples / sample_rate
    if unit_time is None:
        unit_time = n_poses / fps
    stride_time = (n_poses - n_pre) / fps
    num = 1 if clip_length < unit_time else math.ceil((clip_length - unit_time) / stride_time) + 1
    audio_sample_length = int(unit_time * sample_rate)
    plan = []
    for k in range(num):
        t0 = min(k * stride_time, clip_length)
        t1 = min(t0 + unit_time, clip_length)
        if t0 >= t1:
            continue
        plan.append((t0, t1, math.floor(t0 / clip_length * n_samples)))
    return plan, audio_sample_length


def window_text(clip_words, t0: float, t1: float, n_frames: int, word_index) -> Tensor:
    """:1255-1271: every word of the window lands on the frame its start time falls into (PAD = 0 elsewhere)."""
    ext = np.zeros(n_frames)
    frame_duration = (t1 - t0) / n_frames
    for word in words_in_time_range(clip_words, t0, t1):
        ext[max(0, int(np.floor((word[1] - t0) / frame_duration)))] = word_index(word[0])
    return torch.LongTensor(ext).unsqueeze(0)


def crossfade_append(out_list: list, out_seq: np.ndarray, n_pre: int) -> None:
    """:1296-1322: the previous window gives up its last n_pre frames, which are blended into the first n_pre of the new
    one with weights (n - j) / (n + 1) and (j + 1) / (n + 1)."""
    if len(out_list) > 0:
        last_poses = out_list[-1][-n_pre:]
        out_list[-1] = out_list[-1][:-n_pre]
        n = len(last_poses)
        for j in range(n):
            out_seq[j] = last_poses[j] * (n - j) / (n + 1) + out_seq[j] * (j + 1) / (n + 1)
    out_list.append(out_seq)


def synthesize_clip(sdG: SD, sdT: SD, cfg: ModelCfg, seed_seq, clip_audio: np.ndarray, sample_rate: int, clip_words,
                    mfcc_windows, vid: int, eps, word_index, fps: float = 15.0, pose_dim: int = 27):
    """render_clip's synthesis loop: per window the tri-modal baseline and the s2ag generator run at batch 1, the last
    n_pre output frames seed the next window, consecutive windows are cross-faded.  ``eps``: z noise per forward in
    call order (tri-modal, s2ag per window); ``mfcc_windows[k]``: the MFCC image of window k (librosa is outside the
    path).  Returns (out_dir_vec_trimodal, out_dir_vec), each (W * (n_poses - n_pre) + n_pre, pose_dim)."""
    n_frames, n_pre = cfg.n_poses, cfg.n_pre_poses
    plan, alen = synthesis_windows(len(clip_audio), sample_rate, n_frames, n_pre, fps)
    pre_t = torch.zeros(1, n_frames, pose_dim + 1)
    pre_g = torch.zeros(1, n_frames, pose_dim + 1)
    seed = torch.as_tensor(np.asarray(seed_seq)[:n_pre], dtype=torch.float32)
    for pre in (pre_t, pre_g):
        pre[0, :n_pre, :-1] = seed
        pre[0, :n_pre, -1] = 1
    vid_t = torch.LongTensor([vid])
    out_t = out_g = None
    list_t, list_g = [], []
    for k, (t0, t1, a0) in enumerate(plan):
        win = np.asarray(clip_audio[a0:a0 + alen], dtype=np.float32)
        if len(win) < alen:
            win = np.pad(win, (0, alen - len(win)), 'constant')
        in_audio = torch.from_numpy(win).unsqueeze(0)
        in_mfcc = torch.as_tensor(np.asarray(mfcc_windows[k]), dtype=torch.float32).unsqueeze(0)
        in_text = window_text(clip_words, t0, t1, n_frames, word_index)
        if k > 0:
            pre_t[0, :n_pre, :-1] = out_t[0, -n_pre:]
            pre_t[0, :n_pre, -1] = 1
            pre_g[0, :n_pre, :-1] = out_g[0, -n_pre:]
            pre_g[0, :n_pre, -1] = 1
        out_t = pose_generator_trimodal(sdT, cfg, pre_t, in_text, in_audio, vid_t, False,
                                        Noise({'eps': torch.as_tensor(eps[2 * k])}))[0]
        out_g = pose_generator(sdG, cfg, pre_g, in_text, in_mfcc, vid_t, False,
                               Noise({'eps': torch.as_tensor(eps[2 * k + 1])}))[0]
        crossfade_append(list_t, out_t[0].detach().numpy().copy(), n_pre)
        crossfade_append(list_g, out_g[0].detach().numpy().copy(), n_pre)
    return np.vstack(list_t), np.vstack(list_g)


def aff_discriminator(sd: SD, poses: Tensor, training: bool, noise: Noise, fast: bool = False) -> Tensor:
    """AffDiscriminator.forward (:568-585): GRU dropout is hard-wired to 0.3 (:558)."""
    feat = aff_encoder(sd, 'aff_encoder.', poses, training)
    if fast and noise.pinned is None:
        y = gru_fast(sd, 'gru.', feat, training, 0.3)
    else:
        y = gru(sd, 'gru.', feat, training, 0.3, noise, 'gru')
    H = y.shape[2] // 2
    y = y[:, :, :H] + y[:, :, H:]
    y = F.linear(y, sd['out.weight'], sd['out.bias']).squeeze(2)
    return torch.sigmoid(F.linear(y, sd['out2.weight'], sd['out2.bias']))


def conv_discriminator(sd: SD, poses: Tensor, training: bool, noise: Noise, fast: bool = False) -> Tensor:
    """ConvDiscriminatorTriModal / ConvDiscriminator (:390-435): three valid k=3 convs; the two
    ``nn.LeakyReLU(True)`` are identities (slope 1.0, :399,:402)."""
    x = poses.transpose(1, 2)
    x = F.conv1d(x, sd['pre_conv.0.weight'], sd['pre_conv.0.bias'])
    x = _bn(sd, 'pre_conv.1.', x, training)
    x = F.conv1d(x, sd['pre_conv.3.weight'], sd['pre_conv.3.bias'])
    x = _bn(sd, 'pre_conv.4.', x, training)
    x = F.conv1d(x, sd['pre_conv.6.weight'], sd['pre_conv.6.bias']).transpose(1, 2)
    if fast and noise.pinned is None:
        y = gru_fast(sd, 'gru.', x, training, 0.3)
    else:
        y = gru(sd, 'gru.', x, training, 0.3, noise, 'gru')
    H = y.shape[2] // 2
    y = y[:, :, :H] + y[:, :, H:]
    y = F.linear(y, sd['out.weight'], sd['out.bias']).squeeze(2)
    return torch.sigmoid(F.linear(y, sd['out2.weight'], sd['out2.bias']))


# ----------------------------------------------------------------------------------------------
# evaluation metrics: pose auto-encoder (net/embedding_net.py) + Frechet gesture distance
# ----------------------------------------------------------------------------------------------
def _bn1(sd: SD, p: str, x: Tensor, training: bool) -> Tensor:
    """BatchNorm1d over (B, C) or (B, C, L) with the running estimates living in ``sd`` (updated in place)."""
    return F.batch_norm(x, sd[p + 'running_mean'], sd[p + 'running_var'], sd[p + 'weight'], sd[p + 'bias'], training,
                        0.1, 1e-5)


def pose_encoder_conv(sd: SD, p: str, poses: Tensor, training: bool) -> Tuple[Tensor, Tensor]:
    """PoseEncoderConv.forward (net/embedding_net.py:42-83) without the re-parametrisation: (mu, log_var)."""
    x = poses.transpose(1, 2)
    for i, stride in ((0, 1), (1, 1), (2, 2)):
        x = F.conv1d(x, sd[f'{p}net.{i}.0.weight'], sd[f'{p}net.{i}.0.bias'], stride=stride)
        x = F.leaky_relu(_bn1(sd, f'{p}net.{i}.1.', x, training), 0.2)
    x = F.conv1d(x, sd[p + 'net.3.weight'], sd[p + 'net.3.bias']).flatten(1)
    x = _bn1(sd, p + 'out_net.1.', F.linear(x, sd[p + 'out_net.0.weight'], sd[p + 'out_net.0.bias']), training)
    x = _bn1(sd, p + 'out_net.4.', F.linear(x, sd[p + 'out_net.3.weight'], sd[p + 'out_net.3.bias']), training)
    x = F.linear(x, sd[p + 'out_net.6.weight'], sd[p + 'out_net.6.bias'])      # nn.LeakyReLU(True) = slope 1 = identity
    return (F.linear(x, sd[p + 'fc_mu.weight'], sd[p + 'fc_mu.bias']),
            F.linear(x, sd[p + 'fc_log_var.weight'], sd[p + 'fc_log_var.bias']))


def pose_decoder_conv(sd: SD, p: str, feat: Tensor, training: bool) -> Tensor:
    """PoseDecoderConv.forward (:165-217), length 34, no pre-pose branch."""
    x = _bn1(sd, p + 'pre_net.1.', F.linear(feat, sd[p + 'pre_net.0.weight'], sd[p + 'pre_net.0.bias']), training)
    x = F.linear(x, sd[p + 'pre_net.3.weight'], sd[p + 'pre_net.3.bias']).view(feat.shape[0], 4, -1)
    for i in (0, 3):
        x = F.conv_transpose1d(x, sd[f'{p}net.{i}.weight'], sd[f'{p}net.{i}.bias'])
        x = F.leaky_relu(_bn1(sd, f'{p}net.{i + 1}.', x, training), 0.2)
    x = F.conv1d(x, sd[p + 'net.6.weight'], sd[p + 'net.6.bias'])
    return F.conv1d(x, sd[p + 'net.7.weight'], sd[p + 'net.7.bias']).transpose(1, 2)


def embedding_net_pose(sd: SD, poses: Tensor, training: bool, eps: Optional[Tensor] = None):
    """EmbeddingNet.forward in 'pose' mode (:277-301): (poses_feat, mu, log_var, out_poses); ``eps`` = variational."""
    mu, log_var = pose_encoder_conv(sd, 'pose_encoder.', poses, training)
    z = mu if eps is None else mu + eps * torch.exp(0.5 * log_var)
    return z, mu, log_var, pose_decoder_conv(sd, 'decoder.', z, training)


def embedding_net_shapes(pose_dim: int = 27, n_frames: int = 34) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    e, d = 'pose_encoder.', 'decoder.'
    for i, (ci, co, k) in enumerate(((pose_dim, 32, 3), (32, 64, 3), (64, 64, 4))):
        s[f'{e}net.{i}.0.weight'], s[f'{e}net.{i}.0.bias'] = (co, ci, k), (co,)
        s.update(_bn_entries(f'{e}net.{i}.1.', co))
    s[e + 'net.3.weight'], s[e + 'net.3.bias'] = (32, 64, 3), (32,)
    flat = 32 * ((((n_frames - 2) - 2) - 4) // 2 + 1 - 2)
    for i, (ci, co) in zip((0, 3, 6), ((flat, 256), (256, 128), (128, 32))):
        s[f'{e}out_net.{i}.weight'], s[f'{e}out_net.{i}.bias'] = (co, ci), (co,)
    s.update(_bn_entries(e + 'out_net.1.', 256))
    s.update(_bn_entries(e + 'out_net.4.', 128))
    for n in ('fc_mu', 'fc_log_var'):
        s[f'{e}{n}.weight'], s[f'{e}{n}.bias'] = (32, 32), (32,)
    s[d + 'pre_net.0.weight'], s[d + 'pre_net.0.bias'] = (64, 32), (64,)
    s.update(_bn_entries(d + 'pre_net.1.', 64))
    s[d + 'pre_net.3.weight'], s[d + 'pre_net.3.bias'] = (4 * n_frames, 64), (4 * n_frames,)
    s[d + 'net.0.weight'], s[d + 'net.0.bias'] = (4, 32, 3), (32,)
    s[d + 'net.3.weight'], s[d + 'net.3.bias'] = (32, 32, 3), (32,)
    s.update(_bn_entries(d + 'net.1.', 32))
    s.update(_bn_entries(d + 'net.4.', 32))
    s[d + 'net.6.weight'], s[d + 'net.6.bias'] = (32, 32, 3), (32,)
    s[d + 'net.7.weight'], s[d + 'net.7.bias'] = (pose_dim, 32, 3), (pose_dim,)
    return s


def frechet_distance(mu1, sigma1, mu2, sigma2, eps: float = 1e-6) -> float:
    """EmbeddingSpaceEvaluator.calculate_frechet_distance (net/embedding_space_evaluator.py:105-156)."""
    from scipy import linalg
    diff = np.asarray(mu1) - np.asarray(mu2)
    root, _ = linalg.sqrtm(np.asarray(sigma1).dot(np.asarray(sigma2)), disp=False)
    if not np.isfinite(root).all():
        off = np.eye(len(diff)) * eps
        root = linalg.sqrtm((sigma1 + off).dot(sigma2 + off))
    if np.iscomplexobj(root):
        if not np.allclose(np.diagonal(root).imag, 0, atol=1e-3):
            raise ValueError('Imaginary component')
        root = root.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(root))


def fgd_scores(generated_feats: np.ndarray, real_feats: np.ndarray) -> Tuple[float, float]:
    """EmbeddingSpaceEvaluator.get_scores (:74-103): (frechet_dist, mean L1 feature distance)."""
    try:
        fd = frechet_distance(generated_feats.mean(0), np.cov(generated_feats, rowvar=False), real_feats.mean(0),
                              np.cov(real_feats, rowvar=False))
    except ValueError:
        fd = 1e+10
    return fd, float(np.abs(real_feats - generated_feats).sum(1).mean())


# ----------------------------------------------------------------------------------------------
# the GAN step  (processor_v2.py:776-957) and Adam (processor_v2.py:215-220)
# ----------------------------------------------------------------------------------------------
# skeleton of utils/ted_db_utils.py:14-16 (parent, child, bone length)
DIR_VEC_PAIRS = ((0, 1, 0.26), (1, 2, 0.18), (2, 3, 0.14), (1, 4, 0.22), (4, 5, 0.36), (5, 6, 0.33), (1, 7, 0.22),
                 (7, 8, 0.36), (8, 9, 0.33))


def dir_vec_to_pose(vec: np.ndarray) -> np.ndarray:
    """utils/ted_db_utils.py:81-102 (batch, seq, 27) branch: joint[child] = joint[parent] + length * direction, ten
    joints, root at the origin; float64 accumulation of float32 products, as numpy does upstream."""
    vec = np.asarray(vec).reshape(vec.shape[:-1] + (-1, 3))
    joints = np.zeros(vec.shape[:2] + (10, 3))
    for j, (par, chi, length) in enumerate(DIR_VEC_PAIRS):
        joints[:, :, chi] = joints[:, :, par] + length * vec[:, :, j]
    return joints


def push_samples_metrics(out_dir_vec: Tensor, target: Tensor, mean_dir_vec, n_pre: int) -> Tuple[float, float, float]:
    """The three meter values of Processor.push_samples (processor_v2.py:738-774): L1 of the direction vectors (:746),
    MAE of the joint coordinates behind the seed poses (:753-766), acceleration difference (:768-771).  The inputs are
    left untouched (upstream adds the mean IN PLACE to host copies of them)."""
    l1 = float(F.l1_loss(out_dir_vec, target))
    mean = np.array(mean_dir_vec).squeeze()
    o = out_dir_vec.detach().cpu().numpy().copy()
    o += mean
    t = target.detach().cpu().numpy().copy()
    t += mean
    po, pt = dir_vec_to_pose(o), dir_vec_to_pose(t)
    mae = float(np.mean(np.absolute(po[:, n_pre:] - pt[:, n_pre:])))
    acc = float(np.mean(np.abs(np.diff(pt, n=2, axis=1) - np.diff(po, n=2, axis=1))))
    return l1, mae, acc


@dataclass
class StepCfg:
    """config/multimodal_context_v2.yml:29-36 + parse_args.py defaults."""
    n_pre_poses: int = 4
    loss_warmup: int = 0
    loss_gan_weight: float = 5.0
    loss_regression_weight: float = 500.0
    loss_kld_weight: float = 0.1
    loss_reg_weight: float = 0.05
    z_type: str = 'speaker'
    lr_gen: float = 5e-4
    lr_dis: float = 1e-4          # learning_rate * discriminator_lr_weight (0.2)
    betas: Tuple[float, float] = (0.5, 0.999)
    adam_eps: float = 1e-8


def is_param(key: str) -> bool:
    return not (key.endswith('running_mean') or key.endswith('running_var') or key.endswith('num_batches_tracked'))


def param_keys(sd: SD) -> List[str]:
    """Trainable entries, with the TCN's duplicated aliases (net.0 == conv1, net.4 == conv2) removed."""
    return [k for k in sd if is_param(k) and '.net.' not in k]


@dataclass
class AdamState:
    step: int = 0
    m: Dict[str, Tensor] = field(default_factory=dict)
    v: Dict[str, Tensor] = field(default_factory=dict)


def adam_update(sd: SD, grads: Dict[str, Tensor], st: AdamState, lr: float, betas=(0.5, 0.999), eps=1e-8) -> None:
    """torch.optim.Adam (no weight decay, no amsgrad): p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).
    Parameters whose grad is None are skipped, as torch does."""
    st.step += 1
    b1, b2 = betas
    bc1 = 1 - b1 ** st.step
    bc2 = 1 - b2 ** st.step
    for k, g in grads.items():
        if g is None:
            continue
        if k not in st.m:
            st.m[k] = torch.zeros_like(g)
            st.v[k] = torch.zeros_like(g)
        st.m[k].mul_(b1).add_(g, alpha=1 - b1)
        st.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (st.v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        sd[k].data.addcdiv_(st.m[k], denom, value=-lr / bc1)
        # keep the reference's aliased TCN keys in sync
        if '.conv1.' in k or '.conv2.' in k:
            alias = k.replace('.conv1.', '.net.0.').replace('.conv2.', '.net.4.')
            if alias in sd and alias != k:
                sd[alias] = sd[k]


def make_pre_seq(target: Tensor, n_pre: int) -> Tensor:
    """processor_v2.py:784-788."""
    pre = target.new_zeros(target.shape[0], target.shape[1], target.shape[2] + 1)
    pre[:, :n_pre, :-1] = target[:, :n_pre]
    pre[:, :n_pre, -1] = 1
    return pre


def dis_loss(dis_real: Tensor, dis_fake: Tensor) -> Tensor:
    """processor_v2.py:811 (non-saturating GAN)."""
    return torch.sum(-torch.mean(torch.log(dis_real + 1e-8) + torch.log(1 - dis_fake + 1e-8)))


def uses_divergence_term(scfg: StepCfg) -> bool:
    """processor_v2.py:899-900: the branch with the second generator pass (other speakers), the divergence regulariser and --
    for z_type 'speaker' only (:924-929) -- the KLD term.  Otherwise (:933-934) the loss is the regression term alone."""
    return scfg.z_type in ('speaker', 'random') and scfg.loss_reg_weight > 0.0


def gen_losses(scfg: StepCfg, out, target, dis_out, out_rand, z, z_rand, mu, log_var, use_gan: bool):
    """processor_v2.py:893-937.  Returns (total, dict of unweighted components; kld / div_reg are None where the reference
    leaves them None: out_rand is None <=> the branch of :933-934)."""
    huber = F.smooth_l1_loss(out / 0.1, target / 0.1) * 0.1
    gen_error = -torch.mean(torch.log(dis_out + 1e-8))
    kld = div_reg = None
    if uses_divergence_term(scfg):
        pose_l1 = (F.smooth_l1_loss(out / 0.05, out_rand.detach() / 0.05, reduction='none') * 0.05).sum(dim=(1, 2))
        z_l1 = (z.detach() - z_rand.detach()).abs().mean(1)
        div_reg = torch.clamp(-(pose_l1 / (z_l1 + 1.0e-5)), min=-1000).mean()
        if scfg.z_type == 'speaker':
            kld = -0.5 * torch.mean(1 + log_var - mu.pow(2) - log_var.exp())
            loss = scfg.loss_regression_weight * huber + scfg.loss_kld_weight * kld + scfg.loss_reg_weight * div_reg
        else:
            loss = scfg.loss_regression_weight * huber + scfg.loss_reg_weight * div_reg
    else:
        loss = scfg.loss_regression_weight * huber
    if use_gan:
        loss = loss + scfg.loss_gan_weight * gen_error
    return loss, dict(huber=huber, gen=gen_error, div_reg=div_reg, kld=kld)


def _leaf(sd: SD, keys: Sequence[str]) -> SD:
    """Detach-and-require-grad view of a model for autograd; buffers are shared (BN updates land in sd)."""
    out = dict(sd)
    for k in keys:
        out[k] = sd[k].detach().requires_grad_(True)
    for k in list(out):
        if '.net.0.' in k or '.net.4.' in k:
            out[k] = out[k.replace('.net.0.', '.conv1.').replace('.net.4.', '.conv2.')]
    return out


@dataclass
class StepNoise:
    """Noise for the seven forward passes of one step.  ``perm`` is the speaker permutation (:905)."""
    g_dis: Noise
    d_real: Noise
    d_fake: Noise
    pgt: Noise
    g_main: Noise
    d_gen: Noise
    g_rand: Noise
    perm: Optional[Tensor] = None

    @staticmethod
    def fresh() -> 'StepNoise':
        return StepNoise(*[Noise(None) for _ in range(7)])

    @staticmethod
    def off(batch: int) -> 'StepNoise':
        return StepNoise(*[Noise('off') for _ in range(7)], perm=torch.arange(batch).flip(0))


def gan_step(G: SD, D: SD, PGT: SD, g_opt: AdamState, d_opt: AdamState, mcfg: ModelCfg, scfg: StepCfg,
             in_text, in_audio, in_mfcc, target, vid, epoch: int, noise: StepNoise, train: bool = True,
             fast: bool = False, d_drop_noise_off: bool = False, ablation: str = 'none',
             signs: Optional[Dict[str, Dict[str, Tensor]]] = None):
    """Processor.forward_pass_s2ag (processor_v2.py:776-957), train branch, use_mfcc = True.

    G and D run in train mode (per_train_epoch :961-962); the frozen tri-modal baseline PGT is never
    put in eval mode by the reference either, so it also runs in train mode (BN batch stats, dropout).
    Returns (metric, loss_dict, grads) where metric is the 7-tuple's first element.
    """
    # ``ablation``: 'none' = PoseGenerator + AffDiscriminator (net/multimodal_context_net_v2.py); 'aff' = the
    # _abl_aff pairing (generator without the affective encoder + ConvDiscriminator); 'audio' = the _abl_audio generator
    # (raw waveform, use_mfcc False at processor_v2.py:794-797) + AffDiscriminator.
    if ablation == 'aff':
        pose_generator, aff_discriminator = pose_generator_abl_aff, conv_discriminator
    elif ablation == 'audio':
        pose_generator, aff_discriminator = pose_generator_abl_audio, globals()['aff_discriminator']
        in_mfcc = in_audio
    else:
        pose_generator, aff_discriminator = globals()['pose_generator'], globals()['aff_discriminator']
    # ``signs``: {pass name: {site: bool tensor}} -- branch decisions to replay per module pass (use_signs above; the parity
    # tests record them from the product's own step, tests/s2ag_testing.py StepSignTap); ``signs_used`` lists what was consumed
    sg = signs or {}
    used: Dict[str, List[str]] = {}
    audit: Dict[str, Dict[str, float]] = {}          # '<pass>/<site>' -> use_signs.audit entry (audit_benign checks it)

    class _P:                              # use_signs scope of one pass that also remembers which sites were consumed
        def __init__(self, name):
            self.name, self.ctx = name, use_signs(sg.get(name))

        def __enter__(self):
            return self.ctx.__enter__()

        def __exit__(self, *a):
            used[self.name] = list(self.ctx.used)
            audit.update({f'{self.name}/{k}': v for k, v in self.ctx.audit.items()})
            return self.ctx.__exit__(*a)
    gan_step.signs_used = used
    gan_step.signs_audit = audit
    pre_seq = make_pre_seq(target, scfg.n_pre_poses)
    use_gan = epoch > scfg.loss_warmup and scfg.loss_gan_weight > 0.0
    losses: Dict[str, float] = {}
    grads_out: Dict[str, Dict[str, Tensor]] = {}
    gk, dk = param_keys(G), param_keys(D)

    if use_gan:
        Gl, Dl = _leaf(G, gk), _leaf(D, dk)
        with torch.no_grad(), _P('g_dis'):
            fake, *_ = pose_generator(Gl, mcfg, pre_seq, in_text, in_mfcc, vid, train, noise.g_dis, fast)
        with _P('d_real'):
            d_real = aff_discriminator(Dl, target, train, noise.d_real, fast)
        with _P('d_fake'):
            d_fake = aff_discriminator(Dl, fake.detach(), train, noise.d_fake, fast)
        d_err = dis_loss(d_real, d_fake)
        losses['dis'] = float(d_err.detach())
        if train:
            gd = torch.autograd.grad(d_err, [Dl[k] for k in dk], allow_unused=True)
            grads_out['D'] = dict(zip(dk, gd))
            adam_update(D, grads_out['D'], d_opt, scfg.lr_dis, scfg.betas, scfg.adam_eps)

    Gl, Dl = _leaf(G, gk), _leaf(D, dk)
    with torch.no_grad(), _P('pgt'):
        out_tri, *_ = pose_generator_trimodal(PGT, mcfg, pre_seq, in_text, in_audio, vid, True, noise.pgt, fast)
    with _P('g_main'):
        out, z, mu, log_var = pose_generator(Gl, mcfg, pre_seq, in_text, in_mfcc, vid, train, noise.g_main, fast)
    with _P('d_gen'):
        d_out = aff_discriminator(Dl, out, train, noise.d_gen, fast)
    out_rand = z_rand = None
    if uses_divergence_term(scfg):         # processor_v2.py:899-910 (z_type 'random': rand_vids = None -- the speaker-embedding
        # generator then asserts, net/multimodal_context_net_v2.py:511, exactly as the reference's own step does)
        if scfg.z_type == 'speaker':
            perm = noise.perm if noise.perm is not None else torch.randperm(vid.shape[0])
            rand_vids = vid[perm]
        else:
            rand_vids = None
        with torch.no_grad(), _P('g_rand'):
            out_rand, z_rand, _, _ = pose_generator(Gl, mcfg, pre_seq, in_text, in_mfcc, rand_vids, train,
                                                    noise.g_rand, fast)
    loss, comp = gen_losses(scfg, out, target, d_out, out_rand, z, z_rand, mu, log_var,
                            epoch > scfg.loss_warmup)
    losses.update(loss=scfg.loss_regression_weight * float(comp['huber'].detach()))
    if comp['kld'] is not None:
        losses['KLD'] = scfg.loss_kld_weight * float(comp['kld'].detach())
    if comp['div_reg'] is not None:
        losses['DIV_REG'] = scfg.loss_reg_weight * float(comp['div_reg'].detach())
    if use_gan:
        losses['gen'] = scfg.loss_gan_weight * float(comp['gen'].detach())
    losses['total'] = float(loss.detach())
    if train:
        gg = torch.autograd.grad(loss, [Gl[k] for k in gk], allow_unused=True)
        grads_out['G'] = dict(zip(gk, gg))
        adam_update(G, grads_out['G'], g_opt, scfg.lr_gen, scfg.betas, scfg.adam_eps)
    metric = float(F.l1_loss(out.detach(), target)) - float(F.l1_loss(out_tri, target))
    return metric, losses, grads_out


# ----------------------------------------------------------------------------------------------
# model construction helpers (test/bench side): shapes of every state_dict entry
# ----------------------------------------------------------------------------------------------
def _bn_entries(p: str, c: int) -> Dict[str, Tuple[int, ...]]:
    return {p + 'weight': (c,), p + 'bias': (c,), p + 'running_mean': (c,), p + 'running_var': (c,),
            p + 'num_batches_tracked': ()}


def _aff_encoder_shapes(p: str) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    for name, cin, ks in (('st_gcn1.', 3, 5), ('st_gcn2.', 48, 3)):
        q = p + name
        s[q + 'gcn.conv.weight'] = (80, cin, 9, 1)
        s[q + 'gcn.conv.bias'] = (80,)
        s.update(_bn_entries(q + 'tcn.0.', 16))
        s[q + 'tcn.2.weight'] = (16, 16, 9, ks)
        s[q + 'tcn.2.bias'] = (16,)
        s.update(_bn_entries(q + 'tcn.3.', 16))
        s[q + 'residual.0.weight'] = (16, cin, 1, 1)
        s[q + 'residual.0.bias'] = (16,)
        s.update(_bn_entries(q + 'residual.1.', 16))
        if name == 'st_gcn1.':
            s.update(_bn_entries(p + 'batch_norm1.', 144))
    s.update(_bn_entries(p + 'batch_norm2.', 48))
    s[p + 'conv3.weight'] = (16, 48, 5)
    s[p + 'conv3.bias'] = (16,)
    s.update(_bn_entries(p + 'batch_norm3.', 16))
    s[p + 'conv4.weight'] = (8, 16, 3)
    s[p + 'conv4.bias'] = (8,)
    s.update(_bn_entries(p + 'batch_norm4.', 8))
    return s


def _gru_shapes(p: str, in_size: int, H: int, layers: int) -> Dict[str, Tuple[int, ...]]:
    s = {}
    for l in range(layers):
        for suf in ('', '_reverse'):
            s[f'{p}weight_ih_l{l}{suf}'] = (3 * H, in_size if l == 0 else 2 * H)
            s[f'{p}weight_hh_l{l}{suf}'] = (3 * H, H)
            s[f'{p}bias_ih_l{l}{suf}'] = (3 * H,)
            s[f'{p}bias_hh_l{l}{suf}'] = (3 * H,)
    return s


def _text_encoder_shapes(p: str, n_words: int, embed: int, hidden: int, layers: int):
    s = {p + 'embedding.weight': (n_words, embed)}
    for i in range(layers):
        cin = embed if i == 0 else hidden
        for tag, alias, ci in (('conv1', 'net.0', cin), ('conv2', 'net.4', hidden)):
            for t in (tag, alias):
                q = f'{p}tcn.network.{i}.{t}.'
                s[q + 'bias'] = (hidden,)
                s[q + 'weight_g'] = (hidden, 1, 1)
                s[q + 'weight_v'] = (hidden, ci, 2)
        if cin != hidden:
            s[f'{p}tcn.network.{i}.downsample.weight'] = (hidden, cin, 1)
            s[f'{p}tcn.network.{i}.downsample.bias'] = (hidden,)
    s[p + 'decoder.weight'] = (32, hidden)
    s[p + 'decoder.bias'] = (32,)
    return s


def _wav_encoder_shapes(p: str):
    s = {}
    fe = p + 'feat_extractor.'
    for idx, (co, ci) in zip((0, 3, 6, 9), ((16, 1), (32, 16), (64, 32), (32, 64))):
        s[f'{fe}{idx}.weight'] = (co, ci, 15)
        s[f'{fe}{idx}.bias'] = (co,)
    for idx, c in zip((1, 4, 7), (16, 32, 64)):
        s.update(_bn_entries(f'{fe}{idx}.', c))
    return s


def _speaker_shapes(n_spk: int):
    return {'speaker_embedding.0.weight': (n_spk, 16), 'speaker_embedding.1.weight': (16, 16),
            'speaker_embedding.1.bias': (16,), 'speaker_mu.weight': (16, 16), 'speaker_mu.bias': (16,),
            'speaker_log_var.weight': (16, 16), 'speaker_log_var.bias': (16,)}


def generator_shapes(cfg: ModelCfg, n_words: int, n_spk: int, mfcc_length: int = 71, num_mfcc: int = 37,
                     pose_dim: int = 27, embed: int = 300, audio: str = 'mfcc', aff: bool = True):
    """state_dict layout of PoseGenerator ('mfcc') / _abl_audio.PoseGenerator ('wav') / _abl_aff.PoseGenerator
    (``aff=False``: no aff_encoder entries, the GRU reads the raw pose_dim + 1 columns)."""
    s: Dict[str, Tuple[int, ...]] = {}
    if audio == 'mfcc':
        chans = [mfcc_length, 64, 64, 48, cfg.n_poses]
        for i, k in zip(range(1, 5), (5, 5, 3, 3)):
            s[f'audio_encoder.conv{i}.weight'] = (chans[i], chans[i - 1], k)
            s[f'audio_encoder.conv{i}.bias'] = (chans[i],)
            s.update(_bn_entries(f'audio_encoder.batch_norm{i}.', chans[i]))
        s['audio_encoder.linear1.weight'] = (32, num_mfcc)
        s['audio_encoder.linear1.bias'] = (32,)
    else:
        s.update(_wav_encoder_shapes('audio_encoder.'))
    s.update(_text_encoder_shapes('text_encoder.', n_words, embed, cfg.hidden_size, cfg.n_layers))
    if aff:
        s.update(_aff_encoder_shapes('aff_encoder.'))
    s.update(_speaker_shapes(n_spk))
    H = cfg.hidden_size_s2eg
    s.update(_gru_shapes('gru.', (8 if aff else pose_dim + 1) + 32 + 32 + 16, H, cfg.n_layers))
    s['out.0.weight'] = (H // 2, H)
    s['out.0.bias'] = (H // 2,)
    s['out.2.weight'] = (pose_dim, H // 2)
    s['out.2.bias'] = (pose_dim,)
    return s


def trimodal_shapes(cfg: ModelCfg, n_words: int, n_spk: int, pose_dim: int = 27, embed: int = 300):
    s: Dict[str, Tuple[int, ...]] = {}
    s.update(_wav_encoder_shapes('audio_encoder.'))
    s.update(_text_encoder_shapes('text_encoder.', n_words, embed, cfg.hidden_size, cfg.n_layers))
    s.update(_speaker_shapes(n_spk))
    H = cfg.hidden_size
    s.update(_gru_shapes('gru.', pose_dim + 1 + 32 + 32 + 16, H, cfg.n_layers))
    s['out.0.weight'] = (H // 2, H)
    s['out.0.bias'] = (H // 2,)
    s['out.2.weight'] = (pose_dim, H // 2)
    s['out.2.bias'] = (pose_dim,)
    return s


def aff_discriminator_shapes(n_poses: int = 34):
    s = _aff_encoder_shapes('aff_encoder.')
    s.update(_gru_shapes('gru.', 8, 64, 4))
    s.update({'out.weight': (1, 64), 'out.bias': (1,), 'out2.weight': (1, n_poses), 'out2.bias': (1,)})
    return s


def conv_discriminator_shapes(pose_dim: int = 27, n_poses: int = 34):
    s = {'pre_conv.0.weight': (16, pose_dim, 3), 'pre_conv.0.bias': (16,),
         'pre_conv.3.weight': (8, 16, 3), 'pre_conv.3.bias': (8,),
         'pre_conv.6.weight': (8, 8, 3), 'pre_conv.6.bias': (8,)}
    s.update(_bn_entries('pre_conv.1.', 16))
    s.update(_bn_entries('pre_conv.4.', 8))
    s.update(_gru_shapes('gru.', 8, 64, 4))
    s.update({'out.weight': (1, 64), 'out.bias': (1,), 'out2.weight': (1, n_poses - 6), 'out2.bias': (1,)})
    return s


def recipe_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int, scale: float = 1.0, tcn_aliases: bool = True) -> SD:
    """Frozen weight recipe shared by the golden generator and the tests: one legacy
    ``np.random.RandomState(seed)`` stream (version-stable), consumed in sorted-key order.
    Fan-in scaled uniform weights; BN gamma in [0.5,1.5], running_var in [0.5,1.5];
    the TCN's aliased keys (net.0/net.4) share the tensor of conv1/conv2."""
    rs = np.random.RandomState(seed)
    sd: SD = {}
    for k in sorted(shapes):
        shp = shapes[k]
        if tcn_aliases and ('.net.0.' in k or '.net.4.' in k):
            continue
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.zeros((), dtype=torch.int64)
            continue
        n = int(np.prod(shp)) if len(shp) else 1
        u = rs.uniform(-1.0, 1.0, size=n).astype(np.float32).reshape(shp)
        if k.endswith('running_var') or (k.endswith('.weight') and len(shp) == 1) or k.endswith('weight_g'):
            t = 1.0 + 0.5 * u
        elif k.endswith('running_mean') or k.endswith('bias'):
            t = 0.1 * u
        elif k.endswith('embedding.weight') or k.endswith('embedding.0.weight'):
            t = u
        else:
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else shp[0]
            t = u * (scale / math.sqrt(fan_in))
        sd[k] = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
    for k in shapes:
        if tcn_aliases and ('.net.0.' in k or '.net.4.' in k):
            sd[k] = sd[k.replace('.net.0.', '.conv1.').replace('.net.4.', '.conv2.')]
    return sd


def recipe_inputs(B: int, T: int, seed: int, n_words: int, n_spk: int, audio_len: int = 36267,
                  mfcc_len: int = 71, num_mfcc: int = 37) -> Dict[str, Tensor]:
    """Synthetic TED-shaped batch of SURVEY.md section 8(d), from a legacy RandomState stream."""
    rs = np.random.RandomState(seed)
    text = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        k = rs.randint(2, 9)
        pos = rs.choice(T, size=k, replace=False)
        text[b, pos] = rs.randint(4, n_words, size=k)
    audio = np.clip(rs.standard_normal((B, audio_len)) * 0.05, -1, 1).astype(np.float32)
    mfcc = (rs.standard_normal((B, num_mfcc, mfcc_len)) * 0.1).astype(np.float32)
    target = (rs.standard_normal((B, T, 27)) * 0.2).astype(np.float32)
    vid = rs.randint(0, n_spk, size=B).astype(np.int64)
    return dict(in_text=torch.from_numpy(text), in_audio=torch.from_numpy(audio), in_mfcc=torch.from_numpy(mfcc),
                target=torch.from_numpy(target), vid=torch.from_numpy(vid))
