"""Generator / discriminator networks of the S2AG GAN step on MI355X.

Drop-in for ``net.multimodal_context_net_v2`` of the reference (file:line cited per class): same class
names, constructor and forward signatures, return tuples and ``state_dict`` keys/shapes (SURVEY.md
Appendix C), so reference checkpoints load with ``strict=True``.  The torch ``nn`` modules held here are
parameter/buffer containers only -- every forward goes through the HIP kernels in ``ops`` on
channels-last tensors; nothing calls ``nn.Conv1d.forward`` & co and there is no CPU path.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .._lib import ACT_LEAKY, ACT_NONE, ACT_SIGMOID
from ..noise import new_site, noise_pass
from .tcn import TemporalConvNet
from .utils.graph import Graph
from .utils.tgcn import STGraphConv, default_cols

# skeleton constants (utils/ted_db_utils.py:14-19 of the reference)
dir_vec_pairs = [(0, 1, 0.26), (1, 2, 0.18), (2, 3, 0.14), (1, 4, 0.22), (4, 5, 0.36),
                 (5, 6, 0.33), (1, 7, 0.22), (7, 8, 0.36), (8, 9, 0.33)]
dir_edge_pairs = [(0, 1), (1, 2), (0, 3), (3, 4), (4, 5), (0, 6), (6, 7), (7, 8)]
body_parts_edge_idx = [np.arange(0, 3), np.arange(3, 6), np.arange(6, 9)]
max_body_part_edges = 3
body_parts_edge_pairs = [(0, 1), (0, 2)]


class GRU(nn.Module):
    """Parameter container + HIP forward of ``nn.GRU(batch_first=True, bidirectional=True)``; parameter names
    and init (U(-1/sqrt(H), 1/sqrt(H))) as torch's, so ``gru.weight_ih_l0`` ... keys match the reference."""

    def __init__(self, input_size, hidden_size, num_layers=1, batch_first=True, bidirectional=True, dropout=0.0):
        super().__init__()
        if not (batch_first and bidirectional):
            raise NotImplementedError('the S2AG nets only use batch_first bidirectional GRUs')
        self.input_size, self.hidden_size, self.num_layers, self.dropout = input_size, hidden_size, num_layers, dropout
        self.batch_first, self.bidirectional = True, True
        self._packed_checked = False
        k = 1.0 / math.sqrt(hidden_size)
        # registration order = arena order (optim.ParamArena follows .parameters()): the two directions of every
        # tensor are adjacent, so ops.gru runs both directions' projections / weight gradients as single GEMMs
        for l in range(num_layers):
            in_l = input_size if l == 0 else 2 * hidden_size
            for kind, shape in (('weight_ih', (3 * hidden_size, in_l)), ('weight_hh', (3 * hidden_size, hidden_size)),
                                ('bias_ih', (3 * hidden_size,)), ('bias_hh', (3 * hidden_size,))):
                for suf in ('', '_reverse'):
                    self.register_parameter(f'{kind}_l{l}{suf}', nn.Parameter(torch.empty(shape).uniform_(-k, k)))
        self.site0 = new_site(num_layers)

    def _pack_frozen(self):
        """A GRU outside any arena (the frozen tri-modal baseline): move its tensors into one buffer, pairs adjacent,
        the first time it runs on the GPU.  Trainable (arena-managed) tensors are never touched."""
        ps = list(self.parameters())
        if any(p.requires_grad or p.grad is not None for p in ps) or not ps[0].is_cuda:
            return
        if all(ps[i].data_ptr() + 4 * ps[i].numel() == ps[i + 1].data_ptr() for i in range(0, len(ps), 2)):
            return
        flat = torch.empty(sum(p.numel() for p in ps), dtype=torch.float32, device=ps[0].device)
        off = 0
        for p in ps:
            v = flat[off:off + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            off += p.numel()

    def flatten_parameters(self):
        pass

    def flat_weights(self):
        out = []
        for l in range(self.num_layers):
            for suf in ('', '_reverse'):
                out += [getattr(self, f'weight_ih_l{l}{suf}'), getattr(self, f'weight_hh_l{l}{suf}'),
                        getattr(self, f'bias_ih_l{l}{suf}'), getattr(self, f'bias_hh_l{l}{suf}')]
        return out

    def run(self, x, noise, sum_dirs, mates=()):
        """``mates`` = [(x_i, noise_i), ...]: no-grad passes run in lockstep with this one (ops._GRU.forward); the
        result is then (out, *mate_outs)."""
        if not self._packed_checked:
            self._pack_frozen()
            self._packed_checked = True
        return ops.gru(x, self.flat_weights(), self.hidden_size, self.num_layers, self.training, self.dropout, noise,
                       self.site0, sum_dirs, mates)

    def forward(self, x, hx=None):
        if hx is not None:
            raise NotImplementedError('the hot path always starts from h0 = 0')
        with noise_pass(x.device) as nz:
            return self.run(x, nz, False), None


class WavEncoder(nn.Module):
    """net/multimodal_context_net_v2.py:14-33.  (B, n_samples) raw waveform -> (B, frames, 32)."""

    def __init__(self):
        super().__init__()
        self.feat_extractor = nn.Sequential(
            nn.Conv1d(1, 16, 15, stride=5, padding=1600),
            nn.BatchNorm1d(16),
            nn.LeakyReLU(0.3, inplace=True),
            nn.Conv1d(16, 32, 15, stride=6),
            nn.BatchNorm1d(32),
            nn.LeakyReLU(0.3, inplace=True),
            nn.Conv1d(32, 64, 15, stride=6),
            nn.BatchNorm1d(64),
            nn.LeakyReLU(0.3, inplace=True),
            nn.Conv1d(64, 32, 15, stride=6),
        )

    def _forward_bf16(self, wav_data):
        """bf16 mode (bf16.py): conv1 reads the fp32 waveform and writes bf16; conv2-4 are implicit GEMMs on the bf16
        matrix pipe over contiguous 15-tap windows; BatchNorm statistics ride in the conv epilogues; (B, 34, 32) leaves
        as fp32."""
        from .. import bf16
        fe = self.feat_extractor
        if self.__dict__.get('_pack16') is None:
            self.__dict__['_pack16'] = bf16.WeightPack()
            for i, (ci, co) in zip((3, 6, 9), ((16, 32), (32, 64), (64, 32))):
                self.__dict__['_pack16'].add(f'c{i}', (lambda c=fe[i]: c.weight), 'reference', co, ci, 15, stride=6)
        pk = self.__dict__['_pack16']
        if all(fe[i].training for i in (1, 4, 7)) and bf16.wave_fused_supported(fe):
            return bf16.wave_encoder_fused(wav_data, fe, pk)       # BatchNorm folded into the convs (wave_fused.hip)
        x = bf16.conv_c1(wav_data, fe[0].weight, fe[0].bias, 5, 1600, bn_stats=fe[1].training)
        x = bf16.batch_norm_act(x, fe[1], slope=0.3)
        x = bf16.conv(x, fe[3].weight, fe[3].bias, pk, 'c3', 16, 32, 15, stride=6, bn_stats=fe[4].training)
        x = bf16.batch_norm_act(x, fe[4], slope=0.3)
        x = bf16.conv(x, fe[6].weight, fe[6].bias, pk, 'c6', 32, 64, 15, stride=6, bn_stats=fe[7].training)
        x = bf16.batch_norm_act(x, fe[7], slope=0.3)
        return bf16.conv(x, fe[9].weight, fe[9].bias, pk, 'c9', 64, 32, 15, stride=6, out_f32=True)

    def forward(self, wav_data):
        from .. import bf16
        if bf16.enabled() and self.training:
            return self._forward_bf16(wav_data)
        from .. import wave12
        fe = self.feat_extractor
        if (fe[1].training and fe[4].training and wav_data.is_cuda and wav_data.dtype == torch.float32 and wav_data.dim() == 2
                and wave12.supported(fe)):
            # conv1 -> BatchNorm -> LeakyReLU -> conv2 without conv1's (B, 7891, 16) output in HBM (csrc/wave12.hip)
            x = wave12.head_f32(wav_data, fe)
        else:
            x = wav_data.unsqueeze(2)                                 # (B, L, 1) channels-last
            x = ops.conv1d_nlc(x, fe[0].weight, fe[0].bias, stride=5, pad=1600, bn_stats=fe[1].training)
            x = ops.batch_norm_act(x, fe[1], slope=0.3)
            x = ops.conv1d_nlc(x, fe[3].weight, fe[3].bias, stride=6, tm_copy=True, bn_stats=fe[4].training)
        x = ops.batch_norm_act(x, fe[4], slope=0.3)
        x = ops.conv1d_nlc(x, fe[6].weight, fe[6].bias, stride=6, tm_copy=True, bn_stats=fe[7].training)
        x = ops.batch_norm_act(x, fe[7], slope=0.3)
        return ops.conv1d_nlc(x, fe[9].weight, fe[9].bias, stride=6, tm_copy=True)  # already (batch x seq x dim)


class MFCCEncoder(nn.Module):
    """:36-58.  (B, num_mfcc, mfcc_length) -> (B, time_steps, 32).  MFCC frames are the conv channels, so the
    input is already channels-last for us: no permute."""

    def __init__(self, mfcc_length, num_mfcc, time_steps):
        super().__init__()
        self.conv1 = nn.Conv1d(mfcc_length, 64, 5, padding=2)
        self.batch_norm1 = nn.BatchNorm1d(64)
        self.conv2 = nn.Conv1d(64, 64, 5, padding=2)
        self.batch_norm2 = nn.BatchNorm1d(64)
        self.conv3 = nn.Conv1d(64, 48, 3, padding=1)
        self.batch_norm3 = nn.BatchNorm1d(48)
        self.conv4 = nn.Conv1d(48, time_steps, 3, padding=1)
        self.batch_norm4 = nn.BatchNorm1d(time_steps)
        self.linear1 = nn.Linear(num_mfcc, 32)
        self.activation = nn.LeakyReLU(0.3, inplace=True)

    def forward(self, mfcc_data):
        x = mfcc_data
        for conv, bn, pad in ((self.conv1, self.batch_norm1, 2), (self.conv2, self.batch_norm2, 2),
                              (self.conv3, self.batch_norm3, 1), (self.conv4, self.batch_norm4, 1)):
            x = ops.batch_norm_act(ops.conv1d_nlc(x, conv.weight, conv.bias, pad=pad, bn_stats=bn.training, tm_copy=True), bn,
                                   slope=0.3)
        x = x.transpose(1, 2).contiguous()                            # (B, time_steps, num_mfcc): layout glue only
        return ops.linear(x, self.linear1.weight, self.linear1.bias, act=ACT_LEAKY, slope=0.3)


class TextEncoderTCN(nn.Module):
    """:61-91.  (B, T) int64 -> ((B, T, 32), 0)."""

    def __init__(self, args, n_words, embed_size=300, pre_trained_embedding=None, kernel_size=2, dropout=0.3,
                 emb_dropout=0.1):
        super().__init__()
        if pre_trained_embedding is not None:
            assert pre_trained_embedding.shape[0] == n_words
            assert pre_trained_embedding.shape[1] == embed_size
            self.embedding = nn.Embedding.from_pretrained(torch.FloatTensor(pre_trained_embedding),
                                                          freeze=args.freeze_wordembed)
        else:
            self.embedding = nn.Embedding(n_words, embed_size)
        num_channels = [args.hidden_size] * args.n_layers
        self.tcn = TemporalConvNet(embed_size, num_channels, kernel_size, dropout=dropout)
        self.decoder = nn.Linear(num_channels[-1], 32)
        self.drop = nn.Dropout(emb_dropout)
        self.emb_dropout = emb_dropout
        self.init_weights()
        self.site = new_site()

    def init_weights(self):
        self.decoder.bias.data.fill_(0)
        self.decoder.weight.data.normal_(0, 0.01)

    def forward(self, in_data):
        from .. import bf16
        with noise_pass(in_data.device) as nz:
            p = self.drop.p if self.training else 0.0
            if bf16.enabled() and self.training and self.tcn.bf16_capable():
                # bf16 mode: (B, T, 320) bf16 rows with zero pad channels from the embedding gather to the decoder
                emb = bf16.embedding(in_data, self.embedding.weight, p, nz, self.site)
                y = self.tcn.forward_nlc_bf16(emb, nz, decoder=self.decoder)
                return y.contiguous(), 0
            emb = ops.embedding(in_data, self.embedding.weight, p, nz, self.site)      # (B, T, E) channels-last
            y = self.tcn.forward_nlc(emb, nz)
            y = ops.linear(y, self.decoder.weight, self.decoder.bias)
        return y.contiguous(), 0

    def lockstep_capable(self, in_data):
        """forward_passes covers the default fp32 training path with the clip-resident TemporalConvNet."""
        from .. import bf16
        return bool(self.training and not bf16.enabled() and in_data.dim() == 2
                    and self.tcn.lockstep_capable(in_data.shape[1], self.embedding.weight.shape[1]))

    def forward_passes(self, in_data, noises, grad=True):
        """Several passes over the same token ids in LOCKSTEP (the trainer's generator passes of a step differ in noise
        only): ``noises[k]`` is pass k's snapshot; only pass 0 carries autograd.  The passes' embedding rows go into one
        (nP * B, T, E) batch and the TemporalConvNet runs once over it -- one workgroup per clip leaves half of the chip
        idle at B = 128 -- then the decoder once for pass 0 and once for the rest.  Returns [(B, T, 32)] * nP, bit-identical to
        the passes run one after the other."""
        nP, (B, T) = len(noises), in_data.shape
        E = self.embedding.weight.shape[1]
        p = self.drop.p if self.training else 0.0
        big = torch.empty(nP * B, T, E, dtype=torch.float32, device=in_data.device)
        outer = torch.is_grad_enabled()
        with torch.set_grad_enabled(bool(grad) and outer):
            emb0 = ops.embedding(in_data, self.embedding.weight, p, noises[0], self.site, out=big[:B])
        with torch.no_grad():
            for k in range(1, nP):
                ops.embedding(in_data, self.embedding.weight, p, noises[k], self.site, out=big[k * B:(k + 1) * B])
        with torch.set_grad_enabled(bool(grad) and outer):
            y0, ym = self.tcn.forward_nlc(emb0, noises[0], batch=big, noises=noises)
            out0 = ops.linear(y0, self.decoder.weight, self.decoder.bias)
        with torch.no_grad():
            outm = ops.linear(ym, self.decoder.weight, self.decoder.bias)
        return [out0.contiguous()] + [outm[(k - 1) * B:k * B] for k in range(1, nP)]


class AffEncoder(nn.Module):
    """:94-175.  (B, T, 27) -> (B, T, 8).  A1/A2 are non-persistent buffers (the reference keeps them as plain
    ``.cuda()`` attributes outside the state_dict)."""

    def __init__(self, coords=3):
        super().__init__()
        self.coords = coords
        self.num_dir_vec_pairs = len(dir_vec_pairs)
        graph1 = Graph(self.num_dir_vec_pairs, dir_edge_pairs, strategy='spatial', max_hop=2)
        self.register_buffer('A1', torch.tensor(graph1.A, dtype=torch.float32), persistent=False)
        self.num_body_parts = len(body_parts_edge_idx)
        graph2 = Graph(self.num_body_parts, body_parts_edge_pairs, strategy='spatial', max_hop=2)
        self.register_buffer('A2', torch.tensor(graph2.A, dtype=torch.float32), persistent=False)

        self.st_gcn1 = STGraphConv(coords, 16, self.A1.size(0), (9, 5), stride=(1, 1), padding=(4, 2))
        self.batch_norm1 = nn.BatchNorm1d(16 * self.num_dir_vec_pairs)
        self.st_gcn2 = STGraphConv(48, 16, self.A2.size(0), (9, 3), stride=(1, 1), padding=(4, 1))
        self.batch_norm2 = nn.BatchNorm1d(16 * self.num_body_parts)
        self.conv3 = nn.Conv1d(48, 16, 5, padding=2)
        self.batch_norm3 = nn.BatchNorm1d(16)
        self.conv4 = nn.Conv1d(16, 8, 3, padding=1)
        self.batch_norm4 = nn.BatchNorm1d(8)
        self.activation = nn.LeakyReLU(inplace=True)

        # Column orders of the channels-last intermediates.  Block 1 writes column p*48 + c*3 + j for edge
        # w = 3p + j and channel c, which IS the (body part p, channel c*3 + j) layout block 2 consumes
        # (:161-167) -- the regrouping costs nothing.  Block 2 writes c*3 + p, the channel order of
        # batch_norm2 / conv3 (:169-172).
        V1, V2, F = self.num_dir_vec_pairs, self.num_body_parts, 16
        self.in1 = default_cols(V1, coords)
        out1 = np.empty((V1, F), dtype=np.int64)
        bn1 = np.empty(V1 * F, dtype=np.int32)
        for p, idx in enumerate(body_parts_edge_idx):
            for j, w in enumerate(idx):
                for c in range(F):
                    out1[w, c] = p * (max_body_part_edges * F) + c * len(idx) + j
                    bn1[out1[w, c]] = c * V1 + w                      # BatchNorm1d(144) channel = c*9 + w (:159-160)
        self.out1 = out1
        self.in2 = default_cols(V2, max_body_part_edges * F)           # (part p, channel) -> p*48 + ch
        self.out2 = (np.arange(F)[None, :] * V2 + np.arange(V2)[:, None]).astype(np.int64)   # (p, c) -> c*3 + p
        self.register_buffer('bn1_map', torch.from_numpy(bn1), persistent=False)

    def forward(self, poses):
        n, t, jc = poses.shape
        f1 = self.st_gcn1.forward_nlc(poses, self.A1, self.in1, self.out1)              # (n, t, 144)
        f1 = ops.batch_norm_act(f1, self.batch_norm1, slope=1.0, chan_map=self.bn1_map)
        f2 = self.st_gcn2.forward_nlc(f1, self.A2, self.in2, self.out2)                 # (n, t, 48)
        f2 = ops.batch_norm_act(f2, self.batch_norm2, slope=1.0)
        x3 = ops.conv1d_nlc(f2, self.conv3.weight, self.conv3.bias, pad=self.conv3.padding[0])
        x3 = ops.batch_norm_act(x3, self.batch_norm3, slope=0.01)
        x4 = ops.conv1d_nlc(x3, self.conv4.weight, self.conv4.bias, pad=self.conv4.padding[0])
        return ops.batch_norm_act(x4, self.batch_norm4, slope=0.01)                      # (n, t, 8)


class _SpeakerZ:
    """speaker embedding -> (mu, log_var) -> re-parametrised z; shared by the generators (:271-276,:470-476)."""

    def _build_speaker(self, z_obj):
        self.speaker_embedding = None
        if z_obj:
            self.z_size = 16
            self.in_size += self.z_size
            if z_obj.__class__.__name__ == 'Vocab':
                self.speaker_embedding = nn.Sequential(nn.Embedding(z_obj.n_words, self.z_size),
                                                       nn.Linear(self.z_size, self.z_size))
                self.speaker_mu = nn.Linear(self.z_size, self.z_size)
                self.speaker_log_var = nn.Linear(self.z_size, self.z_size)
        self.z_site = new_site()

    def _z(self, in_text, vid_indices, nz):
        if not self.z_obj:
            return None, None, None
        if self.speaker_embedding:
            assert vid_indices is not None
            e = ops.embedding(vid_indices, self.speaker_embedding[0].weight)
            e = ops.linear(e, self.speaker_embedding[1].weight, self.speaker_embedding[1].bias)
            z_mu = ops.linear(e, self.speaker_mu.weight, self.speaker_mu.bias)
            z_log_var = ops.linear(e, self.speaker_log_var.weight, self.speaker_log_var.bias)
            return ops.reparametrize(z_mu, z_log_var, nz, self.z_site), z_mu, z_log_var
        z = ops.normal_noise(nz, self.z_site, (in_text.shape[0], self.z_size))
        return z, None, None

    # Data-parallel trainers cut the autograd graph at the decoder's input: everything behind the cut (GRU, out) is
    # back-propagated first and its gradient bucket goes on the wire while the encoders' backward still runs
    # (parallel.GradExchange).  ``cut_backward = True`` makes a grad-enabled forward leave ``(full, leaf)`` in ``_cut``:
    # ``loss.backward()`` then stops at ``leaf``; ``torch.autograd.backward(full, leaf.grad)`` runs the rest.  The
    # speaker statistics (z_mu, z_log_var feed the KLD term directly) are cut the same way, so that the first half
    # never enters the encoder side of the graph: ``_cut`` = ([full tensors], [leaves]).
    cut_backward = False
    _cut = None

    def _cut_here(self, *tensors):
        """Detached leaves of ``tensors`` (recorded in ``_cut``) when the backward cut is armed, else ``tensors``."""
        if not (self.cut_backward and torch.is_grad_enabled()):
            return tensors
        if self._cut is None:
            self._cut = ([], [])
        out = []
        for t in tensors:
            if t is None or not t.requires_grad:
                out.append(t)
                continue
            leaf = t.detach().requires_grad_(True)
            self._cut[0].append(t)
            self._cut[1].append(leaf)
            out.append(leaf)
        return tuple(out)

    def _decode(self, in_data, nz, out_slope):
        (in_data,) = self._cut_here(in_data)
        h = self.gru.run(in_data, nz, sum_dirs=True)                                     # (B, T, H), halves summed
        h = ops.linear(h, self.out[0].weight, self.out[0].bias, act=ACT_LEAKY, slope=out_slope)
        return ops.linear(h, self.out[2].weight, self.out[2].bias)

    def _context(self, pre, audio, text, z_context=None):
        """the decoder's input: [pre | audio | text | z broadcast over the frames] (one launch)"""
        if self.input_context == 'both':
            parts = (pre, audio, text)
        elif self.input_context == 'audio':
            parts = (pre, audio)
        elif self.input_context == 'text':
            parts = (pre, text)
        elif self.input_context == 'none':
            parts = (pre,)
        else:
            assert False
        if len(parts) == 1 and z_context is None:
            return pre
        return ops.context_cat(parts, z_context)


class PoseGeneratorTriModal(nn.Module, _SpeakerZ):
    """:247-343.  The frozen tri-modal baseline; ``nn.LeakyReLU(True)`` in ``out`` is the identity (slope 1.0)."""

    def __init__(self, args, pose_dim, n_words, word_embed_size, word_embeddings, z_obj=None):
        super().__init__()
        self.pre_length = args.n_pre_poses
        self.gen_length = args.n_poses - args.n_pre_poses
        self.z_obj = z_obj
        self.input_context = args.input_context
        if self.input_context == 'both':
            self.in_size = 32 + 32 + pose_dim + 1
        elif self.input_context == 'none':
            self.in_size = pose_dim + 1
        else:
            self.in_size = 32 + pose_dim + 1
        self.audio_encoder = WavEncoder()
        self.text_encoder = TextEncoderTCN(args, n_words, word_embed_size, pre_trained_embedding=word_embeddings,
                                           dropout=args.dropout_prob)
        self._build_speaker(z_obj)
        self.hidden_size = args.hidden_size
        self.gru = GRU(self.in_size, hidden_size=self.hidden_size, num_layers=args.n_layers, batch_first=True,
                       bidirectional=True, dropout=args.dropout_prob)
        self.out = nn.Sequential(nn.Linear(self.hidden_size, self.hidden_size // 2), nn.LeakyReLU(True),
                                 nn.Linear(self.hidden_size // 2, pose_dim))
        self.do_flatten_parameters = False
        self._branches = ops.BranchStreams(2)

    def forward(self, pre_seq, in_text, in_audio, vid_indices=None):
        with noise_pass(pre_seq.device) as nz:
            audio = text = None
            fns = [lambda: self._z(in_text, vid_indices, nz)]
            if self.input_context != 'none':
                fns = [lambda: self.audio_encoder(in_audio), lambda: self.text_encoder(in_text)[0]] + fns
            res = self._branches.run(fns, pre_seq.device, ops.PARALLEL_BRANCHES)
            z_context, z_mu, z_log_var = res[-1]
            if self.input_context != 'none':
                audio, text = res[0], res[1]
                assert audio.shape[1] == text.shape[1]
            out = self._decode(self._context(pre_seq, audio, text, z_context), nz, out_slope=1.0)
        return out, z_context, z_mu, z_log_var


class ConvDiscriminatorTriModal(nn.Module):
    """:390-435 (named ``ConvDiscriminator`` in the _abl_aff variant).  Both ``nn.LeakyReLU(True)`` are
    identities; ``out2`` in-features default to n_poses - 6 = 28 as hard-coded upstream."""

    def __init__(self, input_size, n_poses=34):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = 64
        self.pre_conv = nn.Sequential(
            nn.Conv1d(input_size, 16, 3), nn.BatchNorm1d(16), nn.LeakyReLU(True),
            nn.Conv1d(16, 8, 3), nn.BatchNorm1d(8), nn.LeakyReLU(True),
            nn.Conv1d(8, 8, 3),
        )
        self.gru = GRU(8, hidden_size=self.hidden_size, num_layers=4, bidirectional=True, dropout=0.3,
                       batch_first=True)
        self.out = nn.Linear(self.hidden_size, 1)
        self.out2 = nn.Linear(n_poses - 6, 1)
        self.do_flatten_parameters = False

    def forward(self, poses, in_text=None):
        pc = self.pre_conv
        with noise_pass(poses.device) as nz:
            x = ops.batch_norm_act(ops.conv1d_nlc(poses, pc[0].weight, pc[0].bias), pc[1], slope=1.0)
            x = ops.batch_norm_act(ops.conv1d_nlc(x, pc[3].weight, pc[3].bias), pc[4], slope=1.0)
            x = ops.conv1d_nlc(x, pc[6].weight, pc[6].bias)
            h = self.gru.run(x, nz, sum_dirs=True)
            y = ops.linear(h, self.out.weight, self.out.bias).squeeze(2)
            return ops.linear(y, self.out2.weight, self.out2.bias, act=ACT_SIGMOID)


ConvDiscriminator = ConvDiscriminatorTriModal     # name used by net/multimodal_context_net_v2_abl_aff.py:394


# the three text-encoder passes of a step as one batch (TextEncoderTCN.forward_passes): measured neutral on the step
# (-0.5 %: the clip-resident TemporalConvNet is bound by its weight stream from L2, not by its 128 workgroups), off
LOCKSTEP_TEXT = False


class PoseGenerator(nn.Module, _SpeakerZ):
    """:438-546.  forward(pre_seq (B,T,28), in_text (B,T) i64, in_mfcc (B,37,71), vid_indices (B,) i64)
    -> (poses (B,T,27), z_context, z_mu, z_log_var)."""

    audio_kind = 'mfcc'

    def __init__(self, args, pose_dim, n_words, word_embed_size, word_embeddings, mfcc_length, num_mfcc, time_steps,
                 z_obj=None):
        super().__init__()
        self.pre_length = args.n_pre_poses
        self.gen_length = args.n_poses - args.n_pre_poses
        self.z_obj = z_obj
        self.input_context = args.input_context
        self.mfcc_feature_length = 32
        self.text_feature_length = 32
        self.pose_feature_length = 8
        if self.input_context == 'both':
            self.in_size = self.mfcc_feature_length + self.text_feature_length + self.pose_feature_length
        elif self.input_context == 'audio':
            self.in_size = self.mfcc_feature_length + self.pose_feature_length
        elif self.input_context == 'text':
            self.in_size = self.text_feature_length + self.pose_feature_length
        elif self.input_context == 'none':
            self.in_size = self.pose_feature_length
        self.audio_encoder = self._make_audio_encoder(mfcc_length, num_mfcc, time_steps)
        self.text_encoder = TextEncoderTCN(args, n_words, word_embed_size, pre_trained_embedding=word_embeddings,
                                           dropout=args.dropout_prob)
        self.aff_encoder = AffEncoder()
        self._build_speaker(z_obj)
        self.hidden_size = args.hidden_size_s2eg
        self.gru = GRU(self.in_size, hidden_size=self.hidden_size, num_layers=args.n_layers, batch_first=True,
                       bidirectional=True, dropout=args.dropout_prob)
        self.out = nn.Sequential(nn.Linear(self.hidden_size, self.hidden_size // 2), nn.LeakyReLU(inplace=True),
                                 nn.Linear(self.hidden_size // 2, pose_dim))
        self.do_flatten_parameters = False
        self._branches = ops.BranchStreams(3)

    def _make_audio_encoder(self, mfcc_length, num_mfcc, time_steps):
        return MFCCEncoder(mfcc_length, num_mfcc, time_steps)

    # The trainer runs this generator three times per step on the SAME (pre_seq, in_mfcc) with the same weights
    # (processor_v2.py:792-941: for D, for the loss, with shuffled speakers).  The pose and audio encoders have no
    # dropout, so all three passes compute identical tensors there: ``share_passes = k`` (set by the trainer, None =
    # off) runs them once per step -- with autograd on, because the loss pass back-propagates through them -- and lets
    # their BatchNorm running statistics advance k times (ops.bn_repeat), exactly as k separate passes would.
    share_passes = None

    def prepare_shared(self, pre_seq, in_mfcc, audio_stream=None):
        """Compute this step's shared encoder outputs: the pose encoder on the CURRENT stream (the trainer calls this on
        a forked stream, beside the text encoder of the first pass), the audio encoder on ``audio_stream`` when given (a
        second stream forked by the trainer -- the two encoders share nothing, and because backward kernels run on the
        stream of their forward op their two backward chains, the tail of the generator's backward pass, then overlap
        as well).  Later passes wait for the events recorded here before using the outputs."""
        self._shared_encoders(pre_seq, in_mfcc, audio_stream)

    def _shared_encoders(self, pre_seq, in_mfcc, audio_stream=None):
        k = self.share_passes
        if not k or self.input_context == 'none':
            return None
        key = (ops.generation(), pre_seq.data_ptr(), in_mfcc.data_ptr(), self.training, int(k))
        if getattr(self, '_shared', None) is None or self._shared[0] != key:
            cur = torch.cuda.current_stream()
            with torch.set_grad_enabled(self.training), ops.bn_repeat(k if self.training else 1):
                pre = self.aff_encoder(pre_seq[..., :-1])
                evs = [torch.cuda.Event()]
                evs[0].record()
                if audio_stream is None or audio_stream == cur:
                    audio = self.audio_encoder(in_mfcc)
                    evs[0].record()
                    made_on = (cur,)
                else:
                    with torch.cuda.stream(audio_stream):
                        audio = self.audio_encoder(in_mfcc)
                        evs.append(torch.cuda.Event())
                        evs[1].record()
                    made_on = (cur, audio_stream)
            self._shared = (key, pre, audio, evs, made_on)
        _, pre, audio, evs, made_on = self._shared
        now = torch.cuda.current_stream()
        for ev, s in zip(evs, made_on):
            if now != s:
                now.wait_event(ev)
        if not torch.is_grad_enabled():
            pre, audio = pre.detach(), audio.detach()
        return pre, audio

    def forward(self, pre_seq, in_text, in_mfcc, vid_indices=None):
        with noise_pass(pre_seq.device) as nz:
            audio = text = None
            # four independent encoder branches -> four streams (joined before the concat that feeds the GRU)
            if self.share_passes and self.input_context != 'none':
                # pose + audio encoders of this step exist already or are made now; they are fetched LAST, so that a
                # wait for the stream that produced them sits behind this pass's own text encoder
                fns = [lambda: None, lambda: self._z(in_text, vid_indices, nz), lambda: self.text_encoder(in_text)[0],
                       lambda: self._shared_encoders(pre_seq, in_mfcc)]
                res = [f() for f in fns]
                res[0], res[3] = res[3][0], res[3][1]
                fns = None
            else:
                fns = [lambda: self.aff_encoder(pre_seq[..., :-1]), lambda: self._z(in_text, vid_indices, nz)]
                if self.input_context != 'none':
                    fns += [lambda: self.text_encoder(in_text)[0], lambda: self.audio_encoder(in_mfcc)]
            if fns is not None:
                res = self._branches.run(fns, pre_seq.device, ops.PARALLEL_BRANCHES)
            pre, (z_context, z_mu, z_log_var) = res[0], res[1]
            if self.input_context != 'none':
                text, audio = res[2], res[3]
                assert audio.shape[1] == text.shape[1], \
                    'Audio and text features must have the same number of time steps. ' \
                    'Found time steps: audio features: {}, text features: {}.'.format(audio.shape[1], text.shape[1])
            out = self._decode(self._context(pre, audio, text, z_context), nz, out_slope=0.01)
            z_mu, z_log_var = self._cut_here(z_mu, z_log_var)
        return out, z_context, z_mu, z_log_var


    def forward_passes(self, pre_seq, in_text, in_mfcc, passes):
        """Several passes of this generator over the same inputs in LOCKSTEP: ``passes`` = [(vid_indices, noise
        snapshot, grad), ...]; only the first may carry autograd.  The trainer's three passes of a step
        (processor_v2.py:798, :823, :909) differ in noise (and speakers) only, and the recurrent decoder is bound by
        the latency of its T exchanges, not by its width: each decoder layer of all passes is ONE cooperative launch.
        Needs the shared encoders (``share_passes``).  Returns [(out, z_context, z_mu, z_log_var), ...]."""
        from ..noise import use_pass
        assert self.share_passes and self.input_context != 'none' and not any(g for _, _, g in passes[1:])
        outer = torch.is_grad_enabled()
        feats = []
        texts = None
        if LOCKSTEP_TEXT and len(passes) > 1 and self.text_encoder.lockstep_capable(in_text):
            texts = self.text_encoder.forward_passes(in_text, [nz for _, nz, _ in passes], grad=passes[0][2])
        for k, (vid, nz, grad) in enumerate(passes):
            with torch.set_grad_enabled(bool(grad) and outer), use_pass(nz):
                z_context, z_mu, z_log_var = self._z(in_text, vid, nz)
                text = texts[k] if texts is not None else self.text_encoder(in_text)[0]
                pre, audio = self._shared_encoders(pre_seq, in_mfcc)
                assert audio.shape[1] == text.shape[1]
                in_data = self._context(pre, audio, text, z_context)
                (in_data,) = self._cut_here(in_data)
                feats.append((in_data, z_context, z_mu, z_log_var))
        with torch.set_grad_enabled(bool(passes[0][2]) and outer):
            hs = self.gru.run(feats[0][0], passes[0][1], True,
                              mates=[(f[0], p[1]) for f, p in zip(feats[1:], passes[1:])])
        if len(passes) == 1:
            hs = (hs,)
        res = []
        for k, (h, (_, z_context, z_mu, z_log_var), (_, nz, grad)) in enumerate(zip(hs, feats, passes)):
            with torch.set_grad_enabled(bool(grad) and outer), use_pass(nz):       # (each pass's tail in ITS noise scope)
                h = ops.linear(h, self.out[0].weight, self.out[0].bias, act=ACT_LEAKY, slope=0.01)
                out = ops.linear(h, self.out[2].weight, self.out[2].bias)
                z_mu, z_log_var = self._cut_here(z_mu, z_log_var)
            res.append((out, z_context, z_mu, z_log_var))
        return res


class AffDiscriminator(nn.Module):
    """:549-585.  ``out2`` in-features are n_poses (34 upstream, hard-coded at :562); kept as a keyword so the
    136-frame configuration can be built without patching."""

    def __init__(self, input_size, coords=3, n_poses=34):
        super().__init__()
        self.input_size = input_size
        self.coords = coords
        self.hidden_size = 64
        self.aff_encoder = AffEncoder(coords=coords)
        self.gru = GRU(8, hidden_size=self.hidden_size, num_layers=4, bidirectional=True, dropout=0.3,
                       batch_first=True)
        self.out = nn.Linear(self.hidden_size, 1)
        self.out2 = nn.Linear(n_poses, 1)
        self.activation = nn.LeakyReLU(inplace=True)
        self.do_flatten_parameters = False

    def forward(self, poses, in_text=None):
        with noise_pass(poses.device) as nz:
            feat = self.aff_encoder(poses)
            h = self.gru.run(feat, nz, sum_dirs=True)
            y = ops.linear(h, self.out.weight, self.out.bias).squeeze(2)                 # (B, T)
            return ops.linear(y, self.out2.weight, self.out2.bias, act=ACT_SIGMOID)
