"""Affective-encoder ablation: ``net.multimodal_context_net_v2_abl_aff`` of the reference
(net/multimodal_context_net_v2_abl_aff.py:285-392 ``PoseGenerator``, :394-439 ``ConvDiscriminator``) -- the second
trainable configuration of the GAN step: the generator feeds the raw seed-pose sequence (pose_dim + 1 = 28 columns,
constraint bit included) straight into the recurrent decoder instead of the 8 AffEncoder features, and is trained
against the convolutional discriminator.  Same constructor / forward signatures and ``state_dict`` keys as upstream (no
``aff_encoder.*`` entries; ``gru.weight_ih_l0`` is (3H, 108)); every forward runs through the HIP kernels in ``ops``.
"""
import torch.nn as nn

from .. import ops
from ..noise import noise_pass
from .multimodal_context_net_v2 import (GRU, ConvDiscriminator, ConvDiscriminatorTriModal, MFCCEncoder,  # noqa: F401
                                        PoseGeneratorTriModal, TextEncoderTCN, WavEncoder, _SpeakerZ)


class PoseGenerator(nn.Module, _SpeakerZ):
    """:285-392.  forward(pre_seq (B,T,28), in_text (B,T) i64, in_mfcc (B,37,71), vid_indices (B,) i64)
    -> (poses (B,T,27), z_context, z_mu, z_log_var)."""

    audio_kind = 'mfcc'
    share_passes = None          # nothing to share across the passes of a step: no dropout-free pose encoder here

    def __init__(self, args, pose_dim, n_words, word_embed_size, word_embeddings, mfcc_length, num_mfcc, time_steps,
                 z_obj=None):
        super().__init__()
        self.pre_length = args.n_pre_poses
        self.gen_length = args.n_poses - args.n_pre_poses
        self.z_obj = z_obj
        self.input_context = args.input_context
        self.mfcc_feature_length = 32
        self.text_feature_length = 32
        self.pose_feature_length = pose_dim + 1
        self.in_size = self.pose_feature_length + {'both': 64, 'audio': 32, 'text': 32, 'none': 0}[self.input_context]
        self.audio_encoder = MFCCEncoder(mfcc_length, num_mfcc, time_steps)
        self.text_encoder = TextEncoderTCN(args, n_words, word_embed_size, pre_trained_embedding=word_embeddings,
                                           dropout=args.dropout_prob)
        self._build_speaker(z_obj)
        self.hidden_size = args.hidden_size_s2eg
        self.gru = GRU(self.in_size, hidden_size=self.hidden_size, num_layers=args.n_layers, batch_first=True,
                       bidirectional=True, dropout=args.dropout_prob)
        self.out = nn.Sequential(nn.Linear(self.hidden_size, self.hidden_size // 2), nn.LeakyReLU(inplace=True),
                                 nn.Linear(self.hidden_size // 2, pose_dim))
        self.do_flatten_parameters = False

    def prepare_shared(self, pre_seq, in_mfcc):
        pass

    def forward(self, pre_seq, in_text, in_mfcc, vid_indices=None):
        with noise_pass(pre_seq.device) as nz:
            audio = text = None
            z_context, z_mu, z_log_var = self._z(in_text, vid_indices, nz)
            if self.input_context != 'none':
                text = self.text_encoder(in_text)[0]
                audio = self.audio_encoder(in_mfcc)
                assert audio.shape[1] == text.shape[1], \
                    'Audio and text features must have the same number of time steps. ' \
                    'Found time steps: audio features: {}, text features: {}.'.format(audio.shape[1], text.shape[1])
            out = self._decode(self._context(pre_seq, audio, text, z_context), nz, out_slope=0.01)
            z_mu, z_log_var = self._cut_here(z_mu, z_log_var)
        return out, z_context, z_mu, z_log_var
