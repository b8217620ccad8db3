"""Audio-ablation generator: ``net.multimodal_context_net_v2_abl_audio.PoseGenerator`` of the reference
(net/multimodal_context_net_v2_abl_audio.py:413-521) -- the v2 generator with the raw-waveform WavEncoder in
place of the MFCCEncoder (BASELINE config 4, the Conv1d roofline run).  Same constructor signature as the
v2 generator (the mfcc arguments are accepted and unused, as upstream)."""
from .multimodal_context_net_v2 import (AffDiscriminator, AffEncoder, ConvDiscriminator,  # noqa: F401
                                        ConvDiscriminatorTriModal, PoseGenerator as _PoseGeneratorV2,
                                        PoseGeneratorTriModal, TextEncoderTCN, WavEncoder)


class PoseGenerator(_PoseGeneratorV2):
    audio_kind = 'wav'

    def _make_audio_encoder(self, mfcc_length, num_mfcc, time_steps):
        return WavEncoder()

    def forward(self, pre_seq, in_text, in_audio, vid_indices=None):
        return super().forward(pre_seq, in_text, in_audio, vid_indices)
