"""Pose auto-encoder behind the Frechet Gesture Distance: drop-in for the 'pose' mode of ``net.embedding_net`` of the
reference (net/embedding_net.py:10-13 re_parametrize, :42-83 PoseEncoderConv, :165-217 PoseDecoderConv, :262-308
EmbeddingNet) -- the mode ``EmbeddingSpaceEvaluator`` builds (net/embedding_space_evaluator.py:20-27).  Same class names,
constructor / forward signatures, return tuples and ``state_dict`` keys; the torch ``nn`` modules are parameter
containers, every forward runs through the HIP kernels in ``ops`` on channels-last tensors.  The speech ('speech' /
'random') modes need the v1 context encoder, which is outside the v2 path: they raise.
"""
import torch
import torch.nn as nn

from .. import ops
from .._lib import ACT_LEAKY, ACT_NONE
from ..noise import new_site, noise_pass


def re_parametrize(mu, log_var, noise=None, site=None):
    """net/embedding_net.py:10-13: mu + eps * exp(0.5 * log_var), eps from the device noise state."""
    if noise is None:
        with noise_pass(mu.device) as nz:
            return ops.reparametrize(mu, log_var, nz, _RP_SITE[0] if site is None else site)
    return ops.reparametrize(mu, log_var, noise, _RP_SITE[0] if site is None else site)


_RP_SITE = [new_site()]


def conv_norm_relu(in_channels, out_channels, down_sample=False, padding=0, batch_norm=True):
    """:16-39 (parameter container; ``_conv_block`` below runs it)."""
    k, s = (4, 2) if down_sample else (3, 1)
    conv_block = nn.Conv1d(in_channels, out_channels, kernel_size=k, stride=s, padding=padding)
    if batch_norm:
        return nn.Sequential(conv_block, nn.BatchNorm1d(out_channels), nn.LeakyReLU(0.2, True))
    return nn.Sequential(conv_block, nn.LeakyReLU(0.2, True))


def _conv_block(block, x):
    conv = block[0]
    has_bn = isinstance(block[1], nn.BatchNorm1d)
    if has_bn:
        y = ops.conv1d_nlc(x, conv.weight, conv.bias, stride=conv.stride[0], pad=conv.padding[0],
                           bn_stats=block[1].training)
        return ops.batch_norm_act(y, block[1], slope=0.2)
    return ops.conv1d_nlc(x, conv.weight, conv.bias, stride=conv.stride[0], pad=conv.padding[0], act=ACT_LEAKY, slope=0.2)


def _linear_bn(x, lin, bn, slope):
    """Linear + BatchNorm1d over (B, features) + leaky(slope); ``nn.LeakyReLU(True)`` upstream is slope 1.0 = identity."""
    y = ops.linear(x, lin.weight, lin.bias)
    return ops.batch_norm_act(y.unsqueeze(1), bn, slope=slope).squeeze(1)


class PoseEncoderConv(nn.Module):
    """:42-83.  (B, length, dim) -> (z, mu, log_var), each (B, 32)."""

    def __init__(self, length, dim):
        super().__init__()
        self.net = nn.Sequential(conv_norm_relu(dim, 32, batch_norm=True), conv_norm_relu(32, 64, batch_norm=True),
                                 conv_norm_relu(64, 64, True, batch_norm=True), nn.Conv1d(64, 32, 3))
        flat = 32 * ((((length - 2) - 2) - 4) // 2 + 1 - 2)        # 384 for 34 frames (864 for 64), :55-56
        self.out_net = nn.Sequential(nn.Linear(flat, 256), nn.BatchNorm1d(256), nn.LeakyReLU(True),
                                     nn.Linear(256, 128), nn.BatchNorm1d(128), nn.LeakyReLU(True), nn.Linear(128, 32))
        self.fc_mu = nn.Linear(32, 32)
        self.fc_log_var = nn.Linear(32, 32)
        self.site = new_site()

    def forward(self, poses, variational_encoding):
        x = poses                                                   # already (B, seq, dim) channels-last: no transpose
        for blk in (self.net[0], self.net[1], self.net[2]):
            x = _conv_block(blk, x)
        x = ops.conv1d_nlc(x, self.net[3].weight, self.net[3].bias)
        x = x.transpose(1, 2).contiguous().flatten(1)               # upstream flattens (B, C, L): layout glue
        on = self.out_net
        x = _linear_bn(x, on[0], on[1], 1.0)
        x = _linear_bn(x, on[3], on[4], 1.0)
        x = ops.linear(x, on[6].weight, on[6].bias)
        mu = ops.linear(x, self.fc_mu.weight, self.fc_mu.bias)
        log_var = ops.linear(x, self.fc_log_var.weight, self.fc_log_var.bias)
        z = re_parametrize(mu, log_var, site=self.site) if variational_encoding else mu
        return z, mu, log_var


class PoseDecoderConv(nn.Module):
    """:165-217.  (B, 32) [+ pre_poses] -> (B, length, dim).  ConvTranspose1d(k=3, stride 1) is run as the equivalent
    convolution: taps flipped, channel axes swapped, padding k - 1."""

    def __init__(self, length, dim, use_pre_poses=False):
        super().__init__()
        self.use_pre_poses = use_pre_poses
        feat_size = 32
        if use_pre_poses:
            self.pre_pose_net = nn.Sequential(nn.Linear(dim * 4, 32), nn.BatchNorm1d(32), nn.ReLU(), nn.Linear(32, 32))
            feat_size += 32
        if length == 64:
            self.pre_net = nn.Sequential(nn.Linear(feat_size, 128), nn.BatchNorm1d(128), nn.LeakyReLU(True),
                                         nn.Linear(128, 256))
        elif length == 34:
            self.pre_net = nn.Sequential(nn.Linear(feat_size, 64), nn.BatchNorm1d(64), nn.LeakyReLU(True),
                                         nn.Linear(64, 136))
        else:
            assert False
        self.net = nn.Sequential(nn.ConvTranspose1d(4, 32, 3), nn.BatchNorm1d(32), nn.LeakyReLU(0.2, True),
                                 nn.ConvTranspose1d(32, 32, 3), nn.BatchNorm1d(32), nn.LeakyReLU(0.2, True),
                                 nn.Conv1d(32, 32, 3), nn.Conv1d(32, dim, 3))

    @staticmethod
    def _deconv(x, ct, bn):
        w = ct.weight.permute(1, 0, 2).flip(2).contiguous()         # (Cin, Cout, k) -> conv weight (Cout, Cin, k)
        y = ops.conv1d_nlc(x, w, ct.bias, pad=ct.kernel_size[0] - 1, bn_stats=bn.training)
        return ops.batch_norm_act(y, bn, slope=0.2)

    def forward(self, feat, pre_poses=None):
        if self.use_pre_poses:
            pp = self.pre_pose_net
            f = _linear_bn(pre_poses.reshape(pre_poses.shape[0], -1), pp[0], pp[1], 0.0)     # BN + ReLU
            feat = torch.cat((ops.linear(f, pp[3].weight, pp[3].bias), feat), dim=1)
        pn = self.pre_net
        out = _linear_bn(feat, pn[0], pn[1], 1.0)
        out = ops.linear(out, pn[3].weight, pn[3].bias)
        out = out.view(feat.shape[0], 4, -1).transpose(1, 2).contiguous()                    # (B, L, 4) channels-last
        out = self._deconv(out, self.net[0], self.net[1])
        out = self._deconv(out, self.net[3], self.net[4])
        out = ops.conv1d_nlc(out, self.net[6].weight, self.net[6].bias)
        return ops.conv1d_nlc(out, self.net[7].weight, self.net[7].bias)                     # already (B, seq, dim)


class EmbeddingNet(nn.Module):
    """:262-308.  Only ``mode='pose'`` (pose encoder + convolutional decoder) is on the v2 path."""

    def __init__(self, args, pose_dim, n_frames, n_words, word_embed_size, word_embeddings, mode):
        super().__init__()
        if mode != 'pose':
            raise NotImplementedError("EmbeddingNet: only mode='pose' (the FGD evaluator's) is built; the speech modes "
                                      'need the v1 context encoder (net/multimodal_context_net_v1.py)')
        self.context_encoder = None
        self.pose_encoder = PoseEncoderConv(n_frames, pose_dim)
        self.decoder = PoseDecoderConv(n_frames, pose_dim)
        self.mode = mode

    def forward(self, in_text, in_audio, pre_poses, poses, input_mode=None, variational_encoding=False):
        if input_mode is None:
            assert self.mode is not None
            input_mode = self.mode
        context_feat = context_mu = context_log_var = None
        if poses is not None:
            poses_feat, pose_mu, pose_log_var = self.pose_encoder(poses, variational_encoding)
        else:
            poses_feat = pose_mu = pose_log_var = None
        if input_mode != 'pose':
            raise NotImplementedError("input_mode must be 'pose' (no context encoder on the v2 path)")
        out_poses = self.decoder(poses_feat, pre_poses)
        return context_feat, context_mu, context_log_var, poses_feat, pose_mu, pose_log_var, out_poses

    def freeze_pose_nets(self):
        for param in self.pose_encoder.parameters():
            param.requires_grad = False
        for param in self.decoder.parameters():
            param.requires_grad = False
