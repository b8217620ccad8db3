"""Frechet Gesture Distance on the GPU: drop-in for ``net.embedding_space_evaluator.EmbeddingSpaceEvaluator`` of the
reference (net/embedding_space_evaluator.py:15-156).  The pose auto-encoder runs through the HIP kernels; latent
features stay on the device; mean / covariance are formed there in float64; the 32 x 32 matrix square root of the Frechet
distance runs on the host (scipy), as upstream.  UMAP visualisation (:63-72) is outside the path.
"""
import os
from os.path import join as jn

import numpy as np
import torch
from scipy import linalg

from .embedding_net import EmbeddingNet


class EmbeddingSpaceEvaluator:
    def __init__(self, base_path, args, pose_dim, lang_model, device, checkpoint='outputs/embedding_net.pth.tar',
                 allow_random_init=False):
        self.n_pre_poses = args.n_pre_poses
        self.pose_dim = pose_dim
        self.net = EmbeddingNet(args, pose_dim, args.n_poses, lang_model.n_words, getattr(args, 'wordembed_dim', 300),
                                getattr(lang_model, 'word_embedding_weights', None), 'pose').to(device)
        path = jn(base_path, checkpoint) if checkpoint else None
        if path and os.path.exists(path):
            self.net.load_state_dict(torch.load(path, map_location=device)['embedding_dict'])
        elif path and not allow_random_init:
            # upstream fails in torch.load here (:26-27); a silently random auto-encoder would make FGD meaningless
            raise FileNotFoundError('{} not found (pass allow_random_init=True to score with a randomly initialised '
                                    'embedding net: values are then only comparable within one run)'.format(path))
        self.net.train(False)
        self.reset()

    def reset(self):
        self.context_feat_list = []
        self.real_feat_list = []
        self.generated_feat_list = []
        self._recon_err = []          # device pairs (generated, real); see the recon_err_diff property

    @property
    def recon_err_diff(self):
        """Python floats as upstream (:59-61); the device values are read back here, once, not per batch."""
        return self.reconstruction_error_differences()

    def get_no_of_samples(self):
        return len(self.real_feat_list)

    def push_samples(self, context_text, context_spec, generated_poses, real_poses):
        """:46-61 -- latent features of real and generated poses, and the reconstruction-error difference."""
        with torch.no_grad():
            pre_poses = real_poses[:, 0:self.n_pre_poses]
            _, _, _, real_feat, _, _, real_recon = self.net(None, None, pre_poses, real_poses, 'pose',
                                                            variational_encoding=False)
            _, _, _, generated_feat, _, _, generated_recon = self.net(None, None, pre_poses, generated_poses, 'pose',
                                                                      variational_encoding=False)
            self.real_feat_list.append(real_feat.detach())
            self.generated_feat_list.append(generated_feat.detach())
            err = torch.stack(((generated_poses - generated_recon).abs().mean(), (real_poses - real_recon).abs().mean()))
            self._recon_err.append(err)                  # read back lazily: no host sync per batch

    def reconstruction_error_differences(self):
        return [float(e[0] - e[1]) for e in torch.stack(self._recon_err).cpu()] if self._recon_err else []

    def get_scores(self):
        """:74-103 -- (frechet_dist, feat_dist)."""
        gen = torch.cat(self.generated_feat_list).double()
        real = torch.cat(self.real_feat_list).double()

        def moments(x):
            mu = x.mean(0)
            xc = x - mu
            return mu.cpu().numpy(), (xc.t() @ xc / (x.shape[0] - 1)).cpu().numpy()      # np.cov(rowvar=False)
        (mu_g, sig_g), (mu_r, sig_r) = moments(gen), moments(real)
        try:
            frechet_dist = self.calculate_frechet_distance(mu_g, sig_g, mu_r, sig_r)
        except ValueError:
            frechet_dist = 1e+10
        feat_dist = float((real - gen).abs().sum(1).mean())
        return frechet_dist, feat_dist

    @staticmethod
    def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
        """:105-156: d^2 = ||mu_1 - mu_2||^2 + Tr(C_1 + C_2 - 2 sqrt(C_1 C_2)); singular products get eps on the
        diagonals, a non-negligible imaginary part of the square root is an error."""
        mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
        sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
        assert mu1.shape == mu2.shape, 'Training and test mean vectors have different lengths'
        assert sigma1.shape == sigma2.shape, 'Training and test covariances have different dimensions'
        diff = mu1 - mu2
        cov_mean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
        if not np.isfinite(cov_mean).all():
            print('fid calculation produces singular product; adding %s to diagonal of cov estimates' % eps)
            offset = np.eye(sigma1.shape[0]) * eps
            cov_mean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
        if np.iscomplexobj(cov_mean):
            if not np.allclose(np.diagonal(cov_mean).imag, 0, atol=1e-3):
                raise ValueError('Imaginary component {}'.format(np.max(np.abs(cov_mean.imag))))
            cov_mean = cov_mean.real
        return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(cov_mean)
