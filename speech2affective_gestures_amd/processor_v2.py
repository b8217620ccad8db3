"""Trainer of the S2AG GAN step on MI355X.  Drop-in for the hot-path surface of ``processor_v2`` of the
reference (processor_v2.py:53-83 get_epoch_and_loss, :86-220 Processor.__init__, :351-365 load_model_at_epoch,
:589-638 yield_batch, :776-957 forward_pass_s2ag, :959-1069 epoch loops / train / checkpoint naming).

What is MI355X-first here (DESIGN.md has the full story):
  * one process per GPU; parameters and gradients of each network live in one flat arena, so the two
    gradient exchanges of a step are two RCCL all-reduces over xGMI (D: 1.25 MB, G: ~53 MB) and each Adam
    is a single launch;
  * the whole step (3 generator, 3 discriminator and 1 tri-modal forward, 2 backward, 2 Adam) is captured
    once into HIP graphs (three segments separated by the two collectives) and replayed -- the step is
    launch/latency bound, not bandwidth bound;
  * losses stay on the device; one 32-byte read-back per step replaces the reference's 5-7 ``.item()`` syncs.
Rendering, FGD evaluation and the LMDB/npz data pipeline of the reference are outside this path.
"""
import math
import os
import re
import time
from os.path import join as jn

import numpy as np
import torch

from . import bf16, config, noise, ops
from .net.multimodal_context_net_v2 import (AffDiscriminator, ConvDiscriminatorTriModal as CDT, PoseGenerator,
                                            PoseGeneratorTriModal as PGT)
from .optim import FusedAdam, ParamArena
from .parallel import DataParallelContext


def find_all_substr(a_str, sub):
    """Start offsets of the non-overlapping occurrences of ``sub`` (processor_v2.py:43-50 of the reference)."""
    return (m.start() for m in re.finditer(re.escape(sub), a_str))


def get_epoch_and_loss(path_to_model_files, epoch='best'):
    """File-name protocol ``epoch_{E:06d}_loss_{L:.4f}_model.pth.tar`` (processor_v2.py:53-83), including the
    upstream quirk that 'best' picks the SECOND smallest loss (``argpartition(.., 2)[1]``) once >= 3 files exist."""
    all_models = os.listdir(path_to_model_files)
    if len(all_models) < 2:
        return '', None, np.inf

    def parse(name):
        us = list(find_all_substr(name, '_'))
        return name, int(name[us[0] + 1:us[1]]), float(name[us[2] + 1:us[3]])
    if epoch == 'best':
        losses = -1. * np.ones(len(all_models))
        for i, name in enumerate(all_models):
            parts = name.split('_')
            if len(parts) > 1:
                losses[i] = float(parts[3])
        if len(losses) < 3:
            pick = all_models[np.argwhere(losses == min(v for v in losses if v > 0))[0, 0]]
        else:
            pick = all_models[np.argpartition(losses, 2)[1]]
        return parse(pick)
    assert isinstance(epoch, int)
    for name in all_models:
        parts = name.split('_')
        if len(parts) > 1 and epoch == int(parts[1]):
            return parse(name)
    return '', None, np.inf


class _Log:
    """Stand-in for torchlight IO.print_log (torchlight/io.py:121-130): stdout + work_dir/log.txt."""

    def __init__(self, work_dir, save_log=True, print_log=True, rank=0):
        self.work_dir, self.save_log, self.print_to_screen, self.rank = work_dir, save_log, print_log, rank

    def print_log(self, s, print_time=True):
        if self.rank != 0:
            return
        if print_time:
            s = time.strftime('[%m.%d.%y|%X] ', time.localtime()) + s
        if self.print_to_screen:
            print(s)
        if self.save_log and self.work_dir:
            os.makedirs(self.work_dir, exist_ok=True)
            with open(jn(self.work_dir, 'log.txt'), 'a') as f:
                print(s, file=f)

    def print_timer(self):
        pass


PASSES_PER_STEP = 7      # forward passes (= noise snapshots) of one GAN step, in the reference's order: G(dis), D(real),
                         # D(fake), tri-modal baseline, G(main), D(gen), G(rand)


class _GraphSegments:
    """Capture ``fns`` as consecutive HIP graphs sharing one memory pool; ``between[i]`` runs eagerly after
    segment i (the RCCL all-reduces).  Static shapes; inputs are copied into the buffers captured."""

    def __init__(self, fns, between, warmup=3, before=None):
        self.fns, self.between, self.before = fns, between, before
        self.graphs = []
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        if self.before is not None:
            self.before()              # host-side state the captured functions branch on (GradExchange.precheck)
        for i, fn in enumerate(fns):
            g = torch.cuda.CUDAGraph()
            # 'thread_local': the batch feeder's thread allocates / records events on its own stream meanwhile
            with torch.cuda.graph(g, pool=pool, capture_error_mode='thread_local'):
                fn()
            self.graphs.append(g)      # nothing executes during capture, so no collective here

    def _eager(self):
        if self.before is not None:
            self.before()
        for fn, btw in zip(self.fns, self.between):
            fn()
            if btw is not None:
                btw()

    def replay(self):
        if self.before is not None:
            self.before()
        for g, btw in zip(self.graphs, self.between):
            g.replay()
            if btw is not None:
                btw()


class Processor(object):
    """Processor for emotive gesture generation (hot-path subset)."""

    @staticmethod
    def regulariser_branch(cfg) -> bool:
        """The generator-loss branch of the step (processor_v2.py:899-934), fixed at construction like every other static shape
        of the step.  True: z_type 'speaker' and loss_reg_weight > 0 -- the generator runs a third time with shuffled speakers
        and the loss carries the divergence regulariser and the KLD term (:901-929).  False: z_type 'none' or loss_reg_weight
        == 0 -- the loss is the regression term (+ the GAN term) alone and that pass does not exist (:933-934).
        z_type 'random' with the regulariser on hands vid_indices = None to a generator whose z_obj is the speaker Vocab (the
        only kind this class and the reference's build, :140): the reference's own step dies on `assert vid_indices is not
        None` (net/multimodal_context_net_v2.py:511) at its first batch -- here the same configuration fails at construction."""
        z_type = getattr(cfg, 'z_type', 'speaker')
        # any other string is the no-regulariser branch, as upstream (:899 tests for the two names and nothing else)
        on = z_type in ('speaker', 'random') and float(cfg.loss_reg_weight) > 0.0
        if on and z_type == 'random':
            raise ValueError("z_type 'random' with loss_reg_weight > 0 calls the generator with vid_indices=None "
                             "(processor_v2.py:906-910), which a generator built with the speaker model as z_obj -- the only "
                             "kind Processor builds -- refuses (assert vid_indices is not None, "
                             "net/multimodal_context_net_v2.py:511): the reference fails on its first batch with this "
                             "configuration; use z_type 'speaker', or 'none' / loss_reg_weight 0 for no regulariser")
        return on

    def __init__(self, base_path, args, s2ag_config_args, data_loader, pose_dim, coords, audio_sr,
                 min_train_epochs=20, zfill=6):
        if not torch.cuda.is_available():
            raise RuntimeError('speech2affective_gestures_amd.Processor needs an MI355X (HIP); there is no CPU path')
        self.base_path = base_path
        self.args = args
        self.dp = DataParallelContext.from_env()
        self.device = torch.device('cuda', self.dp.local_rank)
        torch.cuda.set_device(self.device)
        self.s2ag_config_args = s2ag_config_args
        self.data_loader = data_loader
        self.result, self.iter_info, self.epoch_info = dict(), dict(), dict()
        self.meta_info = dict(epoch=0, iter=0)
        self.io = _Log(getattr(args, 'work_dir_s2ag', None), getattr(args, 'save_log', False),
                       getattr(args, 'print_log', True), self.dp.rank)
        self.pose_dim, self.coords, self.audio_sr = pose_dim, coords, audio_sr

        key = 'train_data_s2ag' if getattr(args, 'train_s2ag', True) else 'test_data_s2ag'
        meta = data_loader[key]
        self.time_steps = meta.n_poses
        self.audio_length = meta.expected_audio_length
        self.num_mfcc = meta.num_mfcc_combined
        self.lang_model = meta.lang_model
        self.mfcc_length = int(np.ceil(self.audio_length / 512))

        self.best_s2ag_loss = np.inf
        self.best_s2ag_loss_epoch = None
        self.s2ag_loss_updated = False
        self.min_train_epochs = min_train_epochs
        self.zfill = zfill

        self.train_speaker_model = data_loader['train_data_s2ag'].speaker_model
        self.val_speaker_model = getattr(data_loader.get('val_data_s2ag', meta), 'speaker_model', None)
        wemb = getattr(self.lang_model, 'word_embedding_weights', None)
        cfg = self.s2ag_config_args
        self.trimodal_generator = PGT(cfg, pose_dim=pose_dim, n_words=self.lang_model.n_words,
                                      word_embed_size=cfg.wordembed_dim, word_embeddings=wemb,
                                      z_obj=self.train_speaker_model)
        self.trimodal_discriminator = CDT(pose_dim, n_poses=self.time_steps)
        # The reference selects its ablations by importing PoseGenerator / the discriminator from another net module
        # (net/multimodal_context_net_v2_abl_aff.py, ..._abl_audio.py); here ``args.ablation`` names the pairing:
        # 'none' (default) | 'aff' (no affective encoder, ConvDiscriminator) | 'audio' (raw-waveform encoder, use_mfcc False)
        self.ablation = getattr(args, 'ablation', 'none') or 'none'
        self.use_mfcc = self.ablation != 'audio'
        if self.ablation == 'aff':
            from .net.multimodal_context_net_v2_abl_aff import ConvDiscriminator as Dis, PoseGenerator as Gen
        elif self.ablation == 'audio':
            from .net.multimodal_context_net_v2_abl_audio import PoseGenerator as Gen
            Dis = AffDiscriminator
        elif self.ablation == 'none':
            Gen, Dis = PoseGenerator, AffDiscriminator
        else:
            raise ValueError("args.ablation must be 'none', 'aff' or 'audio'")
        self.s2ag_generator = Gen(cfg, pose_dim=pose_dim, n_words=self.lang_model.n_words,
                                  word_embed_size=cfg.wordembed_dim, word_embeddings=wemb,
                                  mfcc_length=self.mfcc_length, num_mfcc=self.num_mfcc,
                                  time_steps=self.time_steps, z_obj=self.train_speaker_model)
        self.s2ag_discriminator = Dis(pose_dim, n_poses=self.time_steps)
        for m in (self.trimodal_generator, self.trimodal_discriminator, self.s2ag_generator,
                  self.s2ag_discriminator):
            m.to(self.device)
        for p in self.trimodal_generator.parameters():      # frozen baseline (processor_v2.py:1033-1034)
            p.requires_grad_(False)

        ops.init_tickets(self.device)
        # flat arenas (must come after .to(device)); identical initial weights on every rank
        # generator arena = [word embedding | encoders | GRU decoder + out]: reverse order of the backward pass, so the
        # data-parallel exchange works on contiguous buckets (parallel.GradExchange)
        emb = self.s2ag_generator.text_encoder.embedding.weight
        self.gen_arena = ParamArena(self.s2ag_generator.parameters(), first=[emb])
        self.dis_arena = ParamArena(self.s2ag_discriminator.parameters())
        self.gen_exchange = None          # built at the first data-parallel step (see _exchange)
        self.dp.broadcast_module(self.s2ag_generator, self.gen_arena)
        self.dp.broadcast_module(self.s2ag_discriminator, self.dis_arena)
        self.dp.broadcast_module(self.trimodal_generator, None)

        self.test_samples, self.num_test_samples = None, 0
        self.train_samples = getattr(data_loader['train_data_s2ag'], 'samples', None)
        self.val_samples = getattr(data_loader.get('val_data_s2ag', meta), 'samples', None)
        self.num_train_samples = getattr(data_loader['train_data_s2ag'], 'n_samples', 0)
        self.num_val_samples = getattr(data_loader.get('val_data_s2ag', meta), 'n_samples', 0)

        self.lr_s2ag_gen = cfg.learning_rate
        self.lr_s2ag_dis = cfg.learning_rate * cfg.discriminator_lr_weight
        self.s2ag_gen_optimizer = FusedAdam(self.gen_arena, lr=self.lr_s2ag_gen, betas=(0.5, 0.999))
        self.s2ag_dis_optimizer = FusedAdam(self.dis_arena, lr=self.lr_s2ag_dis, betas=(0.5, 0.999))

        self.use_div_reg = self.regulariser_branch(cfg)
        # the weights of the generator loss are read ONCE, with the branch: a captured step holds them as launch arguments,
        # so a later edit of cfg must not change the eager step either (it would silently mix branches); _graph_key carries both
        self._loss_weights = (float(cfg.loss_regression_weight), float(cfg.loss_reg_weight), float(cfg.loss_kld_weight))
        self.use_hip_graph = bool(getattr(args, 'hip_graph', True))
        # independent forward passes of a step run on forked streams (every kernel here fills only part of the chip)
        self.overlap_passes = bool(getattr(args, 'overlap_passes', True))
        # deterministic mode (debug; config switch DETERMINISTIC): every pass of the step on ONE stream, accumulating launches
        # ordered by workgroup index inside the library -- two runs from the same state are bit-identical (ops.set_deterministic)
        self.deterministic = bool(getattr(args, 'deterministic', config.get('DETERMINISTIC')))
        if self.deterministic:
            self.overlap_passes = False
        # process-wide library state: the newest trainer's choice holds, on OR off (a trainer that forks passes while an
        # earlier one left the turn word installed would dead-lock two concurrent accumulating launches)
        ops.set_deterministic(self.deterministic, self.device)
        # the generator's dropout-free encoders run once per step instead of once per pass (see PoseGenerator)
        self.share_encoders = bool(getattr(args, 'share_encoders', True)) \
            and hasattr(self.s2ag_generator, '_shared_encoders')
        self.encoders_aside = bool(getattr(args, 'encoders_aside', True))
        # ... and the audio encoder of the shared pair on a stream of its own (forward and, since backward kernels run on
        # the stream of their forward op, backward: the two chains of small launches end the generator's backward pass)
        self.encoders_apart = self.encoders_aside and \
            bool(getattr(args, 'encoders_apart', True))
        self.early_real_backward = bool(getattr(args, 'early_real_backward', True))
        self.early_rand = bool(getattr(args, 'early_rand', True)) and self.use_div_reg     # no third pass without the regulariser
        # the loss pass of the generator (with autograd) and the frozen tri-modal baseline read nothing the D step
        # writes: they run beside the D step instead of at the head of the generator phase (see _dis_phase)
        self.early_main = int(getattr(args, 'early_main', 3))
        self._side = [torch.cuda.Stream(device=self.device) for _ in range(4)]
        self._graphed = None
        self.last_losses = {}
        self._last_outs = None
        # FGD evaluators of forward_pass_s2ag(calculate_metrics=True) (processor_v2.py:199-206 builds them from
        # outputs/embedding_net.pth.tar); None = the three meters only, as upstream's `if evaluator:`
        self.evaluator = self.evaluator_trimodal = None

    def _exchange(self):
        """The generator's gradient-exchange schedule, or None outside data-parallel runs."""
        if not self.dp.active:
            return None
        if self.gen_exchange is None:
            self.gen_exchange = self._make_gen_exchange(self.s2ag_generator.text_encoder.embedding.weight)
        return self.gen_exchange

    def _make_gen_exchange(self, emb):
        """Buckets of the generator's gradient exchange (see parallel.GradExchange): A = GRU decoder + out (behind the
        backward cut), B = the encoders, touched rows for the word embedding.  Row capacity: a batch of B clips touches
        at most B * K distinct rows, K = the largest number of distinct ids in one clip (PAD included) -- from
        ``args.max_words_per_clip`` (+1 for PAD) or the training set itself; without either, B * T (always safe)."""
        from .parallel import GradExchange, RowKernels
        G, ar = self.s2ag_generator, self.gen_arena
        split = min(ar.offset_of(p) for p in list(G.gru.parameters()) + list(G.out.parameters()))
        assert all(ar.offset_of(p) >= split for p in list(G.gru.parameters()) + list(G.out.parameters()))
        rows, cap, kern = None, 0, None
        if emb.requires_grad and config.get('SPARSE_EMBEDDING'):
            assert ar.offset_of(emb) == 0
            n_entries, dim = emb.shape
            B, T = int(self.args.batch_size), int(self.time_steps)
            k = getattr(self.args, 'max_words_per_clip', None)
            if k is not None:
                k = int(k) + 1
            else:
                seqs = (self.train_samples or {}).get('extended_word_seq') if isinstance(self.train_samples, dict) else None
                if seqs is not None and len(seqs):
                    srt = np.sort(np.asarray(seqs), axis=1)
                    k = int((np.diff(srt, axis=1) != 0).sum(axis=1).max()) + 1
            cap = min(n_entries, B * min(T, k if k is not None else T))
            rows = (0, n_entries * dim, n_entries, dim)
            kern = RowKernels(unique=lambda i, n, u: ops.rows_unique_raw(i, n, u, flag_overflow=False),
                              pack=ops.rows_pack_raw, merge=ops.rows_merge_raw,
                              unique_flagged=lambda i, n, u: ops.rows_unique_raw(i, n, u, flag_overflow=True))
        return GradExchange(self.dp, ar.grad, split, rows=rows, row_cap=cap, kernels=kern)

    # ------------------------------------------------------------------------------------------------
    def count_parameters(self):
        return sum(p.numel() for p in self.s2ag_generator.parameters() if p.requires_grad)

    def load_model_at_epoch(self, epoch='best'):
        model_name, self.best_s2ag_loss_epoch, self.best_s2ag_loss = \
            get_epoch_and_loss(self.args.work_dir_s2ag, epoch=epoch)
        try:
            loaded_vars = torch.load(jn(self.args.work_dir_s2ag, model_name), map_location=self.device)
            self.s2ag_generator.load_state_dict(loaded_vars['gen_model_dict'])
            self.s2ag_discriminator.load_state_dict(loaded_vars['dis_model_dict'])
            return True
        except (FileNotFoundError, IsADirectoryError):
            print('Warning! No saved model found.' if epoch == 'best'
                  else 'Warning! No saved model found at epoch {}.'.format(epoch))
            return False

    def _sync_error_flag(self):
        """Data parallel: the sticky error word becomes the MAX over ranks before an optimizer reads it (parallel.py)."""
        self.dp.sync_error_flag(ops.coop_error_flag(self.device))

    def save_model(self, epoch, loss):
        # never write weights of a run whose sticky error word is raised ON ANY RANK: the word was MAX-reduced over the ranks
        # before the last Adam launch of the last step (and the fused Adam refused to step on every rank alike), so the local
        # word IS the reduced one here.  No collective in this method: which ranks reach it depends on their own validation
        # loss (train(): s2ag_loss_updated), and an unmatched collective would hang the others.
        flag = ops.coop_error_flag(self.device)
        if flag is not None:
            ops.check_coop_flag(flag.item())
        if self.dp.rank != 0:
            return None
        os.makedirs(self.args.work_dir_s2ag, exist_ok=True)
        path = jn(self.args.work_dir_s2ag, 'epoch_{:06d}_loss_{:.4f}_model.pth.tar'.format(epoch, loss))
        torch.save({'gen_model_dict': self.s2ag_generator.state_dict(),
                    'dis_model_dict': self.s2ag_discriminator.state_dict()}, path)
        return path

    # ------------------------------------------------------------------------------------------------
    def load_cache(self, part, dir_name, load_full=True):
        """processor_v2.py:222-271: the cached TED arrays of ``part`` ('train' / 'val' / 'test') in the reference's npz
        layout -- ``<dir_name>/../full/<part>.npz`` with keys extended_word_seq (N, T) int64, vec_seq (N, T, pose_dim),
        audio (N, L) int16, audio_max (N,), mfcc_features (N, num_mfcc, mfcc_length) [kept as float16], vid_indices (N,),
        or one ``<dir_name>/<k:06d>.npz`` per clip when ``load_full`` is False.  (Writing the cache reads the LMDB
        data set and is data preparation: not on this path.)"""
        keys = ('extended_word_seq', 'vec_seq', 'audio', 'audio_max', 'mfcc_features', 'vid_indices')
        if load_full:
            with np.load(jn(dir_name, '../full', part + '.npz'), allow_pickle=True) as npz:
                samples = {k: npz[k] for k in keys}
        else:
            num = getattr(self, {'train': 'num_train_samples', 'val': 'num_val_samples'}.get(part, 'num_test_samples'), 0)
            rows = {k: [] for k in keys}
            for i in range(num):
                with np.load(jn(dir_name, str(i).zfill(self.zfill) + '.npz'), allow_pickle=True) as npz:
                    for k in keys:
                        rows[k].append(npz[k])
            samples = {k: np.stack(v) for k, v in rows.items()}
        samples['mfcc_features'] = samples['mfcc_features'].astype(np.float16)
        n = int(samples['audio'].shape[0])
        if part == 'train':
            self.train_samples, self.num_train_samples = samples, n
        elif part == 'val':
            self.val_samples, self.num_val_samples = samples, n
        else:
            self.test_samples, self.num_test_samples = samples, n
        self.__dict__.get('_feeders', {}).clear()         # feeders hold the previous arrays
        if part == 'train':                               # the touched-row capacity is derived from the training set
            self.gen_exchange, self._graphed = None, None
        return samples

    def yield_batch(self, train):
        """processor_v2.py:589-638: B indices drawn WITH replacement per pseudo pass; audio int16 * max / 32767;
        mfcc fp16 -> fp32; speaker ids drawn from the speakers NOT present in the batch (vectorised)."""
        samples = self.train_samples if train else self.val_samples
        num_data = self.num_train_samples if train else self.num_val_samples
        spk = self.train_speaker_model if train else self.val_speaker_model
        B = self.args.batch_size
        if torch.device(self.device).type == 'cuda' and getattr(
                self.args, 'prefetch_batches', config.get('PREFETCH')):
            # pinned staging + background gather + device-side decode, two batches ahead (data.BatchFeeder)
            from .data import BatchFeeder
            feeders = self.__dict__.setdefault('_feeders', {})
            if train not in feeders:
                feeders[train] = BatchFeeder(samples, num_data, B, self.device, spk)
            yield from feeders[train].batches((num_data + B - 1) // B)
            return
        from .data import host_batch
        for _ in range((num_data + B - 1) // B):
            text, vec, audio, mfcc, vids = host_batch(samples, num_data, B, spk)
            yield tuple(None if a is None else torch.from_numpy(a).to(self.device, non_blocking=True)
                        for a in (text, vec, audio, mfcc, vids))

    # ------------------------------------------------------------------------------------------------
    def synthesize_clip(self, seed_seq, clip_audio, sample_rate, clip_words, mfcc_windows=None, mfcc_fn=None,
                        speaker_vid_idx=0, unit_time=None):
        """The synthesis loop of ``render_clip`` (processor_v2.py:1173-1330; rendering, pkl export and the optional
        fade-out are outside the path): the clip is cut into windows of ``n_poses`` frames with a stride of
        ``n_poses - n_pre_poses``; per window the tri-modal baseline and the s2ag generator run at batch 1, the last
        ``n_pre_poses`` output frames seed the next window, and consecutive windows are cross-faded over those frames.

        ``clip_words``: [[word, start_s, end_s], ...] relative to the clip start; ``mfcc_windows[k]`` (or
        ``mfcc_fn(audio_window)``): the (num_mfcc, mfcc_length) image of window k -- MFCC extraction (librosa) is data
        preparation.  Everything between the one upload of the window inputs and the one download of the result stays
        on the device: seed hand-off and cross-fade included.  Returns ``(out_dir_vec_trimodal, out_dir_vec)`` as
        float32 numpy arrays of shape (W * (n_poses - n_pre_poses) + n_pre_poses, pose_dim)."""
        cfg = self.s2ag_config_args
        T, n_pre, dev = cfg.n_poses, cfg.n_pre_poses, self.device
        fps = cfg.motion_resampling_framerate
        clip_audio = np.asarray(clip_audio, dtype=np.float32)
        clip_length = len(clip_audio) / sample_rate
        if unit_time is None:
            unit_time = T / fps
        stride_time = (T - n_pre) / fps
        num = 1 if clip_length < unit_time else math.ceil((clip_length - unit_time) / stride_time) + 1
        alen = int(unit_time * sample_rate)
        texts, audios, mfccs = [], [], []
        for k in range(num):
            t0 = min(k * stride_time, clip_length)
            t1 = min(t0 + unit_time, clip_length)
            if t0 >= t1:
                continue
            a0 = math.floor(t0 / clip_length * len(clip_audio))
            win = clip_audio[a0:a0 + alen]
            if len(win) < alen:
                win = np.pad(win, (0, alen - len(win)), 'constant')
            ext = np.zeros(T, dtype=np.int64)              # PAD = 0 (utils/vocab.py:9)
            frame_duration = (t1 - t0) / T
            for word, w_s, w_e in clip_words:              # DataPreprocessor.get_words_in_time_range
                if w_s >= t1:
                    break
                if w_e <= t0:
                    continue
                ext[max(0, int(np.floor((w_s - t0) / frame_duration)))] = self.lang_model.get_word_index(word)
            texts.append(ext)
            audios.append(win)
            mfccs.append(np.asarray(mfcc_windows[len(mfccs)] if mfcc_windows is not None else mfcc_fn(win),
                                    dtype=np.float32))
        W = len(texts)
        text_d = torch.from_numpy(np.stack(texts)).to(dev)
        audio_d = torch.from_numpy(np.stack(audios)).to(dev)
        mfcc_d = torch.from_numpy(np.stack(mfccs)).to(dev)
        if cfg.z_type == 'speaker' and speaker_vid_idx is None:
            speaker_vid_idx = np.random.randint(0, self.s2ag_generator.z_obj.n_words)
        was_training = (self.trimodal_generator.training, self.s2ag_generator.training)
        self.trimodal_generator.eval()
        self.s2ag_generator.eval()
        # A training step leaves the generator in "share the pose/audio encoders across the passes of this step" mode,
        # keyed on buffer addresses: here every window reuses the same static buffers, so it must be off (each window
        # encodes its own seed poses and audio), and derived weight tensors of a step in flight are not to be reused.
        was_sharing = self.s2ag_generator.share_passes
        self.s2ag_generator.share_passes = None
        self.s2ag_generator._shared = None
        ops.begin_step()
        # One window = ~400 small launches at batch 1: the step runs on static buffers (inputs of the window, seed
        # poses, outputs) so that from the second window on it is ONE hipGraph replay.  The graph bakes in derived
        # weight tensors, so it is re-captured whenever a weight tensor changed (optimizer step, load_state_dict).
        params = list(self.trimodal_generator.parameters()) + list(self.s2ag_generator.parameters())
        key = (T, alen, tuple(mfcc_d.shape[1:]), cfg.z_type == 'speaker', sum(p._version for p in params),
               getattr(getattr(self, 'gen_arena', None), 'epoch', -1))
        st = self.__dict__.get('_synth')
        if st is None or st['key'] != key:
            st = dict(key=key, graph=None,
                      pre_t=torch.zeros(1, T, self.pose_dim + 1, device=dev),
                      pre_g=torch.zeros(1, T, self.pose_dim + 1, device=dev),
                      text=torch.zeros(1, T, dtype=torch.int64, device=dev),
                      audio=torch.zeros(1, alen, device=dev), mfcc=torch.zeros((1,) + tuple(mfcc_d.shape[1:]), device=dev),
                      vid=torch.zeros(1, dtype=torch.int64, device=dev) if cfg.z_type == 'speaker' else None,
                      out_t=torch.zeros(T, self.pose_dim, device=dev), out_g=torch.zeros(T, self.pose_dim, device=dev))
            self._synth = st
        pre_t, pre_g = st['pre_t'], st['pre_g']
        pre_t.zero_()
        pre_t[0, :n_pre, :-1] = torch.as_tensor(np.asarray(seed_seq)[:n_pre], dtype=torch.float32, device=dev)
        pre_t[0, :n_pre, -1] = 1                            # indicating bit for seed poses
        pre_g.copy_(pre_t)
        if st['vid'] is not None:
            st['vid'].fill_(int(speaker_vid_idx))

        if 'side' not in st:
            st['side'] = torch.cuda.Stream(device=dev)

        def window_step():
            # the two generators of a window are independent: the baseline runs on a forked stream; their noise
            # snapshots are drawn first, in the reference's call order (tri-modal, then s2ag), on this stream
            nz_t, nz_g = noise.begin_pass(dev), noise.begin_pass(dev)
            cur, side = torch.cuda.current_stream(), st['side']
            side.wait_stream(cur)
            with torch.cuda.stream(side), noise.use_pass(nz_t), ops.sequential_branches():
                out_t = self.trimodal_generator(pre_t, st['text'], st['audio'], st['vid'])[0]
                st['out_t'].copy_(out_t[0])
                pre_t[0, :n_pre, :-1] = out_t[0, -n_pre:]   # the seed poses of the next window
            with noise.use_pass(nz_g):
                out_g = self.s2ag_generator(pre_g, st['text'], st['mfcc'], st['vid'])[0]
            st['out_g'].copy_(out_g[0])
            pre_g[0, :n_pre, :-1] = out_g[0, -n_pre:]
            cur.wait_stream(side)

        use_graph = getattr(self, 'use_hip_graph', True) and config.get('SYNTH_GRAPH')
        outs_t = torch.empty(W, T, self.pose_dim, device=dev)
        outs_g = torch.empty(W, T, self.pose_dim, device=dev)
        with torch.no_grad():
            for k in range(W):
                st['text'].copy_(text_d[k:k + 1])
                st['audio'].copy_(audio_d[k:k + 1])
                st['mfcc'].copy_(mfcc_d[k:k + 1])
                if k == 0 or not use_graph:
                    window_step()                           # eager: also warms up everything the capture needs
                else:
                    if st['graph'] is None:
                        torch.cuda.synchronize()
                        gr = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gr, capture_error_mode='thread_local'):     # nothing executes during capture
                            window_step()
                        st['graph'] = gr
                    st['graph'].replay()
                outs_t[k].copy_(st['out_t'])
                outs_g[k].copy_(st['out_g'])
            res = []
            # smoothing motion transition (:1296-1322): window k-1 gives up its last n_pre frames, blended into the
            # first n_pre of window k with weights (n - j) / (n + 1) and (j + 1) / (n + 1)
            j = torch.arange(n_pre, device=dev, dtype=torch.float32).unsqueeze(1)
            w_prev, w_next = (n_pre - j) / (n_pre + 1), (j + 1) / (n_pre + 1)
            for o in (outs_t, outs_g):                      # (W, T, pose_dim)
                if W > 1:
                    o[1:, :n_pre] = o[:-1, -n_pre:] * w_prev + o[1:, :n_pre] * w_next
                res.append(torch.cat([o[:-1, :T - n_pre].reshape(-1, o.shape[-1]), o[-1]]).cpu().numpy())
        self.trimodal_generator.train(was_training[0])
        self.s2ag_generator.train(was_training[1])
        self.s2ag_generator.share_passes = was_sharing
        return res[0], res[1]

    # ------------------------------------------------------------------------------------------------
    def _use_gan(self):
        return self.meta_info['epoch'] > self.s2ag_config_args.loss_warmup and \
            self.s2ag_config_args.loss_gan_weight > 0.0

    def _gen_passes(self):
        """Forward passes of the trainable generator per step (processor_v2.py:798, :823, :909): for D (GAN phase only), for
        the loss, with shuffled speakers (regulariser branch only) -- what the shared encoders' BatchNorm statistics advance by."""
        return 1 + int(self._use_gan()) + int(self.use_div_reg)

    def _make_pre_seq(self, target_poses):
        n_pre = self.s2ag_config_args.n_pre_poses
        return ops.make_pre_seq(target_poses, n_pre)

    def _fork(self, idx):
        """side stream ``idx`` picks up after everything queued on the current stream"""
        s = self._side[idx]
        s.wait_stream(torch.cuda.current_stream())
        ops.mark_side_stream(s)
        return s

    def _dis_phase(self, in_text, in_mfcc, target_poses, vid_indices, pre_seq, train, in_audio=None, cut=False):
        """processor_v2.py:792-814 up to (and including) dis_error.backward().

        Noise snapshots are drawn in the reference's pass order (G, D(real), D(fake)) on the main stream; the generator
        forward then runs on a forked stream beside D(real).  D(fake) starts after both, so D's BatchNorm running
        statistics are still updated real-then-fake."""
        ops.set_main_stream()
        self._early_rand = self._early_main = self._early_tri = None
        ops.stamp('D:start')
        self.s2ag_dis_optimizer.zero_grad()
        dev = pre_seq.device
        # the three passes of this phase + the generator-phase passes that run early (tri-modal 4th, loss pass 5th, shuffled
        # speakers 7th of the step): one launch
        nz_g, nz_real, nz_fake, d_tri, d_main, d_rand = noise.begin_passes(dev, 3, derived=(3, 4, PASSES_PER_STEP - 1))
        cur = torch.cuda.current_stream()
        if self.overlap_passes:
            # The generator forward is the long pole of this phase (D(fake) needs its output): it stays on the main
            # stream and is issued FIRST; D(real) is issued after it on a stream forked from the phase start.
            side = self._fork(0)
            hoist = bool(train and self.early_main and self.s2ag_generator.share_passes and in_audio is not None)
            lockstep = hoist and self.early_main == 3 and self.early_rand and \
                hasattr(self.s2ag_generator, 'forward_passes')
            if self.s2ag_generator.share_passes and self.encoders_aside:
                # the generator's shared pose/audio encoders run on a forked stream beside its text encoder
                enc = self._fork(1)
                # the audio encoder ahead of D(real) on ITS stream (a fifth captured stream made graph replays crash now and
                # then; D(real)'s backward still ends well before D(fake)'s forward does)
                enc_audio = side if (self.encoders_apart and lockstep) else None
                with torch.cuda.stream(enc), ops.sequential_branches():
                    self.s2ag_generator.prepare_shared(pre_seq, in_mfcc, audio_stream=enc_audio)

            def early_rand_pass(side1):
                # The generator's third forward (shuffled speakers, no_grad) depends on nothing the D step changes, and
                # with the pose/audio encoders shared it contains no BatchNorm (no ordering of running statistics): it
                # runs beside the D step (small kernels that leave most CUs idle) instead of on the critical path of the
                # generator phase.  Its noise is that of pass 7 of the step, as upstream.
                if not (self.s2ag_generator.share_passes and self.early_rand):
                    return
                with torch.cuda.stream(side1):
                    nz_rand = d_rand
                with torch.cuda.stream(side1), torch.no_grad(), noise.use_pass(nz_rand), ops.sequential_branches():
                    rand_vids = vid_indices[torch.randperm(vid_indices.shape[0], device=vid_indices.device)]
                    ops.stamp('D:G(rand) begin [side1]')
                    out_rand, z_rand, _, _ = self.s2ag_generator(pre_seq, in_text, in_mfcc, rand_vids)
                    ops.stamp('D:G(rand) end [side1]')
                out_rand.record_stream(cur)
                z_rand.record_stream(cur)
                self._early_rand = (out_rand, z_rand)

            if lockstep:
                # All three generator passes of the step (for D, for the loss, with shuffled speakers) read the same
                # weights and inputs -- G's weights move only at the end of the step, and with the pose/audio encoders
                # shared the passes hold no BatchNorm -- and differ in noise only: they run here in LOCKSTEP, each
                # decoder layer of the three as one cooperative launch (PoseGenerator.forward_passes).  The loss pass
                # is the one with autograd; each pass keeps the noise snapshot of its place in the reference's order
                # (1, 5, 7).  The frozen baseline (pass 4) follows on a forked stream, beside D(fake) and D's backward.
                nz_main, nz_rand = d_main, d_rand
                rand_vids = vid_indices[torch.randperm(vid_indices.shape[0], device=vid_indices.device)]
                self.s2ag_generator.cut_backward = bool(cut)
                self.s2ag_generator._cut = None
                ops.stamp('D:G(x3) begin [main]')
                r_main, r_dis, r_rand = self.s2ag_generator.forward_passes(
                    pre_seq, in_text, in_mfcc, [(vid_indices, nz_main, True), (vid_indices, nz_g, False),
                                                (rand_vids, nz_rand, False)])
                ops.stamp('D:G(x3) end [main]')
                self._early_main = r_main
                self._early_rand = (r_rand[0], r_rand[1])
                out_dir_vec = r_dis[0]
                nz_tri = d_tri
                s_tri = self._fork(1)
                with torch.cuda.stream(s_tri), torch.no_grad(), noise.use_pass(nz_tri), ops.sequential_branches():
                    out_tri, *_ = self.trimodal_generator(pre_seq, in_text, in_audio, vid_indices)
                    ops.stamp('D:tri-modal end [side1]')
                out_tri.record_stream(cur)
                self._early_tri = out_tri
            elif hoist:
                # The generator's LOSS pass (pass 5 of the step, with autograd) and the frozen tri-modal baseline (pass
                # 4) depend on nothing the D step changes -- G's weights move only at the end of the step, and with the
                # pose/audio encoders shared the passes hold no BatchNorm.  The loss pass takes the main stream (its
                # backward nodes then run on the main stream of the generator phase, where weight gradients fork from);
                # the D step's own generator pass and the baseline run on forked streams beside it.  Each pass keeps
                # the noise snapshot of its place in the reference's pass order.
                nz_tri, nz_main = d_tri, d_main
                # four streams in all (the runtime maps streams onto four hardware queues; a fifth concurrent stream
                # shares a queue and serialises behind another pass): main = loss pass, side0 = D(real), side1 = shared
                # encoders then G(rand), side2 = G(dis) then the baseline
                tri_apart = self.early_main == 1     # 1: baseline on a stream of its own; 2: behind G(dis)
                if tri_apart:
                    s_tri = self._fork(3)
                    with torch.cuda.stream(s_tri), torch.no_grad(), noise.use_pass(nz_tri), ops.sequential_branches():
                        out_tri, *_ = self.trimodal_generator(pre_seq, in_text, in_audio, vid_indices)
                        ops.stamp('D:tri-modal end [side3]')
                s_gd = self._fork(2)
                g_dis_done = torch.cuda.Event()
                with torch.cuda.stream(s_gd), torch.no_grad(), ops.sequential_branches():
                    ops.stamp('D:G(dis) begin [side2]')
                    with noise.use_pass(nz_g):
                        out_dir_vec, *_ = self.s2ag_generator(pre_seq, in_text, in_mfcc, vid_indices)
                    ops.stamp('D:G(dis) end [side2]')
                    g_dis_done.record(s_gd)
                    if not tri_apart:
                        with noise.use_pass(nz_tri):
                            out_tri, *_ = self.trimodal_generator(pre_seq, in_text, in_audio, vid_indices)
                        ops.stamp('D:tri-modal end [side2]')
                out_dir_vec.record_stream(cur)
                out_tri.record_stream(cur)
                self._early_tri = out_tri
                rand_first = self.encoders_aside and self.early_main == 2
                if rand_first:
                    early_rand_pass(enc)                # behind the shared encoders, on their stream
                self.s2ag_generator.cut_backward = bool(cut)
                self.s2ag_generator._cut = None
                ops.stamp('D:G(main) begin [main]')
                with noise.use_pass(nz_main):
                    self._early_main = self.s2ag_generator(pre_seq, in_text, in_mfcc, vid_indices)
                ops.stamp('D:G(main) fwd end [main]')
            else:
                ops.stamp('D:G(dis) begin [main]')
                with torch.no_grad(), noise.use_pass(nz_g):    # upstream builds this graph and never uses it
                    out_dir_vec, *_ = self.s2ag_generator(pre_seq, in_text, in_mfcc, vid_indices)
                ops.stamp('D:G(dis) end [main]')
            l_real = real_bwd_done = None
            real_fwd_done = torch.cuda.Event()
            with torch.cuda.stream(side), noise.use_pass(nz_real), ops.sequential_branches():
                ops.stamp('D:D(real) begin [side]')
                dis_real = self.s2ag_discriminator(target_poses, in_text)
                ops.stamp('D:D(real) end [side]')
                real_fwd_done.record(side)
                if train and self.early_real_backward:
                    # The D loss is the sum of a real and a fake term (processor_v2.py:811): the real half is
                    # back-propagated HERE, on the forked stream, while the generator pass that produces the fake half
                    # is still running -- only the fake half is left for the critical path behind D(fake).
                    l_real = ops.dis_loss_half(dis_real, True)
                    with ops.local_backward():
                        ops.backward_from(l_real)
                    ops.stamp('D:D(real) backward end [side]')
                    real_bwd_done = torch.cuda.Event()
                    real_bwd_done.record(side)
            cur.wait_event(real_fwd_done)       # D(fake) follows D(real)'s FORWARD (BatchNorm running statistics order)
            if hoist and not lockstep:
                cur.wait_event(g_dis_done)
            if not lockstep and not (hoist and rand_first):
                early_rand_pass(self._fork(1))          # beside D(fake) and D's backward
        else:
            l_real = None
            with torch.no_grad(), noise.use_pass(nz_g):    # upstream builds this graph and never uses it
                out_dir_vec, *_ = self.s2ag_generator(pre_seq, in_text, in_mfcc, vid_indices)
            with noise.use_pass(nz_real):
                dis_real = self.s2ag_discriminator(target_poses, in_text)
        with noise.use_pass(nz_fake):
            dis_fake = self.s2ag_discriminator(out_dir_vec.detach(), in_text)
        if self.overlap_passes and l_real is not None:
            l_fake = ops.dis_loss_half(dis_fake, False)
            ops.stamp('D:D(fake) end, backward begins')
            # D(real)'s backward ends with the flush of D's derived-parameter stages (read-then-clear, not atomic):
            # the fake half accumulates into the same stages, so it must start after that flush.  D(real)'s backward
            # (~1 ms) began while the generator pass was still running: the wait is normally already satisfied.
            cur.wait_event(real_bwd_done)
            ops.backward_from(l_fake)
            dis_error = l_fake.detach()
            ops.join_side_streams()
            dis_error = dis_error + l_real.detach()
        else:
            dis_error = ops.dis_loss(dis_real, dis_fake)
            ops.stamp('D:D(fake) end, backward begins')
            if train:
                ops.backward_from(dis_error)
            ops.join_side_streams()      # backward kernels ran on the forked streams too
        ops.stamp('D:end')
        return dis_error.detach()

    def _gen_phase(self, in_text, in_audio, in_mfcc, target_poses, vid_indices, pre_seq, train, cut=False):
        """processor_v2.py:816-941 up to (and including) loss.backward().  ``cut`` (data parallel): backward stops at the
        recurrent decoder's input -- GRU and out gradients are complete, ``_gen_backward_rest`` does the encoders."""
        cfg = self.s2ag_config_args
        if not self.use_mfcc:                 # the audio-ablation generator reads the raw waveform (processor_v2.py:794-797)
            in_mfcc = in_audio
        early_main = getattr(self, '_early_main', None)   # the loss pass / the baseline already ran beside the D step
        early_tri = getattr(self, '_early_tri', None)
        self._early_main = self._early_tri = None
        if early_main is None:
            self.s2ag_generator.cut_backward = bool(cut and train)
            self.s2ag_generator._cut = None
        ops.set_main_stream()
        ops.stamp('G:start')
        self.s2ag_gen_optimizer.zero_grad()
        dev = pre_seq.device
        # pass order of the reference: tri-modal baseline, G(main), D(gen), G(rand)
        nz_tri, nz_main, nz_dgen, nz_rand = noise.begin_passes(dev, 4)
        early = getattr(self, '_early_rand', None)        # G(rand) already ran beside the D step (see _dis_phase)
        self._early_rand = None
        with_rand = self.use_div_reg                       # processor_v2.py:899-900; False: the branch of :933-934
        if early is None and with_rand:
            rand_idx = torch.randperm(vid_indices.shape[0], device=vid_indices.device)
            rand_vids = vid_indices[rand_idx]
        cur = torch.cuda.current_stream()
        if early_tri is not None:
            out_tri = early_tri
        elif self.overlap_passes:    # the frozen baseline shares nothing with G/D: run it beside the main forward
            side0 = self._fork(0)
            with torch.cuda.stream(side0), torch.no_grad(), noise.use_pass(nz_tri), ops.sequential_branches():
                ops.stamp('G:tri-modal begin [side0]')
                out_tri, *_ = self.trimodal_generator(pre_seq, in_text, in_audio, vid_indices)
                ops.stamp('G:tri-modal end [side0]')
        else:
            with torch.no_grad(), noise.use_pass(nz_tri):
                out_tri, *_ = self.trimodal_generator(pre_seq, in_text, in_audio, vid_indices)
        if early_main is not None:
            out, z, z_mu, z_log_var = early_main
        else:
            ops.stamp('G:G(main) begin [main]')
            with noise.use_pass(nz_main):
                out, z, z_mu, z_log_var = self.s2ag_generator(pre_seq, in_text, in_mfcc, vid_indices)
            ops.stamp('G:G(main) fwd end [main]')
        if early is not None:
            out_rand, z_rand = early
        elif not with_rand:
            out_rand = z_rand = None
        elif self.overlap_passes:    # G(rand) follows G(main) (BatchNorm running stats order) but runs beside D(gen)
            side1 = self._fork(1)
            with torch.cuda.stream(side1), torch.no_grad(), noise.use_pass(nz_rand), ops.sequential_branches():
                ops.stamp('G:G(rand) begin [side1]')
                out_rand, z_rand, _, _ = self.s2ag_generator(pre_seq, in_text, in_mfcc, rand_vids)
                ops.stamp('G:G(rand) end [side1]')
        # upstream lets loss.backward() also fill D's .grad, which the next D step zeroes unread;
        # skipping those weight-gradient kernels changes nothing observable
        flags = [p.requires_grad for p in self.dis_arena.params]
        for p in self.dis_arena.params:
            p.requires_grad_(False)
        try:
            with noise.use_pass(nz_dgen):
                dis_output = self.s2ag_discriminator(out, in_text)
        finally:
            for p, f in zip(self.dis_arena.params, flags):
                p.requires_grad_(f)
        if self.overlap_passes:
            if early_tri is None:
                cur.wait_stream(side0)
            if early is None and with_rand:
                cur.wait_stream(side1)
        elif early is None and with_rand:
            with torch.no_grad(), noise.use_pass(nz_rand):
                out_rand, z_rand, _, _ = self.s2ag_generator(pre_seq, in_text, in_mfcc, rand_vids)
        self._last_outs = (out_tri.detach(), out.detach())      # forward_pass_s2ag(calculate_metrics=True) reads them
        w_gan = cfg.loss_gan_weight if self.meta_info['epoch'] > cfg.loss_warmup else 0.0
        # the loss weights were snapshotted at construction together with the branch (self._loss_weights): the captured
        # replay and the eager step cannot drift apart when cfg is edited afterwards.  Without the regulariser the fused
        # loss gets out_rand = None: regression (+ GAN) term alone, divergence / KLD not evaluated (processor_v2.py:933-934)
        w_regr, w_div, w_kld = self._loss_weights
        weights = (w_regr, w_gan, w_div, w_kld) if with_rand else (w_regr, w_gan, 0.0, 0.0)
        total, comps = ops.gen_loss(out, dis_output, z_mu, z_log_var, target_poses, out_tri, out_rand if with_rand else None,
                                    z, z_rand if with_rand else None, weights)
        ops.stamp('G:losses done, backward begins')
        if self.overlap_passes and self.encoders_aside and self.s2ag_generator.share_passes and self._use_gan():
            ops.mark_side_stream(self._side[1])      # the shared encoders' backward runs on the stream of their forward
            if self.encoders_apart:
                ops.mark_side_stream(self._side[0])
        if train:
            ops.backward_from(total)
        ops.join_side_streams()
        ops.stamp('G:end' if not self.s2ag_generator.cut_backward else 'G:decoder backward done (bucket A complete)')
        return comps

    def _gen_backward_rest(self, in_text, ex):
        """Second half of the generator's backward pass in data-parallel runs: from the decoder's input through the
        encoders (bucket A is already on the wire), then the touched-row records of the embedding gradient."""
        G = self.s2ag_generator
        fulls, leaves = G._cut
        G._cut, G.cut_backward = None, False
        pairs = [(f, l.grad) for f, l in zip(fulls, leaves) if l.grad is not None]
        ops.set_main_stream()
        if self.overlap_passes and self.encoders_aside and G.share_passes and self._use_gan():
            ops.mark_side_stream(self._side[1])
            if self.encoders_apart:
                ops.mark_side_stream(self._side[0])
        torch.autograd.backward([f for f, _ in pairs], [g for _, g in pairs])
        ops.join_side_streams()
        ex.pack_rows(in_text)
        ops.stamp('G:end')

    def _finish(self, comps, dis_error):
        """ONE device->host read per step (upstream: 5-7 .item() syncs, processor_v2.py:943-956)."""
        cfg = self.s2ag_config_args
        flag = ops.coop_error_flag(comps.device)
        host = torch.cat((comps, dis_error.reshape(1) if dis_error is not None else comps.new_zeros(1), flag)).tolist()
        total, huber, gen_error, div_reg, kld, l1, l1_tri, _, dis, timed_out = host
        ops.check_coop_flag(timed_out)       # a cooperative recurrence that lost a peer continued with wrong values
        w_regr, w_div, w_kld = self._loss_weights
        d = {'loss': w_regr * huber, 'total': total}
        if self.use_div_reg:                  # (the reference's loss_dict has these entries in that branch only, :943-947)
            d['KLD'] = w_kld * kld
            d['DIV_REG'] = w_div * div_reg
        if self._use_gan():
            d['gen'] = cfg.loss_gan_weight * gen_error
            d['dis'] = dis
        self.last_losses = d
        return l1 - l1_tri

    @staticmethod
    def push_samples(evaluator, target, out_dir_vec, in_text_padded, in_audio, losses_all, joint_mae, accel, mean_dir_vec,
                     n_poses, n_pre_poses):
        """processor_v2.py:738-774: L1 of the direction vectors, MAE of the joint coordinates (convert_dir_vec_to_pose on
        dir + mean_dir_vec) behind the seed poses, acceleration difference; latent features into ``evaluator`` (FGD).
        Everything is computed on the device (ops.pose_metrics: one launch, float64 like numpy upstream); ONE 24-byte
        read-back feeds the three meters (upstream: two full tensor downloads + one .item())."""
        batch_size = len(target)
        if out_dir_vec.shape[1] != n_poses:
            raise NotImplementedError('push_samples: generated sequences shorter than n_poses (the seed-less branch of '
                                      'processor_v2.py:763) are not produced on this path')
        l1, mae, acc = ops.pose_metrics(out_dir_vec.detach(), target.detach(), mean_dir_vec, n_pre_poses).tolist()
        losses_all.update(l1, batch_size)
        if evaluator:
            evaluator.push_samples(in_text_padded, in_audio, out_dir_vec, target)
        joint_mae.update(mae, batch_size)
        accel.update(acc, batch_size)
        return evaluator, losses_all, joint_mae, accel

    def forward_pass_s2ag(self, in_text, in_audio, in_mfcc, target_poses, vid_indices, train, target_seq=None,
                          words=None, aux_info=None, save_path=None, make_video=False, calculate_metrics=False,
                          losses_all_trimodal=None, joint_mae_trimodal=None, accel_trimodal=None, losses_all=None,
                          joint_mae=None, accel=None):
        """One GAN step (processor_v2.py:776-957).  Returns the reference's 7-tuple; the loss components of the
        step are left in ``self.last_losses``.  ``calculate_metrics`` (:866-890) pushes the step's two generated
        sequences (tri-modal baseline, s2ag generator) against ``target_seq`` into the six meters and the two FGD
        evaluators (``self.evaluator_trimodal`` / ``self.evaluator``: EmbeddingSpaceEvaluator or None).  ``make_video``
        belongs to the rendering path (matplotlib / ffmpeg) that is out of scope here."""
        if make_video:
            raise NotImplementedError('rendering is outside the MI355X hot path')
        if calculate_metrics:
            for name, v in (('target_seq', target_seq), ('losses_all_trimodal', losses_all_trimodal),
                            ('joint_mae_trimodal', joint_mae_trimodal), ('accel_trimodal', accel_trimodal),
                            ('losses_all', losses_all), ('joint_mae', joint_mae), ('accel', accel)):
                assert v is not None, '{} cannot be None when calculate_metrics is True'.format(name)
        ops.begin_step()
        # encoder sharing is scoped to THIS step (the cache is keyed on buffer addresses): off again when the step ends
        self.s2ag_generator.share_passes = self._gen_passes() if self.share_encoders else None
        try:
            ret = self._step(in_text, in_audio, in_mfcc, target_poses, vid_indices, train)
            if calculate_metrics:
                cfg = self.s2ag_config_args
                out_tri, out = self._last_outs
                self.evaluator_trimodal, losses_all_trimodal, joint_mae_trimodal, accel_trimodal = Processor.push_samples(
                    getattr(self, 'evaluator_trimodal', None), target_seq, out_tri, in_text, in_audio, losses_all_trimodal,
                    joint_mae_trimodal, accel_trimodal, cfg.mean_dir_vec, cfg.n_poses, cfg.n_pre_poses)
                self.evaluator, losses_all, joint_mae, accel = Processor.push_samples(
                    getattr(self, 'evaluator', None), target_seq, out, in_text, in_audio, losses_all, joint_mae, accel,
                    cfg.mean_dir_vec, cfg.n_poses, cfg.n_pre_poses)
            return ret + (losses_all_trimodal, joint_mae_trimodal, accel_trimodal, losses_all, joint_mae, accel)
        finally:
            self._last_outs = None
            self.s2ag_generator.share_passes = None
            self.s2ag_generator._shared = None

    def _step(self, in_text, in_audio, in_mfcc, target_poses, vid_indices, train):
        pre_seq = self._make_pre_seq(target_poses)
        dis_error = None
        ex = self._exchange() if train else None
        if ex is not None:
            ex.precheck(in_text)        # distinct word ids of the batch, MAX over ranks on its way to the host
        if self._use_gan():
            dis_error = self._dis_phase(in_text, in_mfcc if self.use_mfcc else in_audio, target_poses, vid_indices,
                                        pre_seq, train, in_audio=in_audio, cut=ex is not None)
            if train:
                self.dp.all_reduce_grads(self.dis_arena)
                self._sync_error_flag()
                self.s2ag_dis_optimizer.step(self.dp.grad_scale)
        comps = self._gen_phase(in_text, in_audio, in_mfcc, target_poses, vid_indices, pre_seq, train, cut=ex is not None)
        if train:
            if ex is not None:
                ex.launch_a()                              # GRU + out gradients travel ...
                self._gen_backward_rest(in_text, ex)       # ... while the encoders are back-propagated
                ex.exchange_rest()
                self._sync_error_flag()
                ex.merge_rows()
            self.s2ag_gen_optimizer.step(self.dp.grad_scale)
        return (self._finish(comps, dis_error),)

    # ---- hipGraph replay of the training step (static shapes) ---------------------------------------------
    def _build_graphed(self, in_text, in_audio, in_mfcc, target_poses, vid_indices):
        st = dict(text=in_text.clone(), audio=in_audio.clone(), mfcc=in_mfcc.clone(), target=target_poses.clone(),
                  vid=vid_indices.clone())
        out = {}
        use_gan = self._use_gan()
        ex = self._exchange()

        def seg_dis():
            ops.begin_step()
            self.s2ag_generator.share_passes = self._gen_passes() if self.share_encoders else None
            out['pre'] = self._make_pre_seq(st['target'])
            out['dis'] = self._dis_phase(st['text'], st['mfcc'] if self.use_mfcc else st['audio'], st['target'],
                                         st['vid'], out['pre'], True, in_audio=st['audio'],
                                         cut=ex is not None) if use_gan else None

        def seg_gen():
            if use_gan:
                self.s2ag_dis_optimizer.step(self.dp.grad_scale)
            out['comps'] = self._gen_phase(st['text'], st['audio'], st['mfcc'], st['target'], st['vid'], out['pre'],
                                           True, cut=ex is not None)

        def seg_gen_rest():
            self._gen_backward_rest(st['text'], ex)

        def seg_opt():
            if ex is not None:
                ex.merge_rows()
            self.s2ag_gen_optimizer.step(self.dp.grad_scale)
            ops.stamp('step end (G-Adam done)')
            self.s2ag_generator.share_passes = None        # sharing is scoped to the step (see forward_pass_s2ag)
            self.s2ag_generator._shared = None

        # Single process: three segments, nothing between them.  Data parallel: the collectives run from the host
        # between the segments -- the graphs never contain RCCL nodes -- and bucket A's all-reduce runs BESIDE the
        # segment that back-propagates the encoders.
        if ex is None:
            fns, between = [seg_dis, seg_gen, seg_opt], [None, None, None]
        else:
            fns = [seg_dis, seg_gen, seg_gen_rest, seg_opt]
            def after_dis():
                self.dp.all_reduce_grads(self.dis_arena)
                self._sync_error_flag()              # before D's Adam (first kernel of the next segment)

            def after_gen_rest():
                ex.exchange_rest()
                self._sync_error_flag()              # before G's Adam
            between = [after_dis if use_gan else None, ex.launch_a, after_gen_rest, None]
        segs = _GraphSegments(fns, between, before=(lambda: ex.precheck(st['text'])) if ex is not None else None)
        self._graphed = dict(st=st, out=out, segs=segs, key=self._graph_key(in_text, in_audio, in_mfcc, target_poses))

    def _graph_key(self, in_text, in_audio, in_mfcc, target_poses):
        # a captured graph holds the kernels of ONE precision mode / split-piece count: switching either re-captures
        return (tuple(in_text.shape), tuple(in_audio.shape), tuple(in_mfcc.shape), tuple(target_poses.shape),
                self._use_gan(), self.meta_info['epoch'] > self.s2ag_config_args.loss_warmup,
                self.use_div_reg, self._loss_weights, float(self.s2ag_config_args.loss_gan_weight),
                bf16.enabled(), bf16.step_mode(), ops._lib().s2ag_gru_coop_split_pieces())

    def train_step(self, in_text, in_audio, in_mfcc, target_poses, vid_indices, sync=True):
        """The training branch of forward_pass_s2ag, replayed from HIP graphs when shapes are static.
        NOTE: capturing runs three un-timed warm-up steps (they DO update the weights)."""
        if not self.use_hip_graph:
            return self.forward_pass_s2ag(in_text, in_audio, in_mfcc, target_poses, vid_indices, True)[0]
        key = self._graph_key(in_text, in_audio, in_mfcc, target_poses)
        if self._graphed is None or self._graphed['key'] != key:
            self._build_graphed(in_text, in_audio, in_mfcc, target_poses, vid_indices)
        g = self._graphed
        for name, t in (('text', in_text), ('audio', in_audio), ('mfcc', in_mfcc), ('target', target_poses),
                        ('vid', vid_indices)):
            g['st'][name].copy_(t, non_blocking=True)
        g['segs'].replay()
        # the Python side effects of a step do not replay with the graph: the arenas changed (tensors derived from the
        # weights -- weight-normed / folded / tap-major / split planes, the synthesis graph -- are stale) and a new
        # step generation begins
        self.gen_arena.epoch += 1
        self.dis_arena.epoch += 1
        ops.begin_step()
        if not sync:
            self._async_flag_check()
            return None
        return self._finish(g['out']['comps'], g['out']['dis'])

    def _async_flag_check(self, every=16):
        """sync=False steps never read anything back: every ``every`` steps the sticky cooperative-GRU time-out word is
        copied to pinned host memory WITHOUT waiting, and the copy issued ``every`` steps earlier is inspected."""
        st = self.__dict__.setdefault('_flag_poll', dict(n=0, host=None, ev=None))
        st['n'] += 1
        if st['n'] % every:
            return
        if st['host'] is None:
            st['host'] = torch.zeros(1, dtype=torch.float32).pin_memory()
        elif st['ev'].query():
            ops.check_coop_flag(st['host'][0].item())
        st['host'].copy_(ops.coop_error_flag(self.device), non_blocking=True)
        st['ev'] = torch.cuda.Event()
        st['ev'].record()

    # ------------------------------------------------------------------------------------------------
    def per_train_epoch(self):
        self.s2ag_generator.train()
        self.s2ag_discriminator.train()
        batch_s2ag_loss = 0.
        num_batches = self.num_train_samples // self.args.batch_size + 1
        self.meta_info['iter'] = 0
        log_interval = getattr(self.args, 'log_interval', 200)
        for extended_word_seq, vec_seq, audio, mfcc_features, vid_indices in self.yield_batch(train=True):
            loss = self.train_step(extended_word_seq, audio, mfcc_features, vec_seq, vid_indices)
            batch_s2ag_loss += loss
            self.iter_info['s2ag_loss'] = loss
            self.iter_info['lr_gen'] = '{}'.format(self.lr_s2ag_gen)
            self.iter_info['lr_dis'] = '{}'.format(self.lr_s2ag_dis)
            if self.meta_info['iter'] % log_interval == 0:
                self.io.print_log('\tIter {} Done.'.format(self.meta_info['iter']) + ''.join(
                    ' | {}: {:.4f}'.format(k, v) if isinstance(v, float) else ' | {}: {}'.format(k, v)
                    for k, v in self.iter_info.items()))
            self.meta_info['iter'] += 1
        batch_s2ag_loss /= num_batches
        self.epoch_info['mean_s2ag_loss'] = batch_s2ag_loss
        self.io.print_log('\tmean_s2ag_loss: {}. Best so far: {:.4f}.'.format(batch_s2ag_loss, self.best_s2ag_loss))

    def per_val_epoch(self):
        self.s2ag_generator.eval()
        self.s2ag_discriminator.eval()
        batch_s2ag_loss = 0.
        num_batches = self.num_val_samples // self.args.batch_size + 1
        self.meta_info['iter'] = 0
        for extended_word_seq, vec_seq, audio, mfcc_features, vid_indices in self.yield_batch(train=False):
            with torch.no_grad():
                loss, *_ = self.forward_pass_s2ag(extended_word_seq, audio, mfcc_features, vec_seq, vid_indices,
                                                  train=False)
            batch_s2ag_loss += loss
            self.meta_info['iter'] += 1
        batch_s2ag_loss /= num_batches
        self.epoch_info['mean_s2ag_loss'] = batch_s2ag_loss
        if batch_s2ag_loss < self.best_s2ag_loss and self.meta_info['epoch'] > self.min_train_epochs:
            self.best_s2ag_loss = batch_s2ag_loss
            self.best_s2ag_loss_epoch = self.meta_info['epoch']
            self.s2ag_loss_updated = True
        else:
            self.s2ag_loss_updated = False
        self.io.print_log('\tval mean_s2ag_loss: {}. Best so far: {:.4f}.'.format(batch_s2ag_loss,
                                                                                 self.best_s2ag_loss))

    def train(self):
        """processor_v2.py:1032-1069 (frozen tri-modal weights, resume protocol, save cadence)."""
        tri_path = jn(self.base_path, 'outputs', 'trimodal_gen.pth.tar')
        if os.path.exists(tri_path):
            ckpt = torch.load(tri_path, map_location=self.device)
            self.trimodal_generator.load_state_dict(ckpt['trimodal_gen_dict'])
        else:
            self.io.print_log('Warning! {} not found: tri-modal baseline keeps its random init.'.format(tri_path))
        if getattr(self.args, 's2ag_load_last_best', False):
            found = self.load_model_at_epoch(epoch=self.args.s2ag_start_epoch)
            if not found and self.args.s2ag_start_epoch != 'best':
                found = self.load_model_at_epoch(epoch='best')
                self.args.s2ag_start_epoch = self.best_s2ag_loss_epoch if found else 0
        else:
            self.args.s2ag_start_epoch = 0
        for epoch in range(self.args.s2ag_start_epoch, self.args.s2ag_num_epoch):
            self.meta_info['epoch'] = epoch
            self.io.print_log('s2ag training epoch: {}'.format(epoch))
            self.per_train_epoch()
            self.io.print_log('Done.')
            if (epoch % self.args.val_interval == 0) or (epoch + 1 == self.args.s2ag_num_epoch):
                self.io.print_log('s2ag val epoch: {}'.format(epoch))
                self.per_val_epoch()
                self.io.print_log('Done.')
            if self.s2ag_loss_updated or (epoch % self.args.save_interval == 0 and epoch > self.min_train_epochs):
                self.save_model(epoch, self.epoch_info['mean_s2ag_loss'])
