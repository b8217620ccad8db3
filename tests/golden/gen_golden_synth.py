"""Golden fixture for the sliding-window synthesis loop: runs the REFERENCE's ``Processor.render_clip``
(processor_v2.py:1144-1440) on CPU with recipe weights and records the concatenated, cross-faded direction vectors
of both generators.  Build container only (needs /root/reference); see gen_golden.py for the import recipe.

Pinned / stubbed, nothing else: ``cmn.get_mfcc_features`` (librosa) returns the fixture MFCC of the window it is
called for; ``en.re_parametrize`` consumes fixture eps in call order (tri-modal, then s2ag, per window);
``convert_dir_vec_to_pose`` records its argument (= out_dir_vec + mean_dir_vec) -- render_clip returns poses only.

    python tests/golden/gen_golden_synth.py          # rewrites tests/golden/synth_small.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (sets up the stubs and imports the reference)

torch, P, O, en = gg.torch, gg.P, gg.O, gg.en

from synth_recipe import FPS, HIDDEN, N_SPK, N_WORDS, SEED0, SPEAKER, SR, VOCAB, clip_fixture  # noqa: E402


def main():
    cfg, mods = gg.build(HIDDEN, N_WORDS, N_SPK, 0.0, SEED0)
    for m in mods.values():
        m.eval()
    cfg.mean_dir_vec = np.zeros((1, 27)).tolist()
    cfg.motion_resampling_framerate = FPS
    cfg.z_type = 'speaker'
    audio, words, mfcc, poses, eps = clip_fixture()
    lang = P.Vocab('words') if hasattr(P, 'Vocab') else gg.Vocab('words')
    for w in VOCAB:
        lang.index_word(w)

    pr = object.__new__(P.Processor)
    pr.s2ag_config_args = cfg
    pr.pose_dim = 27
    pr.device = torch.device('cpu')
    pr.lang_model = lang
    pr.args = gg.types.SimpleNamespace(train_s2ag=True, video_save_path='/tmp')
    pr.data_loader = {'train_data_s2ag': gg.types.SimpleNamespace(num_mfcc=37)}
    pr.s2ag_generator, pr.trimodal_generator = mods['G'], mods['T3']
    pr.best_s2ag_loss_epoch = 0

    calls = {'mfcc': 0}

    def fake_mfcc(audio_window, sr=SR, num_mfcc=37):
        k = calls['mfcc']
        calls['mfcc'] += 1
        return mfcc[k]
    P.cmn.get_mfcc_features = fake_mfcc
    gg.pin_eps([torch.from_numpy(e) for e in eps])
    rec = []
    real_conv = P.convert_dir_vec_to_pose

    def rec_conv(v):
        rec.append(np.array(v, dtype=np.float64).copy())
        return real_conv(v)
    P.convert_dir_vec_to_pose = rec_conv

    # a target clip is needed for the seed poses (first n_pre frames of its direction vectors)
    target_dir_vec = P.convert_pose_seq_to_dir_vec(P.resample_pose_seq(poses, 6.0, FPS)).reshape(-1, 27)
    with torch.no_grad():
        P.Processor.render_clip(pr, {'clip_duration_range': [1, 100], 'audio_sr': SR}, 'vid', 0, 1, poses, audio, SR,
                                [list(w) for w in words], [0.0, 6.0], test_samples=['vid_0.00_6.00'], speaker_vid_idx=SPEAKER,
                                check_duration=False, fade_out=False, make_video=False, save_pkl=False)
    assert calls['mfcc'] == 3 and len(rec) == 2, (calls, len(rec))
    # inputs are NOT stored: tests regenerate them from synth_recipe.clip_fixture()
    out = dict(seed_seq=target_dir_vec[:4].astype(np.float32), out_trimodal=rec[0].astype(np.float32),
               out_s2ag=rec[1].astype(np.float32))
    np.savez_compressed(os.path.join(HERE, 'synth_small.npz'), **out)
    print('synth_small.npz:', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
