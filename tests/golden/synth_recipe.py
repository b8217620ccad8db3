"""Inputs of the sliding-window synthesis golden (pure numpy draws from a frozen legacy RandomState -- restates nothing
of the reference): a 6.0 s clip = 3 windows of 34 frames at 15 fps (stride 30 frames), the last one zero-padded."""
import numpy as np

HIDDEN, N_WORDS, N_SPK, SEED0 = 32, 64, 12, 8100
SR, FPS, SPEAKER = 16000, 15, 3
VOCAB = ['w%d' % i for i in range(7)]          # indexed 4..10 behind PAD/SOS/EOS/UNK (utils/vocab.py:9-12)


def clip_fixture():
    rs = np.random.RandomState(SEED0)
    audio = (rs.randn(6 * SR) * 0.05).astype(np.float32)
    words = [['w%d' % (i % 7), 0.3 + 0.6 * i, 0.3 + 0.6 * i + 0.4] for i in range(9)]
    mfcc = (rs.randn(3, 37, 71) * 0.1).astype(np.float32)
    poses = (rs.randn(90, 10, 3) * 0.3).astype(np.float64)              # 6 s at 15 fps, 10 joints
    eps = rs.randn(6, 1, 16).astype(np.float32)                          # (window, generator) in call order
    return audio, words, mfcc, poses, eps
