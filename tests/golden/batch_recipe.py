"""Tiny synthetic sample dict for the ``yield_batch`` golden (numbers only; shared by the generator script that runs
the reference and by the tests that run the product)."""
import numpy as np

N_DATA, BATCH, T, POSE_DIM, AUDIO_LEN, NUM_MFCC, MFCC_LEN, N_SPK, SEED = 23, 8, 34, 27, 600, 37, 5, 12, 4242


def samples(n=N_DATA, seed=99):
    r = np.random.RandomState(seed)
    return dict(
        extended_word_seq=r.randint(0, 50, size=(n, T)).astype(np.int64),
        vec_seq=r.randn(n, T, POSE_DIM),                                        # float64, as the cache holds it
        audio=r.randint(-32767, 32768, size=(n, AUDIO_LEN)).astype(np.int16),
        audio_max=np.abs(r.randn(n)) * 0.3,
        mfcc_features=(r.randn(n, NUM_MFCC, MFCC_LEN) * 0.1).astype(np.float16),
        vid_indices=r.randint(0, N_SPK, size=n).astype(np.int64))


class Vocab:                       # duck-typed speaker model (utils/vocab.py): class name must be 'Vocab'
    def __init__(self, n):
        self.n_words = n
        self.word2index = {'v%d' % i: i for i in range(n)}
