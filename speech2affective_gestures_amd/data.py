"""Host-to-HBM batch feeder for ``Processor.yield_batch`` (reference: processor_v2.py:589-638).

The reference draws B clip indices with replacement, fancy-indexes five numpy arrays, decodes the audio in float64 on
the host (``int16 * audio_max / 32767``), converts to float32 and copies each tensor synchronously -- ~10 ms of host
work and 20 MB of PCIe per batch of 128, i.e. as long as a whole training step takes on the MI355X.  Here

* a background thread draws the indices (the reference's ``np.random`` stream: batch keys through
  ``choice(n, B, True, p=uniform)``, then the "other speaker" ids -- one vectorised draw that consumes the stream exactly
  as the reference's B scalar draws do; pinned by tests/golden/batch_small.npz, recorded from the reference's own
  ``yield_batch``), gathers the RAW rows (int16 audio, float64 peaks / poses, float16 MFCCs, int64 words)
  straight into pinned staging buffers (``np.take(..., out=pinned)``),
* copies them on its own HIP stream and decodes on the device (``s2ag_audio_decode`` / ``s2ag_to_f32``: the reference's
  float64 arithmetic, bit-identical results, half the PCIe bytes for the audio),
* and hands (tensors, event) to the trainer through a bounded queue, ``depth`` batches ahead; the trainer's stream
  waits on the event, never on the host.
"""
import ctypes as C
import queue
import threading

import numpy as np
import torch

from . import _lib as L

_FIELDS = ('extended_word_seq', 'vec_seq', 'audio', 'audio_max', 'mfcc_features')


def draw_keys(num_data: int, size: int) -> np.ndarray:
    """The clip indices of one batch, drawn exactly as the reference does (processor_v2.py:598-601): WITH replacement and
    with an explicit uniform ``p`` -- passing ``p`` makes numpy take its cdf / searchsorted path (one uniform double per
    index), a different consumption of the RandomState stream than the ``p=None`` integer path, so under the same
    ``np.random.seed`` only this call reproduces the reference's batches."""
    prob_dist = np.ones(num_data) / float(num_data)
    return np.random.choice(num_data, size=size, replace=True, p=prob_dist)


def host_batch(samples, num_data: int, size: int, speaker_model=None):
    """One batch on the host, decoded with the reference's arithmetic (float64 ``int16 * max / 32767``, then float32):
    numpy arrays (text i64, vec f32, audio f32, mfcc f32, vids i64 | None).  The reference-shaped path of
    ``Processor.yield_batch`` (prefetch off) and the checker of the feeder's device-side decode."""
    keys = draw_keys(num_data, size)
    text = samples['extended_word_seq'][keys]
    vec = samples['vec_seq'][keys].astype(np.float32)
    audio = (samples['audio'][keys] * samples['audio_max'][keys, None] / 32767).astype(np.float32)
    mfcc = samples['mfcc_features'][keys].astype(np.float32)
    vids = None
    if speaker_model is not None and speaker_model.__class__.__name__ == 'Vocab':
        vids = other_speakers(speaker_model, samples['vid_indices'][keys], size).astype(np.int64)
    return text, vec, audio, mfcc, vids


def other_speakers(speaker_model, present: np.ndarray, size: int):
    """Speaker ids for the batch: drawn from the speakers NOT present in it (processor_v2.py:622-635; the reference
    recomputes the same set difference for every clip of the batch)."""
    others = np.setdiff1d(np.fromiter(speaker_model.word2index.values(), dtype=np.int64), present)
    return np.random.choice(others, size=size)


class BatchFeeder:
    """Iterates ``n_batches`` batches ``(text i64, vec f32, audio f32, mfcc f32, vids i64 | None)`` resident on ``device``."""

    def __init__(self, samples, num_data, batch_size, device, speaker_model=None, depth=2):
        self.samples, self.num_data, self.B = samples, int(num_data), int(batch_size)
        self.device = torch.device(device)
        self.spk = speaker_model if (speaker_model is not None and
                                     speaker_model.__class__.__name__ == 'Vocab') else None
        self.depth = max(1, int(depth))
        self.lib = L.load()
        self.stream = torch.cuda.Stream(device=self.device)
        self.src = {}
        for k in _FIELDS:
            a = np.ascontiguousarray(samples[k])
            self.src[k] = a if k != 'audio_max' else a.astype(np.float64, copy=False)
        assert self.src['audio'].dtype == np.int16, 'audio is stored as int16 (processor_v2.py:603)'
        self.src['vec_seq'] = self.src['vec_seq'].astype(np.float64, copy=False)
        mf = self.src['mfcc_features']
        self.mfcc_f16 = mf.dtype == np.float16
        if not self.mfcc_f16:
            self.src['mfcc_features'] = mf.astype(np.float64, copy=False)
        # pinned staging, one set per batch in flight (+1: the set being filled)
        self.slots = []
        for _ in range(self.depth + 1):
            slot = {}
            for k in _FIELDS:
                a = self.src[k]
                t = torch.empty((self.B,) + a.shape[1:], dtype=torch.from_numpy(a[:1]).dtype).pin_memory()
                slot[k] = (t, t.numpy())
            slot['vids'] = torch.empty(self.B, dtype=torch.int64).pin_memory()
            slot['free'] = torch.cuda.Event()          # recorded when the copies out of this slot are done
            self.slots.append(slot)

    # ------------------------------------------------------------------------------------------------
    def _stage(self, slot):
        """Host part of one batch (runs on the feeder thread): index draw + row gather into pinned memory."""
        keys = draw_keys(self.num_data, self.B)
        for k in _FIELDS:
            np.take(self.src[k], keys, axis=0, out=slot[k][1], mode='clip')
        has_vids = self.spk is not None
        if has_vids:
            slot['vids'].numpy()[:] = other_speakers(self.spk, self.samples['vid_indices'][keys], self.B)
        return has_vids

    def _upload(self, slot, has_vids):
        """Device part: async copies + decode kernels on the feeder's stream; returns (tensors, ready event)."""
        dev, lib = self.device, self.lib
        sp = C.c_void_p(self.stream.cuda_stream)
        with torch.cuda.stream(self.stream):
            raw = {k: slot[k][0].to(dev, non_blocking=True) for k in _FIELDS}
            vids = slot['vids'].to(dev, non_blocking=True) if has_vids else None
            slot['free'].record(self.stream)
            B = self.B
            audio = torch.empty(raw['audio'].shape, dtype=torch.float32, device=dev)
            L.check(lib.s2ag_audio_decode(C.c_void_p(raw['audio'].data_ptr()), C.c_void_p(raw['audio_max'].data_ptr()),
                                          C.c_void_p(audio.data_ptr()), B, raw['audio'].numel() // B, sp), 'audio_decode')
            vec = torch.empty(raw['vec_seq'].shape, dtype=torch.float32, device=dev)
            L.check(lib.s2ag_to_f32(C.c_void_p(raw['vec_seq'].data_ptr()), 0, C.c_void_p(vec.data_ptr()),
                                    raw['vec_seq'].numel(), sp), 'to_f32')
            mfcc = torch.empty(raw['mfcc_features'].shape, dtype=torch.float32, device=dev)
            L.check(lib.s2ag_to_f32(C.c_void_p(raw['mfcc_features'].data_ptr()), int(self.mfcc_f16),
                                    C.c_void_p(mfcc.data_ptr()), raw['mfcc_features'].numel(), sp), 'to_f32')
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return (raw['extended_word_seq'], vec, audio, mfcc, vids), ready

    def _run(self, n_batches, q, stop):
        try:
            torch.cuda.set_device(self.device)
            for i in range(n_batches):
                if stop.is_set():
                    break
                slot = self.slots[i % len(self.slots)]
                slot['free'].synchronize()             # the copies that last read this slot have finished
                has_vids = self._stage(slot)
                item = self._upload(slot, has_vids)
                while not stop.is_set():
                    try:
                        q.put(item, timeout=0.1)
                        break
                    except queue.Full:
                        pass
            q.put(None)
        except BaseException as e:                     # surface feeder errors in the consumer
            q.put(e)

    def batches(self, n_batches):
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        th = threading.Thread(target=self._run, args=(n_batches, q, stop), daemon=True, name='s2ag-feeder')
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                tensors, ready = item
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ready)
                for t in tensors:
                    if t is not None:
                        t.record_stream(cur)
                yield tensors
        finally:
            stop.set()
            th.join(timeout=5)
