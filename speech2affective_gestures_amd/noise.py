"""Device-resident random state.

``rng`` is a 2-word device array {seed, pass counter} (see include/s2ag_hip.h).  Every top-level forward
pass takes a *snapshot* of it and bumps the counter (one one-thread kernel for both), so
  * consecutive passes (the three generator forwards of one GAN step) see different noise,
  * the backward pass of a forward regenerates exactly that forward's masks from its snapshot,
  * a captured hipGraph keeps advancing the noise on every replay (the counter lives on the device).
Random *sites* (one per dropout / noise call site) are small integers handed out at module construction.
"""
import ctypes as C
import threading
from contextlib import contextmanager

import torch

from . import _lib as L

_state = {}
_site_counter = [0]
_tls = threading.local()


def new_site(n: int = 1) -> int:
    s = _site_counter[0]
    _site_counter[0] += n
    return s


def reset_sites(start: int = 0) -> None:
    """Restart site numbering (tests: two identically constructed models then draw identical noise)."""
    _site_counter[0] = start


def _dev_state(device) -> torch.Tensor:
    device = L.require_gpu_device(device, 'S2AG noise state')
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _state:
        _state[idx] = torch.tensor([0x5EED5EED, 0], dtype=torch.int64, device=f'cuda:{idx}')
    return _state[idx]


def manual_seed(seed: int, device='cuda') -> None:
    st = _dev_state(device)
    st.copy_(torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64))


def begin_pass(device) -> torch.Tensor:
    st = _dev_state(device)
    snap = torch.empty_like(st)
    L.check(L.load().s2ag_rng_snapshot(C.c_void_p(st.data_ptr()), C.c_void_p(snap.data_ptr()),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'rng_snapshot')
    return snap


def begin_passes(device, n: int, derived=()):
    """The snapshots of ``n`` consecutive passes (what n ``begin_pass`` calls return) plus, for every offset in
    ``derived``, the first snapshot with its pass counter advanced by that offset -- ONE launch.  Returns a list of
    n + len(derived) two-word tensors (rows of one buffer)."""
    st = _dev_state(device)
    out = torch.empty((n + len(derived), 2), dtype=st.dtype, device=st.device)
    offs = (C.c_int * max(1, len(derived)))(*[int(d) for d in derived])
    L.check(L.load().s2ag_rng_snapshots(C.c_void_p(st.data_ptr()), C.c_void_p(out.data_ptr()), int(n), offs,
                                        len(derived), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
            'rng_snapshots')
    return [out[i] for i in range(n + len(derived))]


@contextmanager
def noise_pass(device):
    """Scope of one forward pass: nested module forwards share the outermost pass's snapshot."""
    stack = getattr(_tls, 'stack', None)
    if stack is None:
        stack = _tls.stack = []
    if stack:
        yield stack[-1]
        return
    snap = begin_pass(device)
    stack.append(snap)
    try:
        yield snap
    finally:
        stack.pop()


@contextmanager
def use_pass(snapshot: torch.Tensor):
    """Run the enclosed forward pass with a snapshot drawn earlier by ``begin_pass`` -- lets the trainer draw the
    snapshots of several passes in program order on the main stream and then run the passes on forked streams."""
    stack = getattr(_tls, 'stack', None)
    if stack is None:
        stack = _tls.stack = []
    stack.append(snapshot)
    try:
        yield snapshot
    finally:
        stack.pop()
