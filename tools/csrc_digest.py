#!/usr/bin/env python3
"""sha256 over the kernel sources the release library is built from (csrc/*.hip, csrc/*.h, include/s2ag_hip.h), names and
contents: the identity profiles/r06_isa_diff_since_298c878.txt was made for (tools/isa_diff_since.sh stamps it, a CPU test
compares)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def digest() -> str:
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, 'speech2affective_gestures_amd', 'csrc', '*.hip')) +
                   glob.glob(os.path.join(ROOT, 'speech2affective_gestures_amd', 'csrc', '*.h'))) + \
        [os.path.join(ROOT, 'include', 's2ag_hip.h')]
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:20]


if __name__ == '__main__':
    print(digest())
