"""Build libs2ag_hip.so (gfx950) in-tree:  python -m speech2affective_gestures_amd.build [--force]"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, 'libs2ag_hip.so')
SRC = sorted(glob.glob(os.path.join(PKG, 'csrc', '*.hip')))
HDR = sorted(glob.glob(os.path.join(PKG, 'csrc', '*.h'))) + [os.path.join(ROOT, 'include', 's2ag_hip.h')]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in SRC + HDR)


def build(force: bool = False, verbose: bool = True) -> str:
    """hipcc cross-compiles for gfx950 without a GPU present."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
           '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(PKG, 'csrc')] + SRC + ['-o', LIB + '.tmp']
    if verbose:
        print('[s2ag build]', ' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
