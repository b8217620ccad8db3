"""Build libs2ag_hip.so (gfx950) in-tree:  python -m speech2affective_gestures_amd.build [--force]

Every csrc/*.hip is compiled to its own object under csrc/_obj/ (git-ignored; re-compiled only when the source or a
header is newer), a few at a time, then linked -- editing one kernel file costs one compile, not ten."""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, 'libs2ag_hip.so')
OBJ = os.path.join(PKG, 'csrc', '_obj')
SRC = sorted(glob.glob(os.path.join(PKG, 'csrc', '*.hip')))
HDR = sorted(glob.glob(os.path.join(PKG, 'csrc', '*.h'))) + [os.path.join(ROOT, 'include', 's2ag_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'),
         '-I' + os.path.join(PKG, 'csrc')]


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(f) > t for f in deps)


def needs_build() -> bool:
    return _stale(LIB, SRC + HDR)


def build(force: bool = False, verbose: bool = True) -> str:
    """hipcc cross-compiles for gfx950 without a GPU present."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(OBJ, exist_ok=True)
    todo = [s for s in SRC if force or _stale(_obj(s), [s] + HDR)]

    def compile_one(src):
        cmd = [hipcc] + FLAGS + ['-c', src, '-o', _obj(src)]
        if verbose:
            print('[s2ag build]', ' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    with cf.ThreadPoolExecutor(max_workers=int(os.environ.get('S2AG_BUILD_JOBS', '4'))) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + [_obj(s) for s in SRC] + ['-o', LIB + '.tmp']
    if verbose:
        print('[s2ag build]', ' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
