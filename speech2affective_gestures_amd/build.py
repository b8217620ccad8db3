"""Build libs2ag_hip.so (gfx950) in-tree:  python -m speech2affective_gestures_amd.build [--force] [--debug | --det | --asan]

Every csrc/*.hip is compiled to its own object under csrc/_obj/ (git-ignored; re-compiled only when the source or a
header is newer), a few at a time, then linked -- editing one kernel file costs one compile, not ten.

Flavours (same ABI, selected at run time with S2AG_HIP_LIB=<path>):
  release  libs2ag_hip.so        -O3
  debug    libs2ag_hip_debug.so  -O1 -g -DS2AG_DEBUG=1: device-side bounds asserts in the loaders (S2AG_DBG_ASSERT, s2ag_common.h:
                                 LDS image offsets, segment indices, row / column ranges) and host-side argument checks
  det      libs2ag_hip_det.so    -O3 -DS2AG_DET=1: the accumulating kernels order their atomics by workgroup index when
                                 s2ag_set_deterministic installs a turn word (s2ag_common.h); the release kernels carry no
                                 trace of the mode
  asan     libs2ag_hip_asan.so   -fsanitize=address -shared-libsan -g on gfx950:xnack+ (run with HSA_XNACK=1 and the ROCm ASan
                                 runtime on LD_LIBRARY_PATH: instrumented device loads / stores + the host glue)"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, 'libs2ag_hip.so')
OBJ = os.path.join(PKG, 'csrc', '_obj')
FLAVOURS = {
    'release': dict(lib=LIB, obj=OBJ, arch='gfx950', extra=['-O3']),
    'debug': dict(lib=os.path.join(PKG, 'libs2ag_hip_debug.so'), obj=OBJ + '_debug', arch='gfx950',
                  extra=['-O1', '-g', '-DS2AG_DEBUG=1']),
    'det': dict(lib=os.path.join(PKG, 'libs2ag_hip_det.so'), obj=OBJ + '_det', arch='gfx950', extra=['-O3', '-DS2AG_DET=1']),
    'asan': dict(lib=os.path.join(PKG, 'libs2ag_hip_asan.so'), obj=OBJ + '_asan', arch='gfx950:xnack+',
                 extra=['-O1', '-g', '-fsanitize=address', '-shared-libsan', '-DS2AG_DEBUG=1']),
}
SRC = sorted(glob.glob(os.path.join(PKG, 'csrc', '*.hip')))
HDR = sorted(glob.glob(os.path.join(PKG, 'csrc', '*.h'))) + [os.path.join(ROOT, 'include', 's2ag_hip.h')]


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(f) > t for f in deps)


def needs_build(flavour: str = 'release') -> bool:
    return _stale(FLAVOURS[flavour]['lib'], SRC + HDR)


def build(force: bool = False, verbose: bool = True, flavour: str = 'release') -> str:
    """hipcc cross-compiles for gfx950 without a GPU present."""
    fl = FLAVOURS[flavour]
    lib, objdir = fl['lib'], fl['obj']
    if flavour in ('release', 'det') and not force and not needs_build(flavour):
        return lib

    def obj(src):
        return os.path.join(objdir, os.path.basename(src)[:-4] + '.o')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(objdir, exist_ok=True)
    todo = [s for s in SRC if force or _stale(obj(s), [s] + HDR)]
    flags = ['--offload-arch=' + fl['arch'], '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'),
             '-I' + os.path.join(PKG, 'csrc')] + fl['extra']

    def compile_one(src):
        cmd = [hipcc] + flags + ['-c', src, '-o', obj(src)]
        if verbose:
            print('[s2ag build]', ' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    jobs = int(os.environ.get('S2AG_BUILD_JOBS', '4'))     # registered in config.py (BUILD_JOBS); read here so that build.py runs stand-alone
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(compile_one, todo))
    if todo or not os.path.exists(lib):
        link_extra = ['-fsanitize=address', '-shared-libsan'] if flavour == 'asan' else []
        cmd = [hipcc, '--offload-arch=' + fl['arch'], '-shared', '-fPIC'] + link_extra + [obj(s) for s in SRC] + ['-o', lib + '.tmp']
        if verbose:
            print('[s2ag build]', ' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(lib + '.tmp', lib)
    return lib


if __name__ == '__main__':
    fl = 'debug' if '--debug' in sys.argv else 'asan' if '--asan' in sys.argv else 'det' if '--det' in sys.argv else 'release'
    print(build(force='--force' in sys.argv, flavour=fl))
