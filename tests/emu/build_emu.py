"""TEST INFRASTRUCTURE: build tests/emu/_build/libs2ag_emu.so -- the product's own kernel sources
(speech2affective_gestures_amd/csrc/*.hip) compiled for the HOST against the CPU device model of tests/emu/hip/hip_runtime.h.

    python tests/emu/build_emu.py [--force]

The sources are used as they are except for three textual rewrites that a header cannot express (done on a copy under
_build/src, line numbers kept):
  * `extern __shared__ [attrs] T name[];`  ->  `T* name = (T*)emu::dyn_smem();`   (dynamic LDS of the running workgroup)
  * `asm volatile("s_waitcnt ...")`        ->  nothing   (waits for the wave's own memory operations: no-ops in this model)
  * `asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier")` -> `__syncthreads()`
The library exports the same C ABI as libs2ag_hip.so (include/s2ag_hip.h) plus s2ag_emu_set_sched / s2ag_emu_counters."""
import concurrent.futures as cf
import glob
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'speech2affective_gestures_amd', 'csrc')
DEBUG = os.environ.get('S2AG_EMU_DEBUG', '0') == '1'      # -DS2AG_DEBUG=1: the loaders' index asserts (s2ag_common.h) are live
OUT = os.path.join(HERE, '_build', 'debug') if DEBUG else os.path.join(HERE, '_build')
LIB = os.path.join(OUT, 'libs2ag_emu_debug.so' if DEBUG else 'libs2ag_emu.so')
CXX = os.environ.get('S2AG_EMU_CXX', '/opt/rocm/lib/llvm/bin/clang++')
FLAGS = ['-x', 'c++', '-std=c++17', '-O1', '-g1', '-fPIC', '-fno-strict-aliasing', '-ffp-contract=off', '-pthread',
         '-Wno-unknown-attributes', '-Wno-ignored-attributes', '-Wno-unused-value', '-Wno-pass-failed',
         '-Wno-unknown-pragmas', '-Wno-deprecated-declarations',
         '-DS2AG_DET=1',      # the model's library always carries the deterministic mode (the GPU build: flavour `det` only)
         '-DS2AG_DET_SPIN_LIMIT=(1<<30)',    # a poll of the model is a host-thread yield, not ~100 device cycles
         '-I' + HERE, '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(OUT, 'src')] + (['-DS2AG_DEBUG=1'] if DEBUG else [])

_DYN = re.compile(r'extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_][\w ]*?)\s+(\w+)\[\];')
_ASM_BAR = re.compile(r'asm volatile\("s_waitcnt lgkmcnt\(0\)\\n\\ts_barrier"\s*:::\s*"memory"\)')
_ASM_WAIT = re.compile(r'asm volatile\("s_waitcnt [a-z]+cnt\(\d+\)"\s*:::\s*"memory"\)')


def rewrite(text: str) -> str:
    text = _DYN.sub(lambda m: '%s* %s = (%s*)emu::dyn_smem();' % (m.group(1), m.group(2), m.group(1)), text)
    text = _ASM_BAR.sub('__syncthreads()', text)
    text = _ASM_WAIT.sub('((void)0)', text)
    if 'asm volatile' in text or 'extern __shared__' in text:
        bad = [l for l in text.splitlines() if 'asm volatile' in l or 'extern __shared__' in l]
        raise RuntimeError('construct the emulator build does not know how to rewrite: ' + bad[0].strip())
    return text


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _digest():
    h = hashlib.sha256()
    files = sources() + sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.join(ROOT, 'include', 's2ag_hip.h'),
                                                                       os.path.join(HERE, 'hip', 'hip_runtime.h'),
                                                                       os.path.join(HERE, 'emu_rt.cpp'), os.path.abspath(__file__)]
    for f in files:
        h.update(f.encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(os.path.join(OUT, 'src'), exist_ok=True)
    stamp = os.path.join(OUT, 'digest.txt')
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    for h in glob.glob(os.path.join(CSRC, '*.h')):
        open(os.path.join(OUT, 'src', os.path.basename(h)), 'w').write(rewrite(open(h).read()))
    jobs = []
    for s in sources():
        base = os.path.basename(s)[:-4]
        cpp = os.path.join(OUT, 'src', base + '.cpp')
        new = rewrite(open(s).read())
        if force or not os.path.exists(cpp) or open(cpp).read() != new:
            open(cpp, 'w').write(new)
        jobs.append((cpp, os.path.join(OUT, base + '.o')))
    jobs.append((os.path.join(HERE, 'emu_rt.cpp'), os.path.join(OUT, 'emu_rt.o')))
    hdr_time = max(os.path.getmtime(f) for f in glob.glob(os.path.join(OUT, 'src', '*.h')) +
                   [os.path.join(HERE, 'hip', 'hip_runtime.h'), os.path.join(ROOT, 'include', 's2ag_hip.h')])

    def compile_one(job):
        src, obj = job
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_time):
            return
        cmd = [CXX] + FLAGS + ['-c', src, '-o', obj]
        if verbose:
            print('[s2ag emu build]', ' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    with cf.ThreadPoolExecutor(max_workers=int(os.environ.get('S2AG_BUILD_JOBS', '8'))) as ex:
        list(ex.map(compile_one, jobs))
    cmd = [CXX, '-shared', '-fPIC', '-pthread'] + [o for _, o in jobs] + ['-o', LIB + '.tmp']
    if verbose:
        print('[s2ag emu build]', ' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + '.tmp', LIB)
    open(stamp, 'w').write(dig)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
