"""TEST INFRASTRUCTURE: geometry fuzzing of the fused kernels on the CPU device model (tests/emu).

The parity tests pin a handful of geometries; this calls their bodies with RANDOM ones (clips, samples / frames) -- the model
aborts on an LDS request beyond 160 KB, on a wavefront whose lanes disagree about a collective, on a workgroup in which no
thread can run (deadlock) and on a misaligned transpose read; the test bodies assert parity with the float64 references.
r04 found the bf16 TCN's 167 696-byte backward launch at T = 40 this way.

    python tools/fuzz_emu.py [seed] [cases]        ->  one line per case; exit code 1 on the first failure"""
import os
import random
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['S2AG_EMU'] = '1'
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'emu'), os.path.join(ROOT, 'tests', 'golden')]
import harness  # noqa: E402

harness.install()
import torch  # noqa: E402
from speech2affective_gestures_amd import noise  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    rnd = random.Random(seed)
    import test_gpu_wave12 as w12
    import test_gpu_wave_fused as wf
    import test_gpu_bf16 as b16

    def lin():                # samples -> at least 15 frames behind conv1 (stride 5, pad 1600) and 1 behind conv2
        return rnd.choice([rnd.randint(120, 900), rnd.randint(900, 6000), rnd.randint(6000, 40000)])

    menu = [
        ('wave12 stats+forward', w12.test_statistics_and_forward, lambda: (rnd.randint(1, 6), lin(), rnd.random() < 0.5)),
        ('wave12 backward', w12.test_backward, lambda: (rnd.randint(1, 5), lin(), rnd.choice([0, 0, 2, 3, 5]), rnd.random() < 0.5,
                                                        rnd.choice([1.0, 0.3]))),
        ('wave_fused forward', wf.test_fused_forward_conv, lambda: (*rnd.choice(wf.SHAPES), rnd.randint(1, 6), rnd.randint(1, 200))),
        ('wave_fused dgrad', wf.test_fused_data_gradient, lambda: (*rnd.choice(wf.SHAPES), rnd.randint(1, 6), rnd.randint(1, 200))),
        ('wave_fused wgrad', wf.test_fused_weight_gradient, lambda: (*rnd.choice(wf.SHAPES), rnd.randint(1, 8), rnd.randint(1, 200))),
        ('bf16 tcn vs layer by layer', b16.test_clip_resident_tcn_equals_the_layer_by_layer_bf16_path,
         lambda: (rnd.randint(1, 9), rnd.randint(9, 40))),
    ]
    bad = 0
    for i in range(cases):
        name, fn, draw = menu[i % len(menu)]
        args = draw()
        noise.reset_sites(0)
        print(f'      case {i:3d} {name}{args} ...', flush=True)      # (an abort of the model itself leaves this as the last line)
        try:
            fn(*args)
            print(f'ok    case {i:3d} {name}{args}', flush=True)
        except Exception as e:          # noqa: BLE001
            bad += 1
            print(f'FAIL  case {i:3d} {name}{args}: {type(e).__name__}: {str(e)[:300]}', flush=True)
            traceback.print_exc(limit=3)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
