import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import pytest, torch
import test_gpu_step as S
from speech2affective_gestures_amd import ops, noise

def run(fn, *a):
    mp = pytest.MonkeyPatch()
    try:
        fn(*a, mp)
        return 'ok'
    except AssertionError as e:
        return 'FAIL ' + str(e)[:120]
    finally:
        mp.undo()
for off in list(range(0, 24)) + [31, 32, 33, 63, 64, 65, 78, 100, 127, 128, 129, 255, 256, 1000]:
    noise.reset_sites(off)
    print(off, run(S.test_two_steps_with_dropout_match_the_oracle), flush=True)
