import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import pytest, torch
import test_gpu_modules as M
import test_gpu_step as S
from speech2affective_gestures_amd import ops, noise
GOLD = os.path.join(ROOT, 'tests', 'golden')
mode = sys.argv[1] if len(sys.argv) > 1 else 'none'

def run(fn, *a):
    mp = pytest.MonkeyPatch()
    try:
        if 'monkeypatch' in fn.__code__.co_varnames[:fn.__code__.co_argcount]:
            fn(*a, mp)
        else:
            fn(*a)
        return 'ok'
    except AssertionError as e:
        return 'FAIL ' + str(e)[:300]
    finally:
        mp.undo()

for args in (('G', 32, 64, 3), ('GA', 32, 64, 3), ('G', 300, 2000, 88)):
    print('H', args, run(M.test_generator_train_mode_with_dropout_forward_and_all_gradients, *args)[:60], flush=True)
print('three', run(S.test_three_steps_match_the_reference_trace, GOLD), flush=True)
print('state: ticket idx', ops._TICKETS[0][1], 'generation', ops._GENERATION[0], 'sites', noise._site_counter[0], flush=True)
if mode == 'tickets':
    ops._TICKETS[0][1] = 0
if mode == 'sites':
    noise.reset_sites(0)
if mode == 'gen':
    ops._GENERATION[0] = 0
if mode == 'empty_cache':
    torch.cuda.synchronize(); torch.cuda.empty_cache()
if mode == 'ticketzero':
    print('nonzero tickets:', int((ops._TICKETS[0][0] != 0).sum()))
    ops._TICKETS[0][0].zero_()
print('two', run(S.test_two_steps_with_dropout_match_the_oracle), flush=True)
print('two again', run(S.test_two_steps_with_dropout_match_the_oracle), flush=True)
