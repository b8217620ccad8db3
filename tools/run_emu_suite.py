"""TEST INFRASTRUCTURE driver: the whole `-m gpu` suite on the CPU device model (tests/emu), file by file.

    python tools/run_emu_suite.py [--budget SECONDS_PER_TEST] [--sched N] [files...]  >  profiles/r05_emu_suite.txt

A test that exceeds the budget kills its pytest process (a C call cannot be interrupted); it is recorded as `too slow for
the model` and the file is re-run without it.  Tests listed in tests/emu/harness.py DESELECT never start (hipGraph capture,
second process on the device, full-size batches)."""
import glob
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_file(path, budget, sched, deselect):
    cmd = [sys.executable, '-m', 'pytest', path, '-m', 'gpu', '-v', '-p', 'no:cacheprovider', '--timeout', str(budget),
           '--timeout-method', 'thread']
    for d in deselect:
        cmd += ['--deselect', d]
    env = dict(os.environ, S2AG_EMU='1', S2AG_EMU_SCHED=str(sched))
    t0 = time.time()
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return p.returncode, p.stdout, time.time() - t0


def main():
    args = sys.argv[1:]
    budget, sched = 600, 0
    while args and args[0].startswith('--'):
        if args[0] == '--budget':
            budget = int(args[1])
        elif args[0] == '--sched':
            sched = int(args[1])
        args = args[2:]
    files = args or sorted(glob.glob(os.path.join(ROOT, 'tests', 'test_gpu_*.py')))
    out = []
    total = dict(passed=0, failed=0, slow=0, skipped=0)
    for f in files:
        rel = os.path.relpath(f, ROOT)
        verdicts, slow, deselect, secs_all = {}, [], [], 0.0
        while True:
            rc, log, secs = run_file(rel, budget, sched, deselect)
            secs_all += secs
            seen = re.findall(r'^(tests/\S+::\S+)(?: (PASSED|FAILED|SKIPPED|ERROR|XFAIL))?', log, flags=re.M)
            for n, v in seen:
                if v:
                    verdicts[n] = v
            done = re.search(r'=+ .* in [0-9.]+s', log) and '+++ Timeout +++' not in log
            if done:
                break
            # the process died inside a test (a C call cannot be interrupted): the last test announced without a verdict
            culprit = next((n for n, v in reversed(seen) if not v and n not in verdicts), None)
            if culprit is None:
                verdicts['<driver>'] = 'could not attribute a dead process: ' + log[-200:].replace('\n', ' | ')
                break
            slow.append(culprit)
            deselect = list(verdicts) + slow                      # the re-run starts behind what has a verdict already
        cnt = {k: sum(1 for v in verdicts.values() if v == k) for k in ('PASSED', 'FAILED', 'SKIPPED', 'ERROR')}
        total['passed'] += cnt['PASSED']; total['failed'] += cnt['FAILED'] + cnt['ERROR']; total['skipped'] += cnt['SKIPPED']
        total['slow'] += len(slow)
        lines = [f'{rel}: {cnt["PASSED"]} passed, {cnt["FAILED"] + cnt["ERROR"]} failed, {cnt["SKIPPED"]} skipped, {len(slow)} too slow '
                 f'for the model  ({secs_all:.0f} s, schedule {sched})']
        lines += [f'    {v} {t}' for t, v in verdicts.items() if v not in ('PASSED', 'SKIPPED')]
        lines += [f'    too slow for the model (> {budget} s): {t}' for t in slow]
        out += lines
        print('\n'.join(lines), flush=True)
    out.append(f'TOTAL: {total}')
    print(out[-1])
    return out


if __name__ == '__main__':
    main()
