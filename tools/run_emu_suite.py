"""TEST INFRASTRUCTURE driver: the whole `-m gpu` suite on the CPU device model (tests/emu), file by file.

    python tools/run_emu_suite.py [--budget SECONDS_PER_TEST] [--sched N] [files...]  ->  profiles/r04_emu_suite.txt

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
        slow, deselect = [], []
        while True:
            rc, log, secs = run_file(rel, budget, sched, deselect)
            m = re.search(r'=+ (.*) in [0-9.]+s', log)
            if m and '+++ Timeout +++' not in log:
                summary = m.group(1)
                break
            # the process died inside a test: the last test announced without a verdict is the culprit
            started = re.findall(r'^(tests/\S+::\S+)(?: (PASSED|FAILED|SKIPPED|ERROR))?', log, flags=re.M)
            culprit = next((n for n, v in reversed(started) if not v), None)
            if culprit is None or culprit in deselect:
                summary = 'driver could not attribute a dead process; log tail: ' + log[-300:].replace('\n', ' | ')
                break
            slow.append(culprit)
            deselect.append(culprit)
        failed = re.findall(r'^(tests/\S+::\S+) FAILED', log, flags=re.M)
        n = lambda w: int((re.search(r'(\d+) ' + w, summary) or [0, 0])[1])
        total['passed'] += n('passed'); total['failed'] += n('failed'); total['skipped'] += n('skipped') + n('deselected')
        total['slow'] += len(slow)
        out.append(f'{rel}: {summary}  ({secs:.0f} s, schedule {sched})')
        out += [f'    FAILED {t}' for t in failed]
        out += [f'    too slow for the model (> {budget} s): {t}' for t in slow]
        print(out[-1 - len(failed) - len(slow)], flush=True)
        for l in out[len(out) - len(failed) - len(slow):]:
            print(l, flush=True)
    out.append(f'TOTAL: {total}')
    print(out[-1])
    return out


if __name__ == '__main__':
    main()
