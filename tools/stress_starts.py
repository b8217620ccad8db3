"""Fresh-process stress of the hipGraph-replayed step: N starts of `bench.py --no-extras --no-cpu-baseline` (in-process, no
supervisor) and L loops of `pytest -m gpu`, every signal death logged with the native back trace of csrc/debug.hip.

    python tools/stress_starts.py --starts 100 --pytest-loops 3 --out gpurun_out/stress

Writes <out>/summary.json ({starts, deaths, pytest_loops, pytest_deaths, ...}) and one <out>/death_<k>.log per death (tail of
stderr + stdout: Python's faulthandler trace and the native frames)."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, env, timeout):
    t0 = time.time()
    try:
        r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
        rc, out, err = r.returncode, r.stdout.decode(errors='replace'), r.stderr.decode(errors='replace')
    except subprocess.TimeoutExpired as e:
        rc, out, err = 'timeout', (e.stdout or b'').decode(errors='replace'), (e.stderr or b'').decode(errors='replace')
    return rc, out, err, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--starts', type=int, default=100)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--pytest-loops', type=int, default=0)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'stress'))
    ap.add_argument('--budget-s', type=float, default=1e9, help='stop starting new processes after this many seconds')
    ap.add_argument('--env', action='append', default=[], help='KEY=VALUE for the children (repeatable)')
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    env = dict(os.environ, S2AG_BENCH_SUPERVISE='0', S2AG_CRASH_TRACE='1', PYTHONFAULTHANDLER='1')
    for kv in a.env:
        k, v = kv.split('=', 1)
        env[k] = v
    t_begin = time.time()
    res = dict(starts=0, deaths=0, other_failures=0, pytest_loops=0, pytest_deaths=0, pytest_failures=0, death_rcs=[],
               env=a.env, seconds_per_start=[])
    cmd = [sys.executable, 'bench.py', '--no-extras', '--no-cpu-baseline', '--steps', str(a.steps), '--warmup', '2']

    def death(rc):
        return rc == 'timeout' or rc < 0 or rc in (134, 139)
    k = 0
    for i in range(a.starts):
        if time.time() - t_begin > a.budget_s:
            break
        rc, out, err, dt = run(cmd, env, 300)
        res['starts'] += 1
        res['seconds_per_start'].append(round(dt, 1))
        if death(rc):
            res['deaths'] += 1
            res['death_rcs'].append(rc)
            with open(os.path.join(a.out, f'death_{k}.log'), 'w') as f:
                f.write(f'start {i}: rc {rc}\n--- stderr tail ---\n{err[-20000:]}\n--- stdout tail ---\n{out[-2000:]}\n')
            k += 1
        elif rc != 0:
            res['other_failures'] += 1
            with open(os.path.join(a.out, f'failure_{i}.log'), 'w') as f:
                f.write(f'start {i}: rc {rc}\n{err[-8000:]}\n{out[-2000:]}\n')
        with open(os.path.join(a.out, 'summary.json'), 'w') as f:
            json.dump(res, f)
    for j in range(a.pytest_loops):
        if time.time() - t_begin > a.budget_s:
            break
        rc, out, err, dt = run([sys.executable, '-m', 'pytest', 'tests', '-m', 'gpu', '-x', '-q'], env, 1200)
        res['pytest_loops'] += 1
        if death(rc):
            res['pytest_deaths'] += 1
            res['death_rcs'].append(rc)
            with open(os.path.join(a.out, f'death_{k}.log'), 'w') as f:
                f.write(f'pytest loop {j}: rc {rc}\n--- stderr tail ---\n{err[-20000:]}\n--- stdout tail ---\n{out[-6000:]}\n')
            k += 1
        elif rc != 0:
            res['pytest_failures'] += 1
            with open(os.path.join(a.out, f'pytest_failure_{j}.log'), 'w') as f:
                f.write(out[-8000:] + '\n' + err[-4000:])
        with open(os.path.join(a.out, 'summary.json'), 'w') as f:
            json.dump(res, f)
    res['wall_s'] = round(time.time() - t_begin, 1)
    sps = res.pop('seconds_per_start')
    res['mean_seconds_per_start'] = round(sum(sps) / max(1, len(sps)), 2)
    with open(os.path.join(a.out, 'summary.json'), 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
