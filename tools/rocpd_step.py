"""One training step out of a bench.py kernel trace (rocprofv3 --kernel-trace): the window between two consecutive
generator-Adam launches in the middle of the timed region.  Prints wall span, union-busy time, idle gaps, and the
per-kernel launch counts / time inside that one step."""
import collections
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith('rocpd_kernel_dispatch'))
ks = next(t for t in tabs if t.startswith('rocpd_info_kernel_symbol'))
rows = list(c.execute(f"select d.start, d.end, s.kernel_name, d.queue_id from {kd} d join {ks} s "
                      f"on d.kernel_id=s.id order by d.start"))
adam = [i for i, r in enumerate(rows) if 'adam_k' in r[2]]
# Adam launches come in (D, G) pairs per step; pick a pair boundary two thirds into the run
k = (len(adam) * 2 // 3) // 2 * 2
lo, hi = adam[k - 1] + 1, adam[k + 1] + 1
win = rows[lo:hi]
span = win[-1][1] - win[0][0]
tot = sum(e - s for s, e, _, _ in win)
u, (cs, ce) = 0, win[0][:2]
gaps = []
for s, e, _, _ in win[1:]:
    if s > ce:
        u += ce - cs
        gaps.append(s - ce)
        cs, ce = s, e
    else:
        ce = max(ce, e)
u += ce - cs
print(f'step window: {len(win)} launches on {len(set(r[3] for r in win))} queues, span {span/1e6:.3f} ms, '
      f'sum of kernels {tot/1e6:.3f} ms, union busy {u/1e6:.3f} ms ({100*u/span:.1f}%), '
      f'{len(gaps)} idle gaps = {sum(gaps)/1e6:.3f} ms (median {sorted(gaps)[len(gaps)//2]/1e3:.1f} us)')
by = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, _ in win:
    key = re.sub(r'^_ZN12_GLOBAL__N_1\d+|^_ZN4s2agL\d+', '', n)[:48]
    by[key][0] += 1
    by[key][1] += e - s
for n, (cnt, v) in sorted(by.items(), key=lambda x: -x[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f'  {n:<50} x{cnt:<4} {v/1e3:9.1f} us  {100*v/tot:5.1f}%')
if len(sys.argv) > 3:      # full listing of the window: start offset, duration, queue, grid, kernel
    gcol = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    gx = 'grid_size_x' if 'grid_size_x' in gcol else None
    wgx = 'workgroup_size_x' if 'workgroup_size_x' in gcol else None
    extra = list(c.execute(f"select d.start, {'d.' + gx if gx else 0}, {'d.' + wgx if wgx else 1} from {kd} d order by d.start"))[lo:hi]
    qn = {q: i for i, q in enumerate(sorted(set(r[3] for r in win)))}
    with open(sys.argv[3], 'w') as f:
        for (s, e, n, q), (_, g, w) in zip(win, extra):
            key = re.sub(r'^_ZN12_GLOBAL__N_1\d+|^_ZN4s2agL\d+', '', n)[:60]
            f.write(f'{(s - win[0][0]) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{qn[q]} {int(g) // max(int(w), 1):6d}  {key}\n')
