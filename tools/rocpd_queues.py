"""Per-queue listing of one training step out of a bench.py kernel trace (see rocpd_step.py for the window): for every
hardware queue the launches in order with start (us from the step's first launch), duration and the idle gap since
the previous launch of that queue ended.  argv: results.db [full-list-file]"""
import collections
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith('rocpd_kernel_dispatch'))
ks = next(t for t in tabs if t.startswith('rocpd_info_kernel_symbol'))
rows = list(c.execute(f"select d.start, d.end, s.kernel_name, d.queue_id, d.grid_size_x*d.grid_size_y*d.grid_size_z, "
                      f"d.workgroup_size_x*d.workgroup_size_y*d.workgroup_size_z from {kd} d join {ks} s "
                      f"on d.kernel_id=s.id order by d.start"))
adam = [i for i, r in enumerate(rows) if 'adam_k' in r[2]]
k = (len(adam) * 2 // 3) // 2 * 2
lo, hi = adam[k - 1] + 1, adam[k + 1] + 1
win = rows[lo:hi]
t0 = win[0][0]
print(f'step window: {len(win)} launches, span {(win[-1][1] - t0) / 1e3:.1f} us')
byq = collections.defaultdict(list)
for r in win:
    byq[r[3]].append(r)
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else None
for q, rs in sorted(byq.items(), key=lambda x: x[1][0][0]):
    busy = sum(e - s for s, e, *_ in rs)
    gaps = [max(0, rs[i][0] - rs[i - 1][1]) for i in range(1, len(rs))]
    small = sum(1 for g in gaps if g < 20e3)
    print(f'queue {q}: {len(rs):4d} launches, first {(rs[0][0] - t0) / 1e3:8.1f} us, last end {(rs[-1][1] - t0) / 1e3:8.1f} us, '
          f'busy {busy / 1e3:8.1f} us, gaps<20us: {small} totalling {sum(g for g in gaps if g < 20e3) / 1e3:7.1f} us '
          f'(median {sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0:.1f} us)')
    if out:
        out.write(f'--- queue {q}\n')
        pe = None
        for s, e, n, _, g, w in rs:
            n = re.sub(r'^_ZN12_GLOBAL__N_1\d+|^_ZN4s2agL\d+', '', n)[:44]
            out.write(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {((s - pe) / 1e3 if pe else 0):8.1f}  wg {g // max(w, 1):6d}  {n}\n')
            pe = e
