"""Wall-clock occupancy of the last steps of a bench run: union of kernel intervals vs sum, biggest contributors."""
import sqlite3, sys, re, collections
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith('rocpd_kernel_dispatch')); ks = next(t for t in tabs if t.startswith('rocpd_info_kernel_symbol'))
rows = list(c.execute(f"select d.start, d.end, s.kernel_name, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
# take the last third of the run (timed steps)
t_end = rows[-1][1]; t_beg = rows[0][0]
cut = t_end - (t_end - t_beg) * float(sys.argv[2]) if len(sys.argv) > 2 else t_beg
rows = [r for r in rows if r[0] >= cut]
span = rows[-1][1] - rows[0][0]
tot = sum(e - s for s, e, _, _ in rows)
# union
u = 0; cur_s, cur_e = rows[0][0], rows[0][1]
for s, e, _, _ in rows[1:]:
    if s > cur_e: u += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
u += cur_e - cur_s
print(f'span {span/1e6:.2f} ms, sum of kernels {tot/1e6:.2f} ms, union busy {u/1e6:.2f} ms ({100*u/span:.1f}% of span), gaps {(span-u)/1e6:.2f} ms, queues {len(set(r[3] for r in rows))}')
by = collections.defaultdict(float)
for s, e, n, _ in rows: by[re.sub(r'^_ZN12_GLOBAL__N_1\d+|^_ZN4s2agL\d+', '', n)[:40]] += e - s
for n, v in sorted(by.items(), key=lambda x: -x[1])[:14]: print(f'  {n:<42} {v/1e6:8.2f} ms  {100*v/tot:5.1f}%')
