"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel table:
    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db [top_n] > profiles/rNN_xxx_kernel_stats.txt
"""
import re
import sqlite3
import sys


def main(path, top=40):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith('rocpd_kernel_dispatch'))
    ks = next(t for t in tabs if t.startswith('rocpd_info_kernel_symbol'))
    cols = [r[1] for r in c.execute(f'pragma table_info({kd})')]
    scols = [r[1] for r in c.execute(f'pragma table_info({ks})')]
    name_col = 'kernel_name' if 'kernel_name' in scols else ('display_name' if 'display_name' in scols else scols[-1])
    q = (f'select s.{name_col}, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), '
         f'max(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} '
         f'order by 3 desc')
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows)
    print(f'# {path}: {sum(r[1] for r in rows)} dispatches, {total / 1e6:.3f} ms of kernel time')
    print(f'{"kernel":<70} {"calls":>7} {"total_ms":>10} {"avg_us":>10} {"min_us":>9} {"max_us":>9} {"pct":>6}')
    for name, n, tot, avg, mn, mx in rows[:top]:
        short = re.sub(r'\(.*', '', name)
        short = re.sub(r'^void ', '', short)[:70]
        print(f'{short:<70} {n:>7} {tot / 1e6:>10.3f} {avg / 1e3:>10.2f} {mn / 1e3:>9.2f} {mx / 1e3:>9.2f} '
              f'{100.0 * tot / total:>6.2f}')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
