"""Per-kernel sums of the PMC counters in a rocprofv3 rocpd database (one --pmc pass):
    python tools/rocpd_pmc.py <results.db> > summary.txt"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
t = lambda p: next(x for x in tabs if x.startswith(p))
pe, ip, kd, ks = t('rocpd_pmc_event'), t('rocpd_info_pmc'), t('rocpd_kernel_dispatch'), t('rocpd_info_kernel_symbol')
cols = lambda tb: [r[1] for r in c.execute(f'pragma table_info({tb})')]
print('# pmc_event cols', cols(pe)); print('# info_pmc cols', cols(ip))
q = (f"select s.kernel_name, i.name, count(*), sum(e.value), sum(d.end - d.start) from {pe} e "
     f"join {ip} i on e.pmc_id = i.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
     f"group by 1, 2 order by 5 desc")
print(f'{"kernel":<60} {"counter":<28} {"dispatches":>10} {"sum":>18} {"per_dispatch":>16} {"kernel_ms":>10}')
for name, ctr, n, tot, dur in c.execute(q):
    short = re.sub(r'^_ZN12_GLOBAL__N_1\d+|^_ZN4s2agL\d+', '', re.sub(r'\(.*', '', name))[:60]
    print(f'{short:<60} {ctr:<28} {n:>10} {tot:>18.0f} {tot / n:>16.1f} {dur / 1e6:>10.3f}')
