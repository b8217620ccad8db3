import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1]); pat = sys.argv[2] if len(sys.argv) > 2 else 'conv_'
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith('rocpd_kernel_dispatch')); ks = next(t for t in tabs if t.startswith('rocpd_info_kernel_symbol'))
q = f"select s.kernel_name, d.workgroup_size_x, d.grid_size_x/d.workgroup_size_x, d.grid_size_y, d.grid_size_z, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%{pat}%' group by 1,2,3,4,5 order by 8 desc limit 45"
for r in c.execute(q):
    name = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', r[0])[:44]
    print(f'{name:<46} wg{r[1]:<4} grid({r[2]},{r[3]},{r[4]}) n={r[5]:<5} avg={r[6]:8.1f}us total={r[7]:8.2f}ms')
