"""Fold the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, see MI355X_MICROARCH.md "HBM") into
profiles/r02_pmc_traffic.json: per kernel, bytes per launch = FETCH_SIZE * f_read + WRITE_SIZE * f_write, with the factors
CALIBRATED on this stack from tools/pmc_calib.py's known 1 GiB access patterns (the guide's "x2 for 16 B/lane coalesced
reads" is re-measured, and the 8 B/lane agent-scope pattern of the cooperative GRU's exchange gets its own factor).

    python tools/pmc_traffic.py <fetch_step.db> <write_step.db> <fetch_calib.db> <write_calib.db> > profiles/r02_pmc_traffic.json
"""
import json
import re
import sqlite3
import sys


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tabs if x.startswith(p))
    pe, ip, kd, ks = t('rocpd_pmc_event'), t('rocpd_info_pmc'), t('rocpd_kernel_dispatch'), t('rocpd_info_kernel_symbol')
    q = (f"select s.kernel_name, count(*), sum(e.value) from {pe} e join {ip} i on e.pmc_id = i.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id where i.name = ? group by 1")
    out = {}
    for name, n, tot in c.execute(q, (counter,)):
        short = re.sub(r'\(.*', '', name)
        m = re.search(r'\d+([a-z_0-9]+_k)', short)
        key = m.group(1) if m else short
        ent = out.setdefault(key, dict(launches=0, total=0.0, symbol=short[:90]))
        ent['launches'] += n
        ent['total'] += tot
    return out


def main(fetch_db, write_db, fetch_cal, write_cal, per_iteration=0):
    GiB = float(1 << 30)
    fc, wc = per_kernel(fetch_cal, 'FETCH_SIZE'), per_kernel(write_cal, 'WRITE_SIZE')
    unit = 1024.0                                   # both counters report KB

    def factor(tab, kern):
        e = tab[kern]
        return GiB / (e['total'] / e['launches'] * unit)
    cal = dict(read_16B_per_lane=factor(fc, 'calib_read16_k'), read_8B_agent_scope=factor(fc, 'calib_read8_agent_k'),
               write_16B_per_lane=factor(wc, 'calib_write16_k'), write_8B_agent_scope=factor(wc, 'calib_write8_agent_k'))
    f, w = per_kernel(fetch_db, 'FETCH_SIZE'), per_kernel(write_db, 'WRITE_SIZE')
    kernels = {}
    for k in sorted(set(f) | set(w)):
        fk, wk = f.get(k), w.get(k)
        rd = fk['total'] / fk['launches'] * unit if fk else 0.0
        wr = wk['total'] / wk['launches'] * unit if wk else 0.0
        coop = k.startswith('gru_coop')
        # the cooperative GRU's reads are dominated by 8-byte agent-scope polling loads, its exchange writes likewise;
        # everything else streams 16 B per lane
        fr = cal['read_8B_agent_scope'] if coop else cal['read_16B_per_lane']
        fw = cal['write_8B_agent_scope'] if coop else cal['write_16B_per_lane']
        kernels[k] = dict(launches=(fk or wk)['launches'], fetch_size_kb_per_launch=rd / unit,
                          write_size_kb_per_launch=wr / unit, read_factor=fr, write_factor=fw,
                          bytes_per_launch=rd * fr + wr * fw)
    out = dict(source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, eager runs) + calibration on 1 GiB '
                      'known patterns (tools/pmc_calib.py)', calibration_bytes_per_counted_byte=cal, kernels=kernels)
    if per_iteration:       # every kernel of `per_iteration` identical iterations: HBM bytes of ONE iteration
        out['iterations'] = per_iteration
        out['bytes_per_iteration'] = sum(k['bytes_per_launch'] * k['launches'] for k in kernels.values()) / per_iteration
        out['read_bytes_per_iteration'] = sum(k['fetch_size_kb_per_launch'] * 1024.0 * k['read_factor'] * k['launches']
                                              for k in kernels.values()) / per_iteration
        out['write_bytes_per_iteration'] = sum(k['write_size_kb_per_launch'] * 1024.0 * k['write_factor'] * k['launches']
                                               for k in kernels.values()) / per_iteration
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    n = int(sys.argv[sys.argv.index('--per-iteration') + 1]) if '--per-iteration' in sys.argv else 0
    main(*sys.argv[1:5], per_iteration=n)
