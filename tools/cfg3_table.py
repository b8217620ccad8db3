"""Per-kernel roofline table of BASELINE configs[3] from the tracked r03 evidence: time per launch (rocprofv3 kernel trace, one
stream), measured HBM bytes per launch (FETCH_SIZE / WRITE_SIZE passes + calibration), the HBM rate that implies, MFMA utilisation
(SQ_VALU_MFMA_BUSY_CYCLES).  usage: python tools/cfg3_table.py > profiles/r03_cfg3_roofline_table.md"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')


def short(mangled):
    m = re.match(r'(?:_ZN?[0-9A-Za-z_]*?\d+)?([a-z][a-z0-9_]*?_k)(?=[A-Z]|$|\.)', mangled)
    return m.group(1) if m else mangled.split('.')[0]


def main():
    print('# configs[3] (WavEncoder + TextEncoderTCN fwd+bwd, B = 256): per-kernel time, measured HBM traffic, MFMA utilisation\n')
    print('Sources: `r03_cfg3_<mode>_by_grid.txt` (us per launch, 34 iterations on one stream), `r03_pmc_traffic_cfg3_<mode>.json` '
          '(bytes per launch), `r03_mfma_util_cfg3_<mode>.txt`.  HBM peak 8 000 GB/s.  Kernels below 4 us are omitted.  Traffic and '
          'MFMA utilisation are per kernel NAME: template instantiations that share a name (the two shapes of `wv_dgrad_k`, `wv_fwd_k`, '
          '`wv_wgrad_k`, `gemm_lin_k`, `wgrad_tr32_k` ...) show their launch-weighted average there, their own time and grid here.\n')
    for mode in ('bf16', 'fp32'):
        tr = json.load(open(os.path.join(P, f'r03_pmc_traffic_cfg3_{mode}.json')))['kernels']
        util = {}
        for ln in open(os.path.join(P, f'r03_mfma_util_cfg3_{mode}.txt')):
            parts = ln.split()
            if len(parts) >= 3 and parts[0] != 'kernel':
                util[short(parts[0])] = float(parts[2])
        rows, tot = [], 0.0
        for ln in open(os.path.join(P, f'r03_cfg3_{mode}_by_grid.txt')):
            m = re.match(r'(\S+)\s+wg(\d+)\s+grid\(([^)]*)\)\s+n=(\d+)\s+avg=\s*([\d.]+)us', ln)
            if not m:
                continue
            name, grid, n, us = short(m.group(1)), m.group(3), int(m.group(4)), float(m.group(5))
            if n < 30:
                continue
            tot += us * n / 34.0
            if us < 4.0:
                continue
            b = tr.get(name, {}).get('bytes_per_launch')
            rows.append((us, name, grid, n // 34, b, util.get(name)))
        print(f'## {mode} mode: {tot:.0f} us of kernel time per iteration on one stream\n')
        print('| kernel | grid | launches / iter | us / launch | MB / launch (measured) | GB/s | % of HBM peak | MFMA util % |')
        print('|---|---|---|---|---|---|---|---|')
        for us, name, grid, n, b, u in sorted(rows, reverse=True):
            mb = f'{b / 1e6:.1f}' if b else '-'
            gbs = f'{b / us / 1e3:.0f}' if b else '-'
            pct = f'{100.0 * b / us / 1e3 / 8000.0:.1f}' if b else '-'
            print(f'| `{name}` | ({grid}) | {n} | {us:.1f} | {mb} | {gbs} | {pct} | {u if u is not None else "-"} |')
        print()


main()
