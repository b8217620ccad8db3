"""Per-kernel MFMA utilisation from a tools/rocpd_pmc.py summary of a --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_BUSY_CYCLES pass.  SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1 024 matrix pipes (16 per 16x16x32 bf16 MFMA:
checked against the instruction count of tcn_fwd_k<3>, 2.46 M MFMAs per launch); GRBM_GUI_ACTIVE has one row per XCD (8 per
launch), each the launch's duration in cycles.  So per launch
    MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (mean GRBM_GUI_ACTIVE row * 256 CUs * 4 SIMDs).

    python tools/mfma_util.py profiles/r03_pmc_step_MFMA_BUSY.txt > profiles/r03_mfma_util_step.txt"""
import sys
from collections import defaultdict

rows = defaultdict(dict)
for ln in open(sys.argv[1]):
    parts = ln.split()
    if len(parts) < 6 or parts[0].startswith('#') or parts[0] == 'kernel':
        continue
    name, ctr = parts[0], parts[1]
    rows[name][ctr] = (float(parts[3]), int(parts[2]), float(parts[5]))
print(f'{"kernel":<62} {"kernel_ms_total":>15} {"MfmaUtil_%":>10} {"mfma_busy/launch":>18} {"cycles/launch":>18}')
out = []
for name, c in rows.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c and c['GRBM_GUI_ACTIVE'][0] > 0:
        busy, (act_sum, act_rows, ms) = c['SQ_VALU_MFMA_BUSY_CYCLES'][0], c['GRBM_GUI_ACTIVE']
        launches = act_rows / 8.0
        act = act_sum / act_rows                      # cycles of one launch
        out.append((ms / 8.0, name, 100.0 * (busy / launches) / (act * 256 * 4), busy / launches, act))
for ms, name, util, busy, act in sorted(out, reverse=True)[:40]:
    print(f'{name:<62} {ms:>15.3f} {util:>10.2f} {busy:>18.0f} {act:>18.0f}')
