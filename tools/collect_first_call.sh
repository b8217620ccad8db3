#!/bin/bash
# After tools/gpu_first_call.sh has come back: copy what is to be judged from the scratch directory gpurun_out/ into profiles/
# (tracked).   TAG=r06 bash tools/collect_first_call.sh
TAG=${TAG:-r07}
R=$(cd "$(dirname "$0")/.." && pwd); F=$R/gpurun_out/first; P=$R/profiles
[ -d $F ] || { echo "no $F: the first call has not run"; exit 1; }
cp $F/t_all.log $P/${TAG}_gpu_suite_release.txt 2>/dev/null && echo "suite   -> profiles/${TAG}_gpu_suite_release.txt: $(tail -1 $F/t_all.log)"
[ -s $F/bench_line.json ] && cp $F/bench_line.json $P/${TAG}_bench_line.json && echo "bench   -> profiles/${TAG}_bench_line.json"
[ -s $F/ab_variants.txt ] && cp $F/ab_variants.txt $P/${TAG}_ab_variants.txt && echo "A/B     -> profiles/${TAG}_ab_variants.txt" && grep '^verdict' $F/ab_variants.txt
for f in $R/gpurun_out/${TAG}p/${TAG}_*; do [ -f "$f" ] && cp "$f" $P/ && echo "profile -> profiles/$(basename $f)"; done
python3 - <<PY
import json, sys
try:
    d = json.load(open('$F/bench_line.json'))
except Exception as e:
    sys.exit(0)
print('value', d.get('value'), d.get('unit'), '| ms/step', d.get('ms_per_step'), '| fp32-equivalent', d.get('value_fp32_equivalent'))
r = d.get('roofline') or {}
print('roofline frac', r.get('frac'), 'of executing pipe', r.get('frac_of_executing_pipe'), 'traffic', r.get('traffic'))
c = d.get('conv1d_roofline_run') or {}
print('configs[3] fp32', c.get('ms_per_iter'), 'ms =', (c.get('roofline') or {}).get('frac'), 'of HBM')
print('variants all on', d.get('opt_in_variants_all_on'))
PY
