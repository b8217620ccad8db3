#!/bin/bash
# kernel traces of BASELINE configs[3] in both modes (writes gpurun_out/r02q; summaries are copied to profiles/)
export S2AG_BENCH_SUPERVISE=0   # bench.py in THIS process (rocprofv3 then sees one process)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02q; mkdir -p $O
cd $R
for m in ${MODES:-fp32 bf16}; do
  MODE=$m python tools/run_cfg4.py > $O/run_$m.log 2>&1; tail -1 $O/run_$m.log
  S2AG_CFG3_STREAMS=1 MODE=$m python tools/run_cfg4.py > $O/run1_$m.log 2>&1; tail -1 $O/run1_$m.log | cut -c1-200
  S2AG_CFG3_STREAMS=1 MODE=$m rocprofv3 --kernel-trace --stats -d $O/$m -o cfg3 -- python tools/run_cfg4.py > $O/prof_$m.log 2>&1
  db=$(find $O/$m -name "*results.db" | head -1)
  python tools/rocpd_stats.py $db 45 > $O/r02_cfg3_${m}_kernel_stats.txt
  python tools/rocpd_by_grid.py $db _k > $O/r02_cfg3_${m}_by_grid.txt
done
find $O -name "*.db" -delete
head -40 $O/r02_cfg3_bf16_kernel_stats.txt
