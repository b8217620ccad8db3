#!/bin/bash
# session profile: ATen inventory, phase stamps, kernel trace of the graph-replayed step (+ one-step window)
export S2AG_BENCH_SUPERVISE=0   # bench.py in THIS process (rocprofv3 then sees one process)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-s5}; mkdir -p $O; cd $R
[ -n "$ATEN" ] && timeout 300 python tools/aten_in_step.py > $O/aten.txt 2>&1
[ -n "$PHASES" ] && timeout 300 python tools/trace_phases.py > $O/phases.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/t -o s -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/log.txt 2>&1
DB=$(find $O/t -name "*results.db" | head -1)
python tools/rocpd_stats.py $DB 60 > $O/stats.txt
python tools/rocpd_step.py $DB 40 $O/list.txt > $O/step.txt 2>&1
find $O -name "*.db" -delete
