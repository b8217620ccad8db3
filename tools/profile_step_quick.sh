#!/bin/bash
# quick kernel trace of the step (10 steps); writes gpurun_out/qs/stats_$TAG.txt
export S2AG_BENCH_SUPERVISE=0   # bench.py in THIS process (rocprofv3 then sees one process)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qs; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O/t_$TAG -o s -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/log_$TAG.txt 2>&1
python tools/rocpd_stats.py $(find $O/t_$TAG -name "*results.db" | head -1) 24 > $O/stats_$TAG.txt
find $O -name "*.db" -delete
