#!/bin/bash
# kernel trace of 10 steps, per-queue listing of one step: gpurun_out/qs/queues_$TAG.txt (+ list_$TAG.txt)
export S2AG_BENCH_SUPERVISE=0   # bench.py in THIS process (rocprofv3 then sees one process)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qs; mkdir -p $O; cd $R
rocprofv3 --kernel-trace -d $O/t_$TAG -o s -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/log_$TAG.txt 2>&1
DB=$(find $O/t_$TAG -name "*results.db" | head -1)
python tools/rocpd_queues.py $DB $O/list_$TAG.txt > $O/queues_$TAG.txt
python tools/rocpd_step.py $DB 12 >> $O/queues_$TAG.txt
find $O -name "*.db" -delete
