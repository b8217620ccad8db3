#!/bin/bash
# Round-2 profiles on the GPU box (writes under gpurun_out/r02p; the summaries are copied to profiles/ afterwards).
set -x
export S2AG_BENCH_SUPERVISE=0   # bench.py in THIS process (rocprofv3 then sees one process)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02p; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/step -o step -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/step.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/cfg3 -o cfg3 -- python tools/run_cfg4.py > $O/cfg3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o pf -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-graph > $O/pf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o pw -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-graph > $O/pw.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/cf -o cf -- python tools/pmc_calib.py > $O/cf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/cw -o cw -- python tools/pmc_calib.py > $O/cw.log 2>&1
find $O -name "*.db" | head -20
db() { find $O/$1 -name "*results.db" | head -1; }
python tools/rocpd_stats.py $(db step) 45 > $O/r02_step_kernel_stats.txt
python tools/rocpd_stats.py $(db cfg3) 40 > $O/r02_cfg3_kernel_stats.txt
python tools/rocpd_pmc.py $(db pf) > $O/r02_pmc_FETCH_SIZE.txt
python tools/rocpd_pmc.py $(db pw) > $O/r02_pmc_WRITE_SIZE.txt
python tools/rocpd_pmc.py $(db cf) > $O/r02_pmc_calib_FETCH_SIZE.txt
python tools/rocpd_pmc.py $(db cw) > $O/r02_pmc_calib_WRITE_SIZE.txt
python tools/pmc_traffic.py $(db pf) $(db pw) $(db cf) $(db cw) > $O/r02_pmc_traffic.json
find $O -name "*.db" -size +20M -delete
ls -la $O
