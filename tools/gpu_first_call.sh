#!/bin/bash
# The FIRST gpurun call of the round in which GPU access comes back (it has been closed since the middle of r03):
#   1. tools/gpu_pending.sh  -- full suite at HEAD (the never-hardware-run paths arm themselves, last), the suites through each
#                               opt-in switch, configs[3] and bench A/B per switch
#   2. one full un-profiled bench line of HEAD  -> gpurun_out/first/bench_line.json
#   3. TAG=<round> tools/profile_round.sh       -- kernel stats, FETCH/WRITE_SIZE traffic, MFMA-busy for step and configs[3]
# ~45 GPU-minutes.  Then: flip every switch whose tests are green and whose A/B is not worse, delete the others
# (speech2affective_gestures_amd/config.py), copy the summaries into profiles/.
#   /usr/local/graft/bin/gpurun --timeout 3400 -- 'TAG=r05 bash tools/gpu_first_call.sh'
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/first
bash $R/tools/gpu_pending.sh 2>&1 | tee $R/gpurun_out/first/pending_summary.txt
cd $R; timeout 900 python bench.py > $R/gpurun_out/first/bench.log 2>&1; grep '^{"metric"' $R/gpurun_out/first/bench.log | tail -1 > $R/gpurun_out/first/bench_line.json
S2AG_PRECISION=bf16_step timeout 600 python bench.py --steps 30 --warmup 10 > $R/gpurun_out/first/bench_bf16_step.log 2>&1; grep '^{"metric"' $R/gpurun_out/first/bench_bf16_step.log | tail -1 | cut -c1-300
TAG=${TAG:-r05} bash $R/tools/profile_round.sh > $R/gpurun_out/first/profile.log 2>&1; tail -5 $R/gpurun_out/first/profile.log
