#!/bin/bash
# The FIRST gpurun call of the round in which GPU access comes back (closed since the middle of r03):
#   /usr/local/graft/bin/gpurun --timeout 4500 -- 'TAG=r07 bash tools/gpu_first_call.sh'
#   1. the full -m gpu suite at HEAD (no -x), with durations  -> gpurun_out/first/t_all.log
#   2. one full un-profiled bench line of HEAD               -> gpurun_out/first/bench_line.json
#   3. tools/ab_variants.py: every opt-in variant against its default, interleaved in one process -> gpurun_out/first/ab_variants.txt
#   4. TAG=<round> tools/profile_round.sh -- kernel stats, FETCH/WRITE_SIZE traffic, MFMA-busy for the step and configs[3]
# ~50 GPU-minutes (each part has its own timeout: 40 + 15 + 15 min + the profiles).  Copy the summaries (gpurun_out/<TAG>p/<TAG>_*) into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/first; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu --durations=25 > $O/t_all.log 2>&1; echo "full suite rc=$?"; tail -40 $O/t_all.log
timeout 900 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"; grep '^{"metric"' $O/bench.log | tail -1 > $O/bench_line.json; cut -c1-400 $O/bench_line.json
# 4. the opt-in variants written while the GPU was closed: parity already ran with the suite (tests/test_gpu_zy_variants.py); A/B timing
timeout 900 python tools/ab_variants.py --rounds 15 > $O/ab_variants.txt 2>&1; echo "ab rc=$?"; grep -v '^AB ' $O/ab_variants.txt | tail -30
TAG=${TAG:-r07} bash $R/tools/profile_round.sh > $O/profile.log 2>&1; tail -5 $O/profile.log
