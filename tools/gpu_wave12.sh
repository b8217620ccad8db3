#!/bin/bash
# wave12 bring-up on the GPU box: kernel tests, the encoder tests that go through it, then configs[3] timings + a kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/w12; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_wave12.py -x -q -m gpu -s > $O/t_wave12.log 2>&1; echo "wave12 tests rc=$?"; tail -15 $O/t_wave12.log
timeout 900 python -m pytest tests/test_gpu_wave_fused.py tests/test_gpu_bf16.py -x -q -m gpu > $O/t_fused.log 2>&1; echo "fused/bf16 tests rc=$?"; tail -5 $O/t_fused.log
for m in fp32 bf16; do
  MODE=$m timeout 600 python tools/run_cfg4.py > $O/run_$m.log 2>&1; tail -1 $O/run_$m.log | cut -c1-200
  S2AG_WAVE12=0 MODE=$m timeout 600 python tools/run_cfg4.py > $O/run_${m}_off.log 2>&1; tail -1 $O/run_${m}_off.log | cut -c1-200
done
for m in fp32 bf16; do
  S2AG_CFG3_STREAMS=1 MODE=$m timeout 600 rocprofv3 --kernel-trace --stats -d $O/cfg3_$m -o cfg3 -- python tools/run_cfg4.py > $O/prof_$m.log 2>&1
  python tools/rocpd_by_grid.py $(find $O/cfg3_$m -name "*results.db" | head -1) _k > $O/cfg3_${m}_by_grid.txt
  head -24 $O/cfg3_${m}_by_grid.txt
done
find $O -name "*.db" -delete
