#!/bin/bash
# What is waiting for a GPU (GPU access was closed from outside for the second half of r03 and for r04 so far): ONE gpurun call.
#   1. the full GPU suite on HEAD, defaults, no -x: the opt-in paths' own tests arm themselves (config.override / module
#      attributes) and sit in tests/test_gpu_zz_pending_*.py, which sort last
#   2. the suites that go THROUGH each opt-in path, with the switch exported (registry: speech2affective_gestures_amd/config.py)
#   3. configs[3] timings with each switch off / on, and a per-grid kernel trace of fp32 mode with the tail on
# Results under gpurun_out/pending/; flip a default only when its tests are green AND its timing is not worse; delete it otherwise.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pending; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "full suite rc=$?"; tail -3 $O/t_all.log
S2AG_WAVE_TAIL32=1 timeout 1200 python -m pytest tests/test_gpu_wave12.py tests/test_gpu_modules.py tests/test_gpu_step.py tests/test_gpu_fullsize.py -q -m gpu > $O/t_wave32_suites.log 2>&1; echo "suites with the fp32 tail on rc=$?"; tail -3 $O/t_wave32_suites.log
S2AG_EMB_FWD_ROWS=1 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16.py tests/test_gpu_modules.py -q -m gpu -k "embedding or encoders_in_bf16 or text or golden" > $O/t_emb_rows.log 2>&1; echo "embedding rows rc=$?"; tail -2 $O/t_emb_rows.log
S2AG_TCN_GATHER=1 timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_modules.py tests/test_gpu_step.py -q -m gpu > $O/t_tcn_gather.log 2>&1; echo "suites with the gather inside the TCN launch rc=$?"; tail -2 $O/t_tcn_gather.log
S2AG_TCN_RING_DEEP=1 timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_modules.py tests/test_gpu_step.py -q -m gpu > $O/t_tcn_ring.log 2>&1; echo "suites with the deeper weight rings rc=$?"; tail -2 $O/t_tcn_ring.log
S2AG_W12_FWD_PIPE=1 timeout 900 python -m pytest tests/test_gpu_wave12.py -q -m gpu > $O/t_w12_pipe.log 2>&1; echo "wave12 tests with the pipelined forward rc=$?"; tail -2 $O/t_w12_pipe.log
for m in fp32 bf16; do
  MODE=$m timeout 600 python tools/run_cfg4.py > $O/run_$m.log 2>&1; echo "cfg3 $m default: $(tail -1 $O/run_$m.log | cut -c1-160)"
  for sw in EMB_FWD_ROWS TCN_GATHER TCN_RING_DEEP; do
    env S2AG_$sw=1 MODE=$m timeout 600 python tools/run_cfg4.py > $O/run_${m}_$sw.log 2>&1; echo "cfg3 $m $sw: $(tail -1 $O/run_${m}_$sw.log | cut -c1-160)"
  done
done
for sw in W12_FWD_PIPE WAVE_TAIL32; do
  env S2AG_$sw=1 MODE=fp32 timeout 600 python tools/run_cfg4.py > $O/run_fp32_$sw.log 2>&1; echo "cfg3 fp32 $sw: $(tail -1 $O/run_fp32_$sw.log | cut -c1-160)"
done
for sw in TCN_GATHER TCN_RING_DEEP WAVE_TAIL32; do
  env S2AG_$sw=1 timeout 600 python bench.py --steps 30 --warmup 10 > $O/bench_$sw.log 2>&1; echo "bench $sw: $(grep '^{"metric"' $O/bench_$sw.log | cut -c1-200)"
done
S2AG_WAVE_TAIL32=1 S2AG_CFG3_STREAMS=1 MODE=fp32 timeout 600 rocprofv3 --kernel-trace --stats -d $O/cfg3_tail32 -o cfg3 -- python tools/run_cfg4.py > $O/prof_tail32.log 2>&1
python tools/rocpd_by_grid.py $(find $O/cfg3_tail32 -name "*results.db" | head -1) _k > $O/cfg3_fp32_tail32_by_grid.txt; head -30 $O/cfg3_fp32_tail32_by_grid.txt
find $O -name "*.db" -delete
