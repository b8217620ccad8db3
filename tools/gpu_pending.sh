#!/bin/bash
# What is waiting for a GPU (written while access was closed in r03): run in ONE gpurun call, ~15 min.
#   1. the full GPU suite on HEAD (defaults)
#   2. opt-in paths: S2AG_WAVE_TAIL32=1 (fp32 wave-encoder tail, wave32.py) and S2AG_EMB_FWD_ROWS=1 (row-form embedding forward)
#      S2AG_TCN_GATHER=1 / S2AG_TCN32_GATHER=1 (embedding gather + dropout in the bf16 / fp32 TCN forward launch's loader)
#      S2AG_TCN_RING=8 / S2AG_TCN32_RING=6 (twice the weight fragments in flight in the clip-resident TCN kernels: bit-identical)
#      and S2AG_W12_FWD_PIPE=1 (software-pipelined K loop of the head's fp32 forward: bit-identical results by construction)
#      -- their own tests, then the suites that go through them
#   3. configs[3] timings with each switch off / on, and a per-grid kernel trace of fp32 mode with the tail on
# Results under gpurun_out/pending/; flip a default only when its tests are green AND its timing is not worse.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pending; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "full suite rc=$?"; tail -3 $O/t_all.log
S2AG_WAVE_TAIL32=1 timeout 900 python -m pytest tests/test_gpu_wave32.py -q -m gpu -s > $O/t_wave32.log 2>&1; echo "wave32 tests rc=$?"; grep "wave32 bwd\|passed\|failed\|Error" $O/t_wave32.log | tail -30
S2AG_WAVE_TAIL32=1 timeout 1200 python -m pytest tests/test_gpu_wave12.py tests/test_gpu_modules.py tests/test_gpu_step.py tests/test_gpu_fullsize.py -q -m gpu > $O/t_wave32_suites.log 2>&1; echo "suites with the tail on rc=$?"; tail -5 $O/t_wave32_suites.log
S2AG_EMB_FWD_ROWS=1 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16.py tests/test_gpu_modules.py -q -m gpu -k "embedding or encoders_in_bf16 or text or golden" > $O/t_emb_rows.log 2>&1; echo "embedding rows rc=$?"; tail -3 $O/t_emb_rows.log
for m in fp32 bf16; do
  MODE=$m timeout 600 python tools/run_cfg4.py > $O/run_$m.log 2>&1; echo "cfg3 $m default: $(tail -1 $O/run_$m.log | cut -c1-160)"
  S2AG_EMB_FWD_ROWS=1 MODE=$m timeout 600 python tools/run_cfg4.py > $O/run_${m}_embrows.log 2>&1; echo "cfg3 $m emb rows: $(tail -1 $O/run_${m}_embrows.log | cut -c1-160)"
done
S2AG_TCN_GATHER=1 timeout 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu > $O/t_tcn_gather.log 2>&1; echo "bf16 tests with the gather inside the TCN launch rc=$?"; tail -2 $O/t_tcn_gather.log
S2AG_TCN_GATHER=1 MODE=bf16 timeout 600 python tools/run_cfg4.py > $O/run_bf16_gather.log 2>&1; echo "cfg3 bf16 gather in the TCN launch: $(tail -1 $O/run_bf16_gather.log | cut -c1-160)"
S2AG_TCN32_GATHER=1 timeout 900 python -m pytest tests/test_gpu_tcn_gather.py tests/test_gpu_modules.py tests/test_gpu_step.py -q -m gpu > $O/t_tcn32_gather.log 2>&1; echo "fp32 gather inside the TCN launch: tests rc=$?"; tail -2 $O/t_tcn32_gather.log
S2AG_TCN32_GATHER=1 MODE=fp32 timeout 600 python tools/run_cfg4.py > $O/run_fp32_gather.log 2>&1; echo "cfg3 fp32 gather in the TCN launch: $(tail -1 $O/run_fp32_gather.log | cut -c1-160)"
S2AG_TCN32_GATHER=1 timeout 600 python bench.py --steps 30 --warmup 10 > $O/bench_gather.log 2>&1; grep '^{"metric"' $O/bench_gather.log | cut -c1-200
S2AG_TCN_RING=8 S2AG_TCN32_RING=6 timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_modules.py tests/test_gpu_step.py -q -m gpu > $O/t_tcn_ring.log 2>&1; echo "deeper weight rings in the TCN kernels: tests rc=$?"; tail -2 $O/t_tcn_ring.log
for m in fp32 bf16; do S2AG_TCN_RING=8 S2AG_TCN32_RING=6 MODE=$m timeout 600 python tools/run_cfg4.py > $O/run_${m}_ring.log 2>&1; echo "cfg3 $m deeper rings: $(tail -1 $O/run_${m}_ring.log | cut -c1-160)"; done
S2AG_TCN32_RING=6 timeout 600 python bench.py --steps 30 --warmup 10 > $O/bench_ring.log 2>&1; grep '^{"metric"' $O/bench_ring.log | cut -c1-200
S2AG_W12_FWD_PIPE=1 timeout 900 python -m pytest tests/test_gpu_wave12.py -q -m gpu > $O/t_w12_pipe.log 2>&1; echo "wave12 tests with the pipelined forward rc=$?"; tail -2 $O/t_w12_pipe.log
S2AG_W12_FWD_PIPE=1 MODE=fp32 timeout 600 python tools/run_cfg4.py > $O/run_fp32_pipe.log 2>&1; echo "cfg3 fp32 pipelined head forward: $(tail -1 $O/run_fp32_pipe.log | cut -c1-160)"
S2AG_WAVE_TAIL32=1 MODE=fp32 timeout 600 python tools/run_cfg4.py > $O/run_fp32_tail32.log 2>&1; echo "cfg3 fp32 tail32: $(tail -1 $O/run_fp32_tail32.log | cut -c1-160)"
S2AG_WAVE_TAIL32=1 S2AG_CFG3_STREAMS=1 MODE=fp32 timeout 600 rocprofv3 --kernel-trace --stats -d $O/cfg3_tail32 -o cfg3 -- python tools/run_cfg4.py > $O/prof_tail32.log 2>&1
python tools/rocpd_by_grid.py $(find $O/cfg3_tail32 -name "*results.db" | head -1) _k > $O/cfg3_fp32_tail32_by_grid.txt; head -30 $O/cfg3_fp32_tail32_by_grid.txt
S2AG_WAVE_TAIL32=1 timeout 600 python bench.py --steps 30 --warmup 10 > $O/bench_tail32.log 2>&1; grep '^{"metric"' $O/bench_tail32.log | cut -c1-200
find $O -name "*.db" -delete
