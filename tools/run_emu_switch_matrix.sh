#!/bin/bash
# TEST INFRASTRUCTURE: the suites that go THROUGH each opt-in switch, on the CPU device model (what tools/gpu_pending.sh step 2
# does on a GPU), plus the wave / pending suites under two more wavefront schedules and with the loaders' asserts live.
#   bash tools/run_emu_switch_matrix.sh >> profiles/r04_emu_suite.txt
cd "$(dirname "$0")/.."
B=${BUDGET:-600}
run() { echo "--- $1"; shift; env "$@"; }
run "S2AG_WAVE_TAIL32=1 (fp32 wave tail with folded BatchNorms as the encoder of every module / step test)" S2AG_WAVE_TAIL32=1 python tools/run_emu_suite.py --budget $B tests/test_gpu_wave12.py tests/test_gpu_modules.py
run "S2AG_TCN_GATHER=1 (embedding gather inside the TCN launch, both modes)" S2AG_TCN_GATHER=1 python tools/run_emu_suite.py --budget $B tests/test_gpu_bf16.py tests/test_gpu_step.py
run "S2AG_TCN_RING_DEEP=1 (deep weight rings)" S2AG_TCN_RING_DEEP=1 python tools/run_emu_suite.py --budget $B tests/test_gpu_bf16.py
run "S2AG_EMB_FWD_ROWS=1 (row-form embedding forward)" S2AG_EMB_FWD_ROWS=1 python tools/run_emu_suite.py --budget $B tests/test_gpu_ops.py
run "S2AG_W12_FWD_PIPE=1 (pipelined head forward)" S2AG_W12_FWD_PIPE=1 python tools/run_emu_suite.py --budget $B tests/test_gpu_wave12.py
for s in 2 3; do
  run "wavefront schedule $s (2: reversed order, 3: random; a result that moves is a missing barrier)" python tools/run_emu_suite.py --budget $B --sched $s tests/test_gpu_wave12.py tests/test_gpu_wave_fused.py tests/test_gpu_zz_pending_wave32.py tests/test_gpu_zz_pending_tcn.py
done
run "loaders' index asserts live (-DS2AG_DEBUG=1)" S2AG_EMU_DEBUG=1 python tools/run_emu_suite.py --budget $B tests/test_gpu_wave12.py tests/test_gpu_wave_fused.py tests/test_gpu_zz_pending_wave32.py
