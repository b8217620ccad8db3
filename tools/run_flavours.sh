#!/bin/bash
# The GPU parity suite under the debug flavour (device-side bounds asserts, -O1 -g) and a subset under AddressSanitizer
# (gfx950:xnack+).  Build the flavours first (here or on the box): python -m speech2affective_gestures_amd.build --debug / --asan
# Writes gpurun_out/flavours/{debug,asan}.log + summary.txt (copied to profiles/r03_debug_asan_runs.txt).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/flavours; mkdir -p $O
cd $R
P=$R/speech2affective_gestures_amd
[ -f $P/libs2ag_hip_debug.so ] || python -m speech2affective_gestures_amd.build --debug > $O/build_debug.log 2>&1
[ "$WITH_ASAN" = "1" ] && { [ -f $P/libs2ag_hip_asan.so ] || python -m speech2affective_gestures_amd.build --asan > $O/build_asan.log 2>&1; }
{
echo "== debug flavour (S2AG_DBG_ASSERT on, -O1 -g): whole GPU suite =="
if [ -f $P/libs2ag_hip_debug.so ]; then
  S2AG_HIP_LIB=$P/libs2ag_hip_debug.so timeout 1500 python -m pytest tests -q -m gpu > $O/debug.log 2>&1
  echo "rc=$?  $(tail -1 $O/debug.log)"
else echo "libs2ag_hip_debug.so not built"; fi
echo "== asan flavour (-fsanitize=address, gfx950:xnack+, HSA_XNACK=1): kernel-level suites =="
if [ -f $P/libs2ag_hip_asan.so ]; then
  RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
  HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0 LD_PRELOAD=$RT \
    LD_LIBRARY_PATH=$(dirname $RT):/opt/rocm/lib/asan:$LD_LIBRARY_PATH S2AG_HIP_LIB=$P/libs2ag_hip_asan.so \
    timeout 1500 python -X faulthandler -m pytest tests/test_gpu_wave12.py tests/test_gpu_wave_fused.py tests/test_gpu_ops.py -q -m gpu -x > $O/asan.log 2>&1
  echo "rc=$?  $(tail -1 $O/asan.log)"
  grep -c "ERROR: AddressSanitizer" $O/asan.log | sed 's/^/AddressSanitizer reports: /'
else echo "libs2ag_hip_asan.so not built"; fi
} | tee $O/summary.txt
