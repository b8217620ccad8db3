#!/bin/bash
# TEST INFRASTRUCTURE: ONE execution of the bench's own sizes at HEAD on the CPU device model while no
# MI355X is reachable -- BASELINE configs[3] at B = 256 (WavEncoder + TextEncoderTCN fwd + bwd, strict gradients) and
# configs[1] at B = 128 and configs[4] at B = 64, T = 136 (the whole GAN step, strict gradients, replay audit).  Tens of minutes; run from a snapshot copy of
# the tree so that edits do not rebuild the model's library under it:
#   git archive HEAD | tar -x -C gpurun_out/emu_full && (cd gpurun_out/emu_full && bash tools/run_emu_fullsize.sh > ../emu_fullsize.log 2>&1 &)
# The outcome (pass / fail, the tests' own log lines, wall time) is copied to profiles/r05_emu_fullsize.txt.
cd "$(dirname "$0")/.."
export S2AG_EMU=1 S2AG_EMU_FULLSIZE=1
# (measured in r05 on 8 cores: configs[3] at B = 256 2.5 min, the step at B = 128 9.5 min; test_full_size_training_steps --
# capture + many steps -- ran for more than an hour and was stopped: one step of each size is what this script is for)
# TESTS="node-id node-id ..." selects; switches of the registry (S2AG_TCN32_PAIR=1 ...) may be exported around the call
TESTS=${TESTS:-"tests/test_gpu_fullsize.py::test_conv1d_roofline_run_gradients_match_the_oracle_strictly[256] tests/test_gpu_fullsize.py::test_full_size_step_matches_the_oracle[step] tests/test_gpu_fullsize.py::test_full_size_step_matches_the_oracle[long]"}
for t in $TESTS; do
  echo "=== $t  (started $(date -u +%H:%M:%S))"
  t0=$SECONDS
  timeout ${BUDGET:-14000} python -u -m pytest "$t" -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "Warning\|WeightNorm\|^$" | tail -25
  echo "    wall $((SECONDS - t0)) s"
done
echo "=== done $(date -u +%H:%M:%S)"
