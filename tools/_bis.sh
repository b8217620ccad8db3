H='tests/test_gpu_modules.py::test_generator_train_mode_with_dropout_forward_and_all_gradients'
A='tests/test_gpu_step.py::test_three_steps_match_the_reference_trace'
T='tests/test_gpu_step.py::test_two_steps_with_dropout_match_the_oracle'
for e in "X=1" "S2AG_EARLY_REAL_BWD=0" "S2AG_ENCODERS_ASIDE=0" "S2AG_SHARE_ENCODERS=0" "S2AG_EARLY_RAND=0" "S2AG_FUSE_BWD=0" "S2AG_BN_EPILOGUE=0" "S2AG_TM_COPIES=0"; do echo "== $e"; env $e python -m pytest $H $A $T -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED.*two_steps"; done
