H='tests/test_gpu_modules.py::test_generator_train_mode_with_dropout_forward_and_all_gradients'
A='tests/test_gpu_step.py::test_three_steps_match_the_reference_trace'
T='tests/test_gpu_step.py::test_two_steps_with_dropout_match_the_oracle'
echo "== no overlap"; S2AG_TEST_OVERLAP=0 python -m pytest $H $A $T -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED"
echo "== serial launches"; AMD_SERIALIZE_KERNEL=3 python -m pytest $H $A $T -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED"
echo "== HIP_LAUNCH_BLOCKING"; HIP_LAUNCH_BLOCKING=1 python -m pytest $H $A $T -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED"
