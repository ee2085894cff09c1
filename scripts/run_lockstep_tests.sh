#!/bin/bash
# the -m gpu tests that go through lib.dynamic_eval_many / lockstep groups (a -k expression with spaces does not survive gpu_steps.sh's word splitting)
exec python -m pytest tests/test_model_gpu.py tests/test_harness_gpu.py -x -q -k "lockstep or many or chains or cross_dataset or seq_eval"
