#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round2.py -x -q -m gpu > gpurun_out/b_tests_round2.log 2>&1
echo "round2 tests rc=$?"; tail -4 gpurun_out/b_tests_round2.log
python tools/gpu_ab2.py > gpurun_out/b_ab2.log 2>&1; cat gpurun_out/b_ab2.log
