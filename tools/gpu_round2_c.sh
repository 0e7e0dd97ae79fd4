#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "known_answers or anchors or sweep_minimizers or sweep_syncmers or large_device or window_range or low_complexity or unaligned or capacity or full_size" > gpurun_out/c_tests.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/c_tests.log
MM_PIPE=0 timeout 300 python tools/gpu_pipe_ab.py > gpurun_out/c_ab0.log 2>&1; cat gpurun_out/c_ab0.log
MM_PIPE=1 timeout 300 python tools/gpu_pipe_ab.py > gpurun_out/c_ab1.log 2>&1; cat gpurun_out/c_ab1.log
