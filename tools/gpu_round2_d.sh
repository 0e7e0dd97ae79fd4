#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
MM_PIPE=0 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "known_answers or anchors or sweep or large_device or window_range or low_complexity or mixed_density or unaligned or capacity or reads_mode or batch" > gpurun_out/d_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/d_tests.log
MM_PIPE=0 python tools/gpu_ab2.py > gpurun_out/d_ab2.log 2>&1; cat gpurun_out/d_ab2.log
