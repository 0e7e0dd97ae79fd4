#!/bin/bash
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_round2.py -x -q -m gpu > gpurun_out/t_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/t_tests.log
FWD=1 python tools/gpu_caplimit.py 21:11,15:10,5:7,21:13 0 38 44 51 60 2>&1 | grep -v amdgpu.ids
python tools/gpu_ab2.py 2>&1 | grep -v amdgpu.ids | head -3
