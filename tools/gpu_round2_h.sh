#!/bin/bash
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/h_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/h_tests.log
python tools/gpu_ab2.py 2>&1 | grep -v amdgpu.ids
for w in 38 48 53 55 56; do k=21; if [ $(( (k + w - 1) % 2 )) -eq 0 ]; then k=22; fi; python tools/gpu_jit_w.py $k $w 1 0 "" "-DMM_MIN_BLOCKS=2" "-DMM_MIN_BLOCKS=4" 2>&1 | grep -v amdgpu.ids; done
