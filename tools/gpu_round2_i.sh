#!/bin/bash
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch or full_size or large_device or window_range or sweep_minimizers" > gpurun_out/i_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/i_tests.log
for c in C4 C3 FWD C5 C2; do python tools/run_config.py $c 7 8 2>&1 | grep -v amdgpu.ids; done
MM_NO_ROUNDS=1 python tools/run_config.py C4 7 8 2>&1 | grep -v amdgpu.ids | sed 's/^/NO_ROUNDS /'
python tools/gpu_nblk2.py 31 51 0 27 29
