#!/bin/bash
export TMPDIR=/tmp
python tools/gpu_jit_w.py 21 11 1 0 "" "-DMM_PF=2" "-DMM_PF=4" "-DMM_PF=6" 2>&1 | grep -v amdgpu.ids
MM_CAP_LIMIT=60 python tools/gpu_jit_w.py 21 11 1 0 "-DMM_MIN_BLOCKS=5 -DMM_PF=2" "-DMM_MIN_BLOCKS=5 -DMM_PF=1" "-DMM_MIN_BLOCKS=5 -DMM_PF=4" 2>&1 | grep -v amdgpu.ids | sed 's/^/CAP60 /'
python tools/gpu_jit_w.py 21 11 0 0 "" "-DMM_PF=2" "-DMM_PF=4" 2>&1 | grep -v amdgpu.ids
python tools/gpu_jit_w.py 15 17 1 1 "" "-DMM_PF=2" "-DMM_PF=4" 2>&1 | grep -v amdgpu.ids
python tools/gpu_jit_w.py 19 19 1 0 "" "-DMM_PF=2" "-DMM_PF=4" 2>&1 | grep -v amdgpu.ids
python tools/gpu_jit_w.py 21 25 1 0 "" "-DMM_PF=2" "-DMM_PF=1" 2>&1 | grep -v amdgpu.ids
