#!/bin/bash
export TMPDIR=/tmp
for w in 29 31; do python tools/gpu_jit_w.py 21 $w 1 0 "" "-DMM_PF=2" "-DMM_PF=1" 2>&1 | grep -v amdgpu.ids; done
for w in 35 37 39; do python tools/gpu_jit_w.py 21 $w 1 0 "" "-DMM_MIN_BLOCKS=4 -DMM_PF=2" "-DMM_MIN_BLOCKS=3 -DMM_PF=2" "-DMM_MIN_BLOCKS=4 -DMM_PF=1" 2>&1 | grep -v amdgpu.ids; done
for w in 45 47; do python tools/gpu_jit_w.py 21 $w 1 0 "" "-DMM_MIN_BLOCKS=3 -DMM_PF=2" "-DMM_MIN_BLOCKS=3 -DMM_PF=1" 2>&1 | grep -v amdgpu.ids; done
for w in 57 63; do python tools/gpu_jit_w.py 21 $w 1 0 "" "-DMM_MIN_BLOCKS=3 -DMM_PF=1" "-DMM_MIN_BLOCKS=2 -DMM_PF=1" "-DMM_MIN_BLOCKS=2 -DMM_PF=2" 2>&1 | grep -v amdgpu.ids; done
python tools/gpu_jit_w.py 16 75 1 0 "" "-DMM_MIN_BLOCKS=2 -DMM_PF=1" "-DMM_MIN_BLOCKS=2 -DMM_PF=2" 2>&1 | grep -v amdgpu.ids
python tools/gpu_jit_w.py 16 100 1 0 "" "-DMM_MIN_BLOCKS=2 -DMM_PF=1" "-DMM_PF=1" 2>&1 | grep -v amdgpu.ids
python tools/gpu_jit_w.py 21 33 0 0 "" "-DMM_PF=2" "-DMM_PF=1" 2>&1 | grep -v amdgpu.ids
python tools/gpu_jit_w.py 21 51 0 0 "" "-DMM_PF=2" "-DMM_PF=1" 2>&1 | grep -v amdgpu.ids
