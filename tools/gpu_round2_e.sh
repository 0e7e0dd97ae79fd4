#!/bin/bash
export TMPDIR=/tmp
for M in 0 2 3 4 8; do
  MM_PIPE=1 MM_PIPE_TILES=$M timeout 300 python tools/gpu_pipe_ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/M=$M /" | head -3
done
MM_PIPE=0 timeout 300 python tools/gpu_pipe_ab.py 2>&1 | grep -v amdgpu.ids | head -3
