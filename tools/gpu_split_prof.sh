#!/bin/bash
# rocprofv3 kernel trace of the split path: per-kernel durations and overlap (gpurun_out/split_prof.txt)
export TMPDIR=/tmp
o=gpurun_out/split_prof.txt; : > $o
prof() {  # <label> <canon> <k> <w>   (environment as set by the caller)
  rm -rf /tmp/sp
  rocprofv3 --kernel-trace -d /tmp/sp -o t -- python3 tools/gpu_split_steps.py $2 $3 $4 10 > /tmp/sp.log 2>&1
  echo "== $1" >> $o
  python3 tools/trace_overlap.py /tmp/sp 2>&1 | grep -E "walk|expand|fused" >> $o
}
export MM_SPLIT=1
MM_SPLIT_E=384 MM_SPLIT_NO_EXPAND=1 prof "forward: walk + dump alone" 0 21 11
MM_SPLIT_E=384 MM_SPLIT_SERIAL=1 prof "forward E=384: expander after the walk" 0 21 11
MM_SPLIT_E=512 MM_SPLIT_SERIAL=1 prof "forward E=512: expander after the walk" 0 21 11
MM_SPLIT_E=1024 MM_SPLIT_SERIAL=1 prof "forward E=1024: expander after the walk" 0 21 11
MM_SPLIT_E=384 prof "forward E=384 overlapped" 0 21 11
MM_SPLIT_E=128 MM_SPLIT_NO_EXPAND=1 prof "canonical: walk + dump alone" 1 21 11
MM_SPLIT_E=256 MM_SPLIT_SERIAL=1 prof "canonical E=256: expander after the walk" 1 21 11
MM_SPLIT_E=512 MM_SPLIT_SERIAL=1 prof "canonical E=512: expander after the walk" 1 21 11
MM_SPLIT=0 prof "fused forward" 0 21 11
MM_SPLIT=0 prof "fused canonical" 1 21 11
cat $o
