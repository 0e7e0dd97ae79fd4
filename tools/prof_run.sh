#!/bin/bash
# usage: tools/prof_run.sh <tag>   (on the GPU box, from the repo root)
# kernel trace + stats of the default bench, then HBM traffic counters in separate passes.
TAG=${1:-r01}
export TMPDIR=/tmp
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o r01 -- python3 bench.py --steps 20 --warmup 40 --no-cpu-baseline > gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_fetch -o r01 -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof_write -o r01 -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline > gpurun_out/prof_write.log 2>&1
python3 tools/prof_summary.py gpurun_out/${TAG}_summary.txt
tail -1 gpurun_out/prof_stats.log > gpurun_out/${TAG}_bench_under_profiler.json
