#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/validate_tests.log 2>&1
echo "all gpu tests rc=$?"; tail -5 gpurun_out/validate_tests.log
python bench.py > gpurun_out/validate_bench.json 2> gpurun_out/validate_bench.err; echo "bench rc=$?"; cat gpurun_out/validate_bench.json
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/validate_bench_driver_flags.json 2>/dev/null; cat gpurun_out/validate_bench_driver_flags.json | cut -c1-400
python3 tools/prof_head.py r02 headline C2 FWD C4 C5 READS > gpurun_out/validate_prof.log 2>&1; echo "prof rc=$?"
cat gpurun_out/head_counters.json | head -40
for c in C2 FWD C4 C5 READS; do echo "== $c"; cat gpurun_out/r02_$c.txt; done
