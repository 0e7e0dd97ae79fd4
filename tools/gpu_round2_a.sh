#!/bin/bash
# First GPU pass of round 2: new tests, bench (all workloads, self-spawned ranks), HEAD counters, full suite.
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round2.py -x -q -m gpu > gpurun_out/a_tests_round2.log 2>&1
echo "round2 tests rc=$?" 
tail -5 gpurun_out/a_tests_round2.log
python bench.py > gpurun_out/a_bench_default.json 2> gpurun_out/a_bench_default.err
echo "bench rc=$?"; cat gpurun_out/a_bench_default.json
MM_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 3 --warmup 2 --no-cpu-baseline --no-extra --bases 400000000 > gpurun_out/a_bench_gloo2.json 2> gpurun_out/a_bench_gloo2.err
echo "gloo2 headline rc=$?"; cat gpurun_out/a_bench_gloo2.json
MM_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 3 --warmup 2 --no-cpu-baseline --workload contigs > gpurun_out/a_bench_gloo2_contigs.json 2> gpurun_out/a_bench_gloo2_contigs.err
echo "gloo2 contigs rc=$?"; cat gpurun_out/a_bench_gloo2_contigs.json
MM_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 3 --warmup 2 --no-cpu-baseline --workload strong > gpurun_out/a_bench_gloo2_strong.json 2> gpurun_out/a_bench_gloo2_strong.err
echo "gloo2 strong rc=$?"; cat gpurun_out/a_bench_gloo2_strong.json
python bench.py --workload contigs --no-cpu-baseline > gpurun_out/a_bench_contigs1.json 2> gpurun_out/a_bench_contigs1.err
echo "contigs n=1 rc=$?"; cat gpurun_out/a_bench_contigs1.json
python3 tools/prof_head.py r02a headline > gpurun_out/a_prof_head.log 2>&1
echo "prof rc=$?"; cat gpurun_out/r02a_summary.txt | head -40; cat gpurun_out/head_counters.json
python -m pytest tests -x -q -m gpu > gpurun_out/a_tests_all.log 2>&1
echo "all tests rc=$?"; tail -5 gpurun_out/a_tests_all.log
