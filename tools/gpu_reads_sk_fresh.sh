#!/bin/bash
# READS_SK on a fresh lease: the slow state made to appear and disappear on ONE box.
# 1. the driver's command verbatim, first thing on the box (cold code-object cache);  2. the reads rows alone with a cold
# cache and the OLD warm-up protocol (compile time counted as warm-up), 3. cold cache, new protocol, 4. warm cache, old protocol.
cd "$(dirname "$0")/.."
row() { python3 -c "
import json,sys
for ln in sys.stdin:
    ln=ln.strip()
    if not ln.startswith('{'): continue
    r=json.loads(ln)
    if 'extra' in r:
        print('  headline', r['value'], 'Gbases/s')
        for e in r['extra']:
            if e.get('component') in ('READS','READS_SK'): print('  ', e['component'], e['ms'], 'ms')
    elif r.get('component') in ('READS','READS_SK'): print('  ', r['component'], r['ms'], 'ms')
"; }
echo "1. python3 bench.py --gpus 1 --steps 20 --warmup 5 (fresh box, cold cache, OLD warm-up protocol)"
MM_BENCH_OLD_WARMUP=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | row
echo "2. reads rows alone, COLD cache (MM_JIT_CACHE_DIR=/tmp/jc_a), OLD warm-up protocol"
MM_JIT_CACHE_DIR=/tmp/jc_a MM_BENCH_OLD_WARMUP=1 python3 tools/gpu_reads_rows.py READS READS_SK 2>/dev/null | row
echo "3. reads rows alone, COLD cache (MM_JIT_CACHE_DIR=/tmp/jc_b), NEW protocol (first step before the warm-up clock)"
MM_JIT_CACHE_DIR=/tmp/jc_b python3 tools/gpu_reads_rows.py READS READS_SK 2>/dev/null | row
echo "4. reads rows alone, WARM cache (/tmp/jc_a again), OLD protocol"
MM_JIT_CACHE_DIR=/tmp/jc_a MM_BENCH_OLD_WARMUP=1 python3 tools/gpu_reads_rows.py READS READS_SK 2>/dev/null | row
echo "5. READS_SK before READS, warm cache, new protocol"
MM_JIT_CACHE_DIR=/tmp/jc_a python3 tools/gpu_reads_rows.py READS_SK READS 2>/dev/null | row
echo "6. python3 bench.py --gpus 1 --steps 20 --warmup 5 again: NEW protocol, cold cache (MM_JIT_CACHE_DIR=/tmp/jc_c)"
MM_JIT_CACHE_DIR=/tmp/jc_c python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | row
