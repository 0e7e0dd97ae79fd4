#!/bin/bash
# gpurun -- bash tools/ubench/run_concurrent.sh   (results: gpurun_out/concurrent.txt)
cd tools/ubench
o=../../gpurun_out/concurrent.txt
mkdir -p ../../gpurun_out
: > $o
run() { echo "== $*" >> $o; timeout 120 "$@" >> $o 2>&1; }
# canonical-like walkers: 39 KB LDS, 128 / 120 VGPRs, 32 KB dump, 52 KB out per tile
run ./concurrent_kernels_v127.bin 40000 1000 39408 8 52 256 1
run ./concurrent_kernels_v127.bin 40000 1000 39408 8 52 256 0
run ./concurrent_kernels_v127.bin 40000 1000 39408 8 52 128 1
run ./concurrent_kernels_v119.bin 40000 1000 39408 8 52 256 1
run ./concurrent_kernels_v119.bin 40000 1000 39408 8 52 512 1
run ./concurrent_kernels_v127_e64.bin 40000 1000 39408 8 52 1024 1
run ./concurrent_kernels_v119.bin 40000 1000 39408 8 52 256 0
# forward-like walkers: 15.6 KB LDS, 72 VGPRs, 12 KB dump, 40 KB out per tile
run ./concurrent_kernels_v71.bin 50000 430 15600 3 40 256 1
run ./concurrent_kernels_v71.bin 50000 430 15600 3 40 256 0
run ./concurrent_kernels_v71.bin 50000 430 15600 3 40 512 1
run ./concurrent_kernels_v71.bin 50000 430 15600 3 40 128 1
cat $o
