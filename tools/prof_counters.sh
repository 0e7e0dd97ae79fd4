#!/bin/bash
# usage: tools/prof_counters.sh <tag> <bench args...>   (run on the GPU box from the repo root)
TAG=$1; shift
export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
P3="GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
i=0
P4="SQ_THREAD_CYCLES_VALU SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_BRANCH SQ_IFETCH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU2"
P5="SQ_CYCLES SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_VMEM_TA_CMD_FIFO_FULL"
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace -d gpurun_out/pmc_${TAG}_$i -o r -- python3 bench.py "$@" --no-cpu-baseline > gpurun_out/pmc_${TAG}_$i.log 2>&1
done
python3 - <<PY
import sqlite3,glob,collections
for p in sorted(glob.glob("gpurun_out/pmc_${TAG}_*/*.db")):
    db=sqlite3.connect(p);cur=db.cursor()
    acc=collections.defaultdict(list)
    try:
        for k,c,v,d in cur.execute("select kernel_name,counter_name,value,duration from counters_collection"):
            if 'fused_kernel' in k: acc[c].append((v,d))
    except Exception as e: print(p,e); continue
    for c,vs in acc.items():
        print(f"{c:28s} n={len(vs)} mean={sum(v for v,_ in vs)/len(vs):.4g} dur_us={sum(d for _,d in vs)/len(vs)/1e3:.1f}")
PY
