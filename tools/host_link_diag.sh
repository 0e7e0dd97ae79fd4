#!/bin/bash
# What the host link of THIS box does: topology, where page-locked memory lands, each copy direction
# alone / both at once / kernels storing to host memory (tools/ubench/link_duplex.hip), and three mm_run_host calls.
# usage: tools/host_link_diag.sh <output dir>     (run on the GPU box)
out=${1:-gpurun_out/diag}
mkdir -p "$out"
{
  echo "== host"; nproc; uname -r; cat /proc/cmdline
  echo "== numa nodes"; ls /sys/devices/system/node 2>&1 | head; cat /sys/devices/system/node/online 2>&1
  for n in /sys/devices/system/node/node*; do echo "$n: cpus $(cat $n/cpulist 2>/dev/null)"; grep -E "MemTotal|MemFree" $n/meminfo 2>/dev/null; done
  echo "== gpu pci"
  for d in /sys/bus/pci/devices/*; do
    if [ "$(cat $d/vendor 2>/dev/null)" = "0x1002" ] && [ -e $d/current_link_speed ]; then
      echo "$d class $(cat $d/class) device $(cat $d/device) numa_node $(cat $d/numa_node 2>/dev/null) link $(cat $d/current_link_speed 2>/dev/null) x$(cat $d/current_link_width 2>/dev/null) max $(cat $d/max_link_speed 2>/dev/null) x$(cat $d/max_link_width 2>/dev/null) iommu_group $(basename $(readlink $d/iommu_group 2>/dev/null) 2>/dev/null)"
    fi
  done
  echo "== this shell: allowed cpus / mems"; grep -E "Cpus_allowed_list|Mems_allowed_list" /proc/self/status
  echo "== rocm-smi topo"; rocm-smi --showtoponuma 2>&1 | tail -8
} > "$out/box.txt" 2>&1
tools/ubench/link_duplex.bin 1024 64 > "$out/link.txt" 2>&1
python3 tools/gpu_host_trace.py > "$out/host_calls.txt" 2>&1
tail -3 "$out/host_calls.txt"
python3 tools/gpu_host_modes.py > "$out/host_modes.txt" 2>&1
cat "$out/host_modes.txt"
