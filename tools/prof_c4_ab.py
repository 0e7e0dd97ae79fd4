#!/usr/bin/env python3
"""C4 (canonical k=31 w=51, the 24 contigs, one batch launch) re-fetches every sequence line about four
times (FETCH 3.8 x the packed input).  Two A/Bs with counters, each a rocprofv3 kernel trace + FETCH_SIZE + WRITE_SIZE pass of
tools/run_config.py C4:  (1) lane length - blocks per lane 27 (default) .. 7: shorter lanes put neighbouring lanes' spans into
shared lines;  (2) resident lanes - MM_LDS_PAD pushes the workgroups per CU from 3 to 2, i.e. 16 384 instead of 24 576 lanes
per XCD behind its 4 MB of L2.  Run from the repo root on the GPU box; writes gpurun_out/r06_c4_ab.txt."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import prof_head
rows = []
for label, env in (("default lanes (27 blocks, tapered)", {}), ("22 blocks per lane", {"MM_RUN_NBLK": "22"}), ("18 blocks", {"MM_RUN_NBLK": "18"}),
                   ("14 blocks", {"MM_RUN_NBLK": "14"}), ("10 blocks", {"MM_RUN_NBLK": "10"}), ("7 blocks", {"MM_RUN_NBLK": "7"}),
                   ("27 blocks pinned (uniform tiles)", {"MM_RUN_NBLK": "27"}),
                   ("default lanes, 2 workgroups per CU (MM_LDS_PAD=24000)", {"MM_LDS_PAD": "24000"})):
    for k in ("MM_RUN_NBLK", "MM_LDS_PAD"):
        os.environ.pop(k, None)
    os.environ.update(env)
    tag = "r06ab_" + re.sub(r"[^0-9a-z]+", "_", label.lower())[:24]
    prof_head.config(tag, "C4")
    txt = open(os.path.join(prof_head.OUT, f"{tag}_C4.txt")).read()
    ms = re.search(r'"kernel_ms_median": ([0-9.]+)', txt)
    avg = re.search(r"fused_kernel.*?avg_us=\s*([0-9.]+)", txt)
    fetch = re.search(r"FETCH_SIZE\s+n=\d+ mean_KB=([0-9.]+)", txt)
    write = re.search(r"WRITE_SIZE\s+n=\d+ mean_KB=([0-9.]+)", txt)
    hbm = re.search(r"= (\d+)\n", txt.split("HBM bytes per launch")[-1]) if "HBM bytes per launch" in txt else None
    rows.append(f"{label:58s} kernel {float(ms.group(1)) if ms else -1:7.3f} ms (events) {float(avg.group(1)) if avg else -1:8.1f} us (trace avg)  "
                f"FETCH {float(fetch.group(1)) / 1024 if fetch else -1:8.1f} MB  WRITE {float(write.group(1)) / 1024 if write else -1:8.1f} MB  "
                f"HBM bytes {int(hbm.group(1)) / 1e9 if hbm else -1:6.3f} GB")
    print(rows[-1], flush=True)
open(os.path.join(prof_head.OUT, "r06_c4_ab.txt"), "w").write(__doc__ + "\nalgorithmic bytes of the launch: 1.259 GB (779 MB packed input + 480 MB positions)\n\n" + "\n".join(rows) + "\n")
