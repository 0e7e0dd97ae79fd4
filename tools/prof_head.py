#!/usr/bin/env python3
"""Collects the round's profile evidence on the GPU box (run from the repo root, e.g. through gpurun):

    python3 tools/prof_head.py <tag> [headline] [C2 FWD C4 C5 READS ...]

headline: rocprofv3 --kernel-trace --stats of `bench.py --steps 20 --warmup 40`, then separate PMC passes
  (FETCH_SIZE, WRITE_SIZE, two SQ passes) of `bench.py --steps 3 --warmup 3`; writes
  gpurun_out/<tag>_summary.txt, gpurun_out/<tag>_bench_under_profiler.json and
  gpurun_out/head_counters.json (copy to profiles/: bench.py reads it for roofline.traffic / roofline.valu).
Cn: kernel trace + FETCH_SIZE + WRITE_SIZE passes of tools/run_config.py Cn -> gpurun_out/<tag>_<Cn>.txt.
This process never touches the GPU; rocprofv3 starts python3 directly (no shell / env hop)."""
import collections
import datetime
import glob
import hashlib
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
os.environ["TMPDIR"] = "/tmp"


def rocprof(name, pmc, cmd):
    d = os.path.join(OUT, "prof_" + name)
    subprocess.run(["rm", "-rf", d])
    args = ["rocprofv3", "--kernel-trace"]
    args += ["--pmc"] + pmc if pmc else ["--stats"]
    args += ["-d", d, "-o", "r", "--"] + cmd
    with open(os.path.join(OUT, name + ".log"), "w") as log:
        subprocess.run(args, stdout=log, stderr=subprocess.STDOUT, cwd=ROOT)
    dbs = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
    return dbs[0] if dbs else None


def kernel_durations(db, like="%fused_kernel%"):
    cur = sqlite3.connect(db).cursor()
    return [r[0] / 1e3 for r in cur.execute("select (end - start) from kernels where name like ? order by start", (like,))]


def top_kernels(db):
    cur = sqlite3.connect(db).cursor()
    try:
        return list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    except Exception:
        return []


def counters(db, like="fused_kernel"):
    cur = sqlite3.connect(db).cursor()
    acc = collections.defaultdict(list)
    meta = {}
    for k, c, v, d, g, wg, lds, vg, sg in cur.execute(
            "select kernel_name,counter_name,value,duration,grid_size,workgroup_size,lds_block_size,vgpr_count,sgpr_count "
            "from counters_collection"):
        if like in k:
            acc[c].append((v, d))
            meta = {"kernel": k, "grid": g, "wg": wg, "lds": lds, "vgpr": vg, "sgpr": sg}
    return {c: (sum(v for v, _ in vs) / len(vs), sum(d for _, d in vs) / len(vs) / 1e3, len(vs)) for c, vs in acc.items()}, meta


def src_sha():
    csrc = os.path.join(ROOT, "simd-minimizers_amd", "csrc")
    sys.path.insert(0, csrc)
    from strip_comments import strip  # (the hash is of the source without its comments, as in bench.py)
    h = hashlib.sha256()
    for f in ("mm_fused_impl.h", "mm_common.h"):
        h.update("\n".join(ln for ln in strip(open(os.path.join(csrc, f)).read()).split("\n") if ln.strip()).encode())  # (blank lines - stripped comment lines - do not count)
    return h.hexdigest()[:16]


def headline(tag):
    bench = ["python3", "bench.py", "--no-cpu-baseline", "--no-extra"]
    lines = []
    db = rocprof(tag + "_stats", None, bench + ["--steps", "20", "--warmup", "40"])
    lines.append(f"== rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 40 --no-cpu-baseline --no-extra\n")
    for r in top_kernels(db)[:8]:
        lines.append(f"{r[0][:90]:90s} calls={r[1]:4d} total_us={r[2]:12.1f} avg_us={r[3]:10.2f} pct={r[4]:5.1f}\n")
    d = kernel_durations(db)
    lines.append("-- fused_kernel launches in order, us: " + " ".join(f"{x:.0f}" for x in d) + "\n")
    if len(d) >= 20:
        # launches in order: 40 warm-up, 20 timed, then the steps of the clock probe (a second queue is active there)
        timed = d[40:60] if len(d) >= 60 else d[-20:]
        lines.append(f"-- average of launches 41..60 (the timed steps): {sum(timed) / len(timed):.1f} us\n")
    try:
        log = open(os.path.join(OUT, tag + "_stats.log")).read().strip().split("\n")
        js = [x for x in log if x.startswith("{")][-1]
        open(os.path.join(OUT, tag + "_bench_under_profiler.json"), "w").write(js + "\n")
    except Exception as e:
        lines.append(f"(no bench line: {e})\n")
    rec = {"kernel_source_sha": src_sha(), "collected": datetime.date.today().isoformat() + " " + tag,
           "command": "bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-extra under rocprofv3 --pmc, one counter group per pass"}
    short = bench + ["--steps", "3", "--warmup", "3"]
    passes = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"],
              "sq1": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU",
                      "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES"],
              "sq2": ["SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE", "SQ_BUSY_CU_CYCLES", "SQ_WAIT_INST_ANY",
                      "SQ_ACTIVE_INST_ANY", "SQ_INST_CYCLES_SALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]}
    for name, pmc in passes.items():
        db = rocprof(f"{tag}_{name}", pmc, short)
        if not db:
            lines.append(f"({name}: no database)\n")
            continue
        cs, meta = counters(db)
        lines.append(f"== pass {name}: {' '.join(pmc)}  ({meta})\n")
        for c, (v, dur, cnt) in sorted(cs.items()):
            lines.append(f"{c:28s} n={cnt} mean={v:.6g} kernel_us={dur:.1f}\n")
            rec[c] = v
            if c == "GRBM_GUI_ACTIVE":
                rec["GRBM_GUI_ACTIVE_per_xcd"] = v / 8.0
                rec["counter_pass_kernel_us"] = dur
    if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
        # gfx950: FETCH_SIZE counts half the bytes of a coalesced stream (MI355X_MICROARCH.md, HBM section); KB units
        rec["hbm_bytes_per_launch"] = int((2 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024)
    try:
        b = json.load(open(os.path.join(OUT, tag + "_bench_under_profiler.json")))
        rec["windows_per_launch"] = b["config"]["bases_per_gpu"] - (b["config"]["k"] + b["config"]["w"] - 1) + 1
    except Exception:
        rec["windows_per_launch"] = 3_100_000_000 - 30
    json.dump(rec, open(os.path.join(OUT, "head_counters.json"), "w"), indent=1)
    open(os.path.join(OUT, tag + "_summary.txt"), "w").writelines(lines)


def config(tag, cfg):
    cmd = ["python3", "tools/run_config.py", cfg, "5", "5"]
    lines = []
    db = rocprof(f"{tag}_{cfg}_stats", None, cmd)
    lines.append(f"== rocprofv3 --kernel-trace --stats -- python3 tools/run_config.py {cfg} 5 5\n")
    for r in top_kernels(db)[:6]:
        lines.append(f"{r[0][:90]:90s} calls={r[1]:4d} total_us={r[2]:12.1f} avg_us={r[3]:10.2f} pct={r[4]:5.1f}\n")
    d = kernel_durations(db)
    lines.append("-- fused_kernel launches in order, us: " + " ".join(f"{x:.0f}" for x in d) + "\n")
    try:
        log = open(os.path.join(OUT, f"{tag}_{cfg}_stats.log")).read().strip().split("\n")
        lines.append("-- run_config line (HIP events, under the profiler): " + [x for x in log if x.startswith("{")][-1] + "\n")
    except Exception:
        pass
    tot = {}
    for name, pmc in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"])):
        db = rocprof(f"{tag}_{cfg}_{name}", pmc, ["python3", "tools/run_config.py", cfg, "3", "2"])
        if not db:
            continue
        cs, meta = counters(db)
        for c, (v, dur, cnt) in cs.items():
            lines.append(f"{c:14s} n={cnt} mean_KB={v:.1f} kernel_us={dur:.1f} {meta}\n")
            tot[c] = v
    if len(tot) == 2:
        lines.append(f"-- HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = {int((2 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024)}\n")
    open(os.path.join(OUT, f"{tag}_{cfg}.txt"), "w").writelines(lines)


def stalls(tag, cfg, like="fused_kernel"):
    """What the waves of a configuration's kernel (default: the fused kernel) wait for: SQ wait / active counters,
    separate passes.  stalls:<CFG>[:<kernel name part>]"""
    cmd = ["python3", "tools/run_config.py", cfg, "3", "2"]
    passes = {"w1": ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS",
                     "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"],
              "w2": ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM",
                     "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"],
              "w3": ["SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR", "SQ_ACTIVE_INST_SCA", "SQ_IFETCH", "SQ_WAIT_IFETCH",
                     "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INST_LEVEL_VMEM"],
              "w4": ["TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA_RDREQ_sum",
                     "TCP_PENDING_STALL_CYCLES_sum"]}
    lines = [f"== SQ / cache counters of {like}, python3 tools/run_config.py {cfg} 3 2, one group per pass\n"]
    for name, pmc in passes.items():
        db = rocprof(f"{tag}_{cfg}_{name}", pmc, cmd)
        if not db:
            lines.append(f"({name}: no database; see {tag}_{cfg}_{name}.log)\n")
            continue
        cs, meta = counters(db, like)
        lines.append(f"-- pass {name} ({meta})\n")
        for c, (v, dur, cnt) in sorted(cs.items()):
            lines.append(f"{c:30s} n={cnt} mean={v:.6g} kernel_us={dur:.1f}\n")
    open(os.path.join(OUT, f"{tag}_{cfg}_stalls" + ("" if like == "fused_kernel" else "_" + like) + ".txt"), "w").writelines(lines)


def component(tag, cfg):
    """One of the rows either side of the path (simd_minimizers_amd.workloads): kernel trace of `run_config.py cfg 5 3`
    and FETCH_SIZE / WRITE_SIZE summed over the kernels of one step."""
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("wl", os.path.join(ROOT, "simd-minimizers_amd", "workloads.py"))
    likes = {"READS": ["fused_kernel"], "READS_SK": ["fused_kernel"], "SKIP": ["window_ambiguity_kernel", "fused_kernel"],
             "VALUES": ["values_u64_kernel"], "PACK": ["pack_ascii"], "FASTA": ["fasta"], "FASTQ": ["fastq"],
             # (round 6: the lane-table rows - the table's four kernels are part of the step)
             "READS_VAR": ["fused_kernel", "seg_"], "LONGREADS": ["fused_kernel", "seg_"], "BATCH10K": ["fused_kernel", "seg_"]}[cfg]
    lines = []
    db = rocprof(f"{tag}_{cfg}_stats", None, ["python3", "tools/run_config.py", cfg, "5", "3"])
    lines.append(f"== rocprofv3 --kernel-trace --stats -- python3 tools/run_config.py {cfg} 5 3\n")
    rows = top_kernels(db)
    mine = [r for r in rows if any(l in r[0] for l in likes) or "fillBuffer" in r[0]]  # (the row's own kernels first;
    for r in mine + [r for r in rows if r not in mine][:4]:                            # the rest is input generation)
        lines.append(f"{r[0][:90]:90s} calls={r[1]:4d} total_us={r[2]:12.1f} avg_us={r[3]:10.2f} pct={r[4]:5.1f}\n")
    try:
        log = open(os.path.join(OUT, f"{tag}_{cfg}_stats.log")).read().strip().split("\n")
        lines.append("-- run_config line (torch events, under the profiler): " + [x for x in log if x.startswith("{")][-1] + "\n")
    except Exception:
        pass
    steps = 3 + 2
    tot = {}
    for name, pmc in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"])):
        db = rocprof(f"{tag}_{cfg}_{name}", pmc, ["python3", "tools/run_config.py", cfg, "3", "2"])
        if not db:
            continue
        cur = sqlite3.connect(db).cursor()
        acc = collections.defaultdict(float)
        n = collections.defaultdict(int)
        per_kernel = collections.defaultdict(int)
        for k, c, v in cur.execute("select kernel_name,counter_name,value from counters_collection"):
            if any(l in k for l in likes):
                acc[c] += v
                n[c] += 1
                per_kernel[(c, k)] += 1
        for c in acc:
            # steps of this pass: the row warms up by TIME since round 5, so the step count is what was dispatched - every
            # kernel of the row runs once per step (set-up kernels that ran once or twice are not counted)
            counts = [m for (cc, _k), m in per_kernel.items() if cc == c and m >= 3]
            steps = min(counts) if counts else steps
            tot[c] = acc[c] / steps
            lines.append(f"{c:14s} {n[c]} dispatches of {likes} over {steps} steps: {tot[c]:.1f} KB per step\n")
    if len(tot) == 2:
        lines.append(f"-- HBM bytes per step = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = {int((2 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024)}"
                     "  (FETCH_SIZE doubled: the gfx950 correction for coalesced streams; compare with algorithmic_bytes above)\n")
    open(os.path.join(OUT, f"{tag}_{cfg}.txt"), "w").writelines(lines)


def cleanup():
    """The profiler's databases are large and gpurun copies back at most 64 MiB: keep only the condensed files."""
    for d in glob.glob(os.path.join(OUT, "prof_*")):
        subprocess.run(["rm", "-rf", d])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    import atexit
    atexit.register(cleanup)
    tag = sys.argv[1]
    for what in sys.argv[2:]:
        if what == "headline":
            headline(tag)
        elif what.startswith("stalls:"):
            stalls(tag, *what[7:].split(":"))
        elif what in ("READS", "READS_SK", "SKIP", "VALUES", "PACK", "FASTA", "FASTQ", "READS_VAR", "LONGREADS", "BATCH10K"):
            component(tag, what)
        else:
            config(tag, what)
