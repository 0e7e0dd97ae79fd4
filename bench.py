#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X minimizer engine.

Metric (BASELINE.json): Gbases/s (whole node) for canonical minimizers k=21 w=11 on a 3.1 Gbp
PackedSeq, with the achieved HBM GB/s against the MI355X roofline.

A "step" is one pass of the hot path (one fused-kernel launch: packed-seq decode -> ntHash ->
sliding min -> strand vote -> dedup/collect) over synthetic input that is already resident in HBM.

    python bench.py --gpus N --steps K --warmup W [--workload headline|contigs|strong]

  strong    ONE 3.1 Gbp sequence, canonical k=21 w=11 (BASELINE config 3), cut into N window ranges (absolute
            positions, exact seam, no data-path collective): `value` = 3.1e9 x steps / max-over-ranks time.
            THE DEFAULT: north_star's experiment ("throughput on a synthetic 3.1 Gbp packed sequence ... at 1, 2,
            4 and 8 GPUs").  At N = 1 this IS the headline configuration (one range = the whole sequence, the
            same call, `scaling` "weak" by the contract's definition is moot for one GPU and reads "strong").
  headline  every rank owns one 3.1 Gbp sequence of its own (independent genomes: weak scaling, linear by
            construction).  At N > 1 its figure rides in `extra` of the default line.
  contigs   BASELINE config 4: 24 CHM13-like contigs, canonical k=31 w=51, placed on the N ranks
            greedily longest first, ONE batch launch per rank and step; afterwards the position
            buffers are gathered to rank 0 (RCCL over xGMI), timed separately (strong scaling).  At N > 1 its
            figure rides in `extra` of the default line.

With N > 1 and no launcher in the environment the script starts its N ranks itself (child processes,
before anything in this process touches a GPU); under torchrun it uses the ranks it is given.  The
timed region follows the driver contract: W untimed steps, barrier + synchronize, exactly K steps,
synchronize + barrier, max over ranks.  Rank 0 prints ONE JSON line.  `vs_baseline` stays null: BASELINE.md
holds no published number for this metric on this hardware (the CPU figure measured beside it is
`cpu_baseline`, a reported baseline and not a target).
"""
import argparse
import hashlib
import json
import os
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
N_SIMD = 1024           # 256 CUs x 4 SIMDs
K, W = 21, 11
N_BASES = 3_100_000_000
SEED = 3
CPU_SAMPLE_CHUNK = 256 * 1024 * 1024
CPU_SAMPLE_SECONDS = 12.0
CPU_SAMPLE_MAX_CHUNKS = 12
METRIC = "Gbases/s (whole node) for canonical minimizers k=21 w=11 on 3.1 Gbp; HBM GB/s %peak"


# --------------------------------------------------------------------------- workload plan (pure: tests/test_distributed.py)
def resolve_workload(workload, gpus):
    """`--workload` left to its default: the strong split of ONE 3.1 Gbp sequence (at N = 1 that is the headline
    call itself, reported under its historical name)."""
    if workload:
        return workload
    return "headline" if gpus == 1 else "strong"


def strong_plan(n_bases, world, k=K, w=W):
    """Window range of every rank of the strong split and the bases the whole job covers: the ranges tile the
    windows of ONE sequence exactly, so the bases sum to n_bases whatever N is."""
    from simd_minimizers_amd import sharding
    nw = n_bases - (k + w - 1) + 1
    ranges = sharding.shard_windows(nw, world)
    assert ranges[0][0] == 0 and ranges[-1][1] == nw and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    return {"ranges": ranges, "windows": nw, "total_bases": n_bases, "scaling": "strong"}


# --------------------------------------------------------------------------- launcher
def spawn_ranks(args) -> int:
    """`bench.py --gpus N` without a launcher: start N ranks as child processes of THIS process, which
    never touches a GPU itself (no re-exec of a process that has initialised HIP).  Rank 0's stdout
    (the JSON line) is passed through."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus),
                    "LOCAL_WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0]
    rc = procs[0].returncode
    for p in procs[1:]:
        try:
            p.wait(timeout=120 if rc == 0 else 5)
        except subprocess.TimeoutExpired:
            p.kill()  # this exact child, by pid
            p.wait()
        rc = rc or p.returncode
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return rc


# --------------------------------------------------------------------------- CPU baseline
def cpu_baseline():
    """The oracle's one-pass port of the reference algorithm (oracle/mm_oracle.c: mmo_run_fast,
    two-stacks + ntHash, eight AVX2 lanes per thread when the host has them like the reference's
    SIMD path, window ranges spread over the host cores like the reference's rayon-over-contigs
    benchmark) timed on this box's host cores on a bounded sample of the same
    workload.  Checker/baseline use only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctypes as C
    import tempfile

    import numpy as np

    import mm_oracle as o

    try:
        lib = o.lib(o.build(native=True, out_dir=tempfile.mkdtemp(prefix="mm_oracle_")))
    except Exception:
        lib = o.lib()
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, int(os.environ.get("MM_CPU_THREADS", "128"))))
    quota = None
    try:  # a container may be allowed fewer CPUs than it sees (cgroup v2 cpu.max: quota period)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(per)
    except Exception:
        pass
    h = o.default_hasher(True)
    cap = int(CPU_SAMPLE_CHUNK * 2.3 / (W + 1)) + 4096
    pos = np.ones(cap, dtype=np.uint32)  # touched, so page faults stay out of the timed region
    total_t, total_n, chunks = 0.0, 0, 0
    # one untimed call first: the port keeps its per-thread output slots between calls, like the
    # reference's reusable buffers, so the timed calls do not pay for first-touch page faults
    g0 = o.gen_packed(SEED, CPU_SAMPLE_CHUNK)
    lib.mmo_run_fast(g0.ctypes.data_as(C.POINTER(C.c_uint8)), 0, CPU_SAMPLE_CHUNK, K, W, C.byref(h), 1, threads,
                     pos.ctypes.data_as(C.POINTER(C.c_uint32)), cap)
    while total_t < CPU_SAMPLE_SECONDS and chunks < CPU_SAMPLE_MAX_CHUNKS:
        g = o.gen_packed(SEED, CPU_SAMPLE_CHUNK, first_base=chunks * CPU_SAMPLE_CHUNK)
        t0 = time.perf_counter()
        r = lib.mmo_run_fast(g.ctypes.data_as(C.POINTER(C.c_uint8)), 0, CPU_SAMPLE_CHUNK, K, W, C.byref(h),
                             1, threads, pos.ctypes.data_as(C.POINTER(C.c_uint32)), cap)
        total_t += time.perf_counter() - t0
        assert r > 0
        total_n += CPU_SAMPLE_CHUNK
        chunks += 1
    # the same code on one thread (32 Mbp), and the CPU model, for the record
    one_n = 32 * 1024 * 1024
    t0 = time.perf_counter()
    lib.mmo_run_fast(g0.ctypes.data_as(C.POINTER(C.c_uint8)), 0, one_n, K, W, C.byref(h), 1, 1,
                     pos.ctypes.data_as(C.POINTER(C.c_uint32)), cap)
    one_thread = one_n / (time.perf_counter() - t0) / 1e9
    value = total_n / total_t / 1e9
    flavour = "eight-lane AVX2" if lib.mmo_fast_lanes() == 8 else "scalar"
    model = "unknown CPU"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    limit = ""
    if quota is not None:
        # which limit applied: a quota that throttled would cap the speed-up near the quota
        limit = (f"; cgroup cpu.max reads {quota:.0f} CPUs but the {threads} threads ran {value / one_thread:.0f}x "
                 f"faster than one, so the quota {'did not throttle' if value / one_thread > 1.5 * quota else 'throttled'} this run")
    return {
        "value": round(value, 5), "unit": "Gbases/s", "cores": threads, "kind": "port",
        "sample": f"{chunks} x {CPU_SAMPLE_CHUNK} bases of the same generator (seed {SEED}), canonical "
                  f"k={K} w={W}; {flavour} two-stacks + ntHash port of the reference (oracle/mm_oracle.c "
                  f"mmo_run_fast, gcc -O3 -march=native), window ranges over {threads} threads "
                  f"(sched_getaffinity: {avail} CPUs) on {model}{limit}; one thread: {one_thread:.3f} Gbases/s; the "
                  f"reference's own published figure (unstated x86 AVX2, 1 thread, not measured here) "
                  f"is 0.455 Gbases/s",
    }


# --------------------------------------------------------------------------- recorded counters
def kernel_source_sha():
    """Hash of the kernel source WITHOUT its comments and blank lines (csrc/strip_comments.py):
    a reworded comment does not make the recorded counters stale, a changed instruction does."""
    csrc = os.path.join(ROOT, "simd-minimizers_amd", "csrc")
    sys.path.insert(0, csrc)
    from strip_comments import strip
    h = hashlib.sha256()
    for f in ("mm_fused_impl.h", "mm_common.h"):
        h.update("\n".join(ln for ln in strip(open(os.path.join(csrc, f)).read()).split("\n") if ln.strip()).encode())  # (blank lines - stripped comment lines - do not count)
    return h.hexdigest()[:16]


def recorded_counters(kernel_ms, live_clock_ghz=None):
    """HBM traffic and SQ counters cannot be collected from inside this process: rocprofv3 gathers them
    in separate passes around the same command (tools/prof_head.py), and the condensed result is
    committed as profiles/head_counters.json together with the hash of the kernel source it was
    measured on; the static instruction census of the kernel's main loop (tools/isa_census.py, hipcc -S
    on the build box) is profiles/head_isa_census.json.  Returns (traffic_bytes_per_launch or None,
    valu dict or None, provenance text).

    roofline.valu - the bound that actually binds: VALU-pipe time of one launch / kernel time, where
    VALU-pipe time = (wave-instructions issued, PMC SQ_INSTS_VALU: a property of the kernel and its input, not of
    the run) x (shader cycles per instruction: the census of the main loop priced at the issue rates
    tools/ubench/valu_rate.hip measures in shader cycles, profiles/r03_valu_issue_rates.txt) / (1024 SIMDs x the
    shader clock sampled LIVE beside a few extra steps, mm_clock_probe_*).  `frac` prices the instructions at the
    architectural 2 / 4 cycles per wave64 instruction - a bound that cannot be exceeded; `frac_vs_pure_streams` at the
    rates pure streams of one class reach in tools/ubench/valu_rate.hip (2.31 / 4.14), which a mixed stream can beat
    by a few per cent."""
    sha = kernel_source_sha()
    try:
        c = json.load(open(os.path.join(ROOT, "profiles", "head_counters.json")))
    except Exception:
        return None, None, "no profiles/head_counters.json"
    stale = c.get("kernel_source_sha") != sha
    traffic = c.get("hbm_bytes_per_launch")
    valu = None
    try:
        isa = json.load(open(os.path.join(ROOT, "profiles", "head_isa_census.json")))
        insts = float(c["SQ_INSTS_VALU"])
        windows = float(c["windows_per_launch"])
        pass_clk_hz = float(c["GRBM_GUI_ACTIVE_per_xcd"]) / (float(c["counter_pass_kernel_us"]) * 1e-6)
        clk_hz = live_clock_ghz * 1e9 if live_clock_ghz else pass_clk_hz
        cyc = float(isa["issue_clk_per_valu"])
        ideal = float(isa.get("ideal_clk_per_valu", cyc))
        cycles_per_simd = N_SIMD * clk_hz * kernel_ms * 1e-3  # SIMD cycles the kernel had
        valu = {"insts_per_window": round(insts * 64.0 / windows, 2),
                "main_loop_insts_per_window": isa["valu_per_window"],
                "cycles_per_inst": round(cyc, 3), "cycles_per_inst_ideal": round(ideal, 3),
                "sclk_mhz": round(clk_hz / 1e6, 0),
                "clock_source": ("live: shader cycle counter against the 100 MHz real-time counter, sampled by sleeping "
                                 "waves beside the timed loop (mm_clock_probe_*)") if live_clock_ghz else
                                "GRBM_GUI_ACTIVE / kernel time of the recorded counter pass (no live probe in this run)",
                # the bound: every VALU instruction at its architectural issue cost (2 / 4 shader cycles per wave64
                # instruction): cannot exceed 1
                "frac": round(insts * ideal / cycles_per_simd, 4),
                # the same priced at what a PURE stream of one instruction class reaches on this chip (2.31 / 4.14
                # cycles): a mixed stream issues a little better than its classes alone, so this may read just above 1
                "frac_vs_pure_streams": round(insts * cyc / cycles_per_simd, 4),
                "source": "SQ_INSTS_VALU recorded (profiles/head_counters.json), shader cycles per instruction from the "
                          "main loop's census (profiles/head_isa_census.json) at the measured issue rates "
                          "(profiles/r03_valu_issue_rates.txt: 2.31 / 4.14 cycles per full- / half-rate wave64 "
                          "instruction) for frac_vs_pure_streams and at the architectural 2 / 4 cycles for frac; kernel "
                          "time and shader clock live"
                          + ("; STALE counters: kernel source changed since" if stale else "")
                          + ("; STALE census" if isa.get("kernel_source_sha") != sha else "")}
    except Exception:
        pass
    prov = ("recorded by rocprofv3 PMC passes (profiles/head_counters.json: " + c.get("collected", "?") + ")"
            + ("; STALE: the kernel source changed since" if stale else ""))
    return traffic, valu, prov


# --------------------------------------------------------------------------- one process, several devices
def single_process(args):
    """`--single-process --gpus N`: the several-device entry points a C / Rust caller uses, driven from THIS process
    through the C ABI's device group (no torch.distributed).  `value`: mm_run_sharded_device - the sequence resident on
    the devices (mm_device_group_upload_range: every device its share, once, untimed), one asynchronous launch per device over its window range, the
    positions left on the devices - with the device-to-device gather (mm_device_group_gather) timed
    beside it; `extra`: the host-buffer call mm_run_sharded_host (H2D + kernel + D2H per shard: PCIe-bound)."""
    import numpy as np
    import torch

    import simd_minimizers_amd as sm
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        sys.exit("bench.py --single-process: no GPU")
    # fewer devices than --gpus: entries share devices (a functional check, labelled as such)
    devices = [i % n_dev for i in range(args.gpus)]
    n = min(args.bases, N_BASES)
    ws = sm.default_workspace(0)
    d = sm.generate_device(n, SEED)
    torch.cuda.synchronize()
    hp, hp_owner = sm.pinned_array(((n + 3) // 4 + 64,), np.uint8)
    hp[:] = d.cpu().numpy()
    g = sm.DeviceGroup(devices)
    b = sm.canonical_minimizers(K, W)
    g.upload_range(hp[: (n + 3) // 4 + 1], n)  # (every device only its share of the split + halo)
    for _ in range(max(3, args.warmup)):
        counts = g.run_device(b, n)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        counts = g.run_device(b, n)
    dt = (time.perf_counter() - t0) / args.steps
    total = sum(counts)
    dst = torch.empty(total + 1024, dtype=torch.int32, device="cuda:%d" % devices[0])
    gms = []
    for _ in range(4):
        torch.cuda.synchronize()
        tg = time.perf_counter()
        got = g.gather(0, dst)
        gms.append((time.perf_counter() - tg) * 1e3)
    assert got == total
    # the shards laid end to end must be the one-device result
    out1 = torch.empty(total + 1024, dtype=torch.int32, device="cuda:0")
    c1 = b.workspace(ws).run_device(d, n, out1)
    assert c1 == total and bool(torch.equal(out1[:c1].cpu(), dst[:total].cpu())), "sharded result differs from the one-device result"
    del out1, dst, d
    torch.cuda.empty_cache()
    extra = []
    try:  # BASELINE config 4's shape through the same group: the 24 contigs resident on the entries, one batch launch each
        from simd_minimizers_amd import sharding
        lens_c = list(sharding.CHM13_CONTIG_LENGTHS)
        hosts = []
        for i, m in enumerate(lens_c):
            t = sm.generate_device(m, sharding.CHM13_CONTIG_SEED0 + i)
            hosts.append(t.cpu().numpy()[: (m + 3) // 4 + 1].copy())
            del t
        g.upload_batch(hosts)
        del hosts
        bc = sm.canonical_minimizers(31, 51)
        for _ in range(3):
            cc = g.run_batch_device(bc, lens_c)
        tb = time.perf_counter()
        reps = max(3, args.steps // 2)
        for _ in range(reps):
            cc = g.run_batch_device(bc, lens_c)
        dtb = (time.perf_counter() - tb) / reps
        dstc = torch.empty(sum(cc) + 1024, dtype=torch.int32, device="cuda:%d" % devices[0])
        torch.cuda.synchronize()
        tg = time.perf_counter()
        offs_c = g.gather_batch(0, dstc)
        gb_ms = (time.perf_counter() - tg) * 1e3
        extra.append({"config": f"C4 through mm_run_batch_sharded_device: 24 CHM13-like contigs resident on {args.gpus} entries "
                                "(greedy placement), canonical k=31 w=51, one batch launch per entry, positions left on the devices",
                      "ms_per_step": round(dtb * 1e3, 4), "Gbases_per_s": round(sum(lens_c) / dtb / 1e9, 1),
                      "positions": int(offs_c[-1]), "gather_ms": round(gb_ms, 3)})
        del dstc
        torch.cuda.empty_cache()
    except Exception as e:
        extra.append({"config": "mm_run_batch_sharded_device", "error": str(e)[:200]})
    try:  # the host-buffer call, PCIe-inclusive (at most 1 Gbp: host memory)
        nh = min(n, 1 << 30)
        cap = int(nh * 2.3 / (W + 1)) + 4096
        ho, ho_owner = sm.pinned_array((cap,), np.uint32)
        ho[:] = 0
        g.run(b, hp, nh, out=ho)
        th = time.perf_counter()
        reps = 3
        for _ in range(reps):
            pos, _ = g.run(b, hp, nh, out=ho)
        dth = (time.perf_counter() - th) / reps
        extra.append({"config": f"mm_run_sharded_host: ONE {nh} bp PackedSeq in page-locked host memory, H2D + kernel + D2H per "
                                "shard, one dense host result (PCIe-bound)", "ms_per_step": round(dth * 1e3, 3),
                      "Gbases_per_s": round(nh / dth / 1e9, 2), "outputs": int(len(pos))})
    except Exception as e:
        extra.append({"config": "mm_run_sharded_host", "error": str(e)[:200]})
    print(json.dumps({
        "metric": "Gbases/s, canonical minimizers k=21 w=11 through mm_run_sharded_device (device-resident shards, one process)",
        "value": round(n / dt / 1e9, 3), "unit": "Gbases/s", "n_gpus": args.gpus, "steps": args.steps,
        "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"ONE {n} bp PackedSeq, every device holding its share (mm_device_group_upload_range, untimed), cut "
                               f"into {args.gpus} window ranges by mm_run_sharded_device: one asynchronous launch per entry from "
                               "one host thread, positions left on the devices, absolute, exact seam",
                   "devices": devices, "distinct_devices": len(set(devices)), "outputs": int(total),
                   "outputs_per_entry": counts, "parallelism": f"device_group{args.gpus}",
                   "gather_ms": round(statistics.median(gms), 3),
                   "gather": "mm_device_group_gather: device-to-device copies of the shards into one buffer on entry 0's "
                             "device (hipMemcpyPeerAsync; xGMI between distinct GPUs), median of 4"},
        "extra": extra}), flush=True)
    return 0


# --------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=20,
                    help="untimed steps; the GPU needs about a dozen 2 ms launches after idle to reach steady clocks")
    ap.add_argument("--workload", choices=["headline", "contigs", "strong"], default=None,
                    help="default: the strong split of ONE 3.1 Gbp sequence (at --gpus 1: the headline configuration)")
    ap.add_argument("--bases", type=int, default=N_BASES, help="bases per GPU (headline) / in total (strong)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the untimed secondary configurations, the median-of-5 and the end-to-end figure")
    ap.add_argument("--single-process", action="store_true",
                    help="N devices driven from THIS process through the C ABI's device group (mm_run_sharded_host: "
                         "host buffers in, one dense host result out, one host thread per device, no torch.distributed). "
                         "PCIe-inclusive by construction: a separate line, never the headline `value`")
    args = ap.parse_args()
    args.workload = resolve_workload(args.workload, args.gpus)
    if args.single_process:
        sys.exit(single_process(args))

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(spawn_ranks(args))  # before torch / HIP are even imported
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    import torch
    import torch.distributed as dist

    # Libraries print banners to stdout (RCCL does when its communicator comes up); the contract is ONE
    # JSON line on stdout, so everything before it goes to stderr.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    distributed = world > 1
    # MM_BENCH_BACKEND=gloo lets the multi-rank control flow be exercised on a box with fewer GPUs
    # than ranks (ranks then share devices; a functional check, not a measurement)
    backend = os.environ.get("MM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    elif distributed and torch.cuda.device_count() < world:
        sys.exit(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPUs visible")
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=backend)
    dev = torch.device(f"cuda:{local_rank}")
    rdev = dev if backend == "nccl" else torch.device("cpu")

    import simd_minimizers_amd as sm
    from simd_minimizers_amd import sharding

    L = sm.lib()
    stream = torch.cuda.current_stream(dev)
    ws = sm.Workspace(local_rank, stream.cuda_stream)

    def generate(n, seed):
        t = torch.zeros((n + 3) // 4 + 64, dtype=torch.uint8, device=dev)
        sm._check(L.mm_generate_device_async(ws.h, seed, 0, n, t.data_ptr()))
        return t

    def timed_kernel_ms(step, repeats=5):
        """median of `repeats` single steps, each bracketed by HIP events on the workspace stream
        (bench/src/bin/paper.rs:536-556: warm-up + 5 repeats, median)"""
        ws.enable_timing(True)
        ws.kernel_time(True)
        ms = []
        for _ in range(repeats):
            step()
            torch.cuda.synchronize(dev)
            t, n = ws.kernel_time(True)
            ms.append(t / max(1, n))
        ws.enable_timing(False)
        return statistics.median(ms), ms

    def link_trace(where):
        """MM_BENCH_LINK_TRACE=1 (diagnostics): the link's both-directions rate at this point of the run, to stderr -
        the host-to-host call took 39.5 ms in a fresh process and 71 ms at the end of this script on the same box."""
        if not os.environ.get("MM_BENCH_LINK_TRACE") or rank != 0:
            return
        import ctypes as C_
        import numpy as np_
        nb_ = 256 << 20
        a_, ao_ = sm.pinned_array((nb_,), np_.uint8)
        b_, bo_ = sm.pinned_array((nb_,), np_.uint8)
        a_[:] = 1
        b_[:] = 2
        r_ = (C_.c_double * 3)()
        sm._check(L.mm_link_probe(ws.h, C_.c_void_p(a_.ctypes.data), C_.c_void_p(b_.ctypes.data), nb_, r_))
        print(f"[link] {where}: h2d {r_[0]:.1f} d2h {r_[1]:.1f} both {r_[2]:.1f} GB/s", file=sys.stderr, flush=True)
        del a_, b_, ao_, bo_

    link_trace("start")
    extras = []
    if rank == 0 and world == 1 and not args.no_extra and args.workload == "headline" and args.bases == N_BASES and not os.environ.get("MM_BENCH_SKIP_SECONDARY"):
        # Secondary configurations of BASELINE.json (untimed region, before the headline so that they
        # also bring the clocks up): kernel time by HIP events, median of 5 after 12 warm-up steps.
        def secondary(name, builder, n, seed, density, contigs=None, super_kmers=False):
            b = builder.workspace(ws)
            sk = None
            if contigs is None:
                d = generate(n, seed)
                out = torch.empty(int(n * density * 1.15) + 4096, dtype=torch.int32, device=dev)
                sk = torch.empty_like(out) if super_kmers else None
                cnt = torch.zeros(1, dtype=torch.int64, device=dev)

                def step():
                    b.run_device(d, n, out, sync=False, d_count=cnt, out_sk=sk)
            else:
                d = [generate(m, sharding.CHM13_CONTIG_SEED0 + i) for i, m in enumerate(contigs)]
                n = sum(contigs)
                out = torch.empty(int(n * density * 1.15) + 4096, dtype=torch.int32, device=dev)
                offs = [0]

                def step():
                    offs[:] = sm.run_batch_device(b, d, list(contigs), out)
            # warm-up by TIME, not by count: the chip needs some tens of milliseconds of work to reach its steady
            # clocks, and twelve steps of a 0.1 ms configuration are 1.2 ms (round 3 printed C2 22 % below its
            # steady state for that reason alone: 2 113 against 2 550 Gbases/s on the same box, tools/gpu_size_curve.py)
            tw = time.perf_counter()
            while True:
                for _ in range(12):
                    step()
                torch.cuda.synchronize(dev)
                if time.perf_counter() - tw > 0.06:
                    break
            med, _ = timed_kernel_ms(step)
            ws.check()
            n_out = int(cnt.item()) if contigs is None else int(offs[-1])
            alg = (n + 3) // 4 + 4 * n_out * (2 if super_kmers else 1)  # + 4 n_out with super-k-mer indices
            extras.append({"config": name, "bases": n, "outputs": n_out, "kernel_ms": round(med, 4),
                           "Gbases_per_s": round(n / med / 1e6, 1), "algorithmic_bytes": alg,
                           "frac": round(alg / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)})
            del d, out, sk
            torch.cuda.empty_cache()

        secondary("C2 forward minimizers k=21 w=11, 256 Mbp (G seed 2)", sm.minimizers(21, 11), 268_435_456, 2, 2 / 12)
        secondary("forward minimizers k=21 w=11, 3.1 Gbp (G seed 3)", sm.minimizers(21, 11), N_BASES, SEED, 2 / 12)
        secondary("C4 canonical minimizers k=31 w=51, 24 CHM13-like contigs, one batch launch",
                  sm.canonical_minimizers(31, 51), 0, 0, 2 / 52, contigs=sharding.CHM13_CONTIG_LENGTHS)
        secondary("C5 canonical closed syncmers k=15 w=17, 3.1 Gbp (G seed 3)", sm.canonical_closed_syncmers(15, 17),
                  N_BASES, SEED, 2 / 17)
        # BASELINE config 5 says "with super-k-mer boundary emission"; the reference refuses .super_kmers() on syncmer
        # builders (src/lib.rs:339,496-500), so the conformant operation at that size is the minimizer builder's:
        # canonical_minimizers(21, 11).super_kmers(&mut sk) on the headline sequence (src/lib.rs:341-351,545-576)
        secondary("SK canonical minimizers k=21 w=11 + super-k-mer indices (.super_kmers), 3.1 Gbp (G seed 3)",
                  sm.canonical_minimizers(21, 11), N_BASES, SEED, 2 / 12, super_kmers=True)
        # The rows either side of the path: whole-call device time (torch events on the workspace's
        # stream, median of 5 after 3 warm-up calls), algorithmic bytes and their fraction of the HBM peak
        from simd_minimizers_amd import workloads
        for comp in workloads.COMPONENTS:
            try:
                extras.append(workloads.measure(comp, ws, dev))
            except Exception as e:  # a component must never take the headline down
                extras.append({"component": comp, "error": str(e)[:200]})
        try:  # the reference's short-sequence ladder (round 6)
            extras.append(workloads.ladder(ws, dev))
        except Exception as e:
            extras.append({"component": "LADDER", "error": str(e)[:200]})

    link_trace("after the secondary configurations")
    # ---------------------------------------------------------------- workload set-up
    n = args.bases
    gather_ms = None
    if args.workload == "headline":
        k, w = K, W
        b = sm.canonical_minimizers(k, w).workspace(ws)
        d_packed = generate(n, SEED + rank)
        cap = int(n * 2.3 / (w + 1)) + 4096
        out = torch.empty(cap, dtype=torch.int32, device=dev)
        d_count = torch.zeros(1, dtype=torch.int64, device=dev)
        my_bases, total_bases = n, n * world
        # (one GPU: the default series N = 1, 2, 4, 8 keeps the TOTAL at 3.1 Gbp - its first point is this very call)
        scaling = "weak" if world > 1 else "strong"
        kernel_name = "mm::fused_kernel<11, true, true, 0, false, false>"
        workload = (f"canonical minimizers k={k} w={w}, one {n} bp PackedSeq per GPU "
                    f"(generator G seed {SEED}+rank), device-resident input and output")

        def step():
            b.run_device(d_packed, n, out, sync=False, d_count=d_count)

        def outputs():
            return int(d_count.item())
    elif args.workload == "strong":
        k, w = K, W
        b = sm.canonical_minimizers(k, w).workspace(ws)
        d_packed = generate(n, SEED)
        plan_s = strong_plan(n, world, k, w)
        wb, we = plan_s["ranges"][rank]
        cap = int((we - wb) * 2.3 / (w + 1)) + 4096
        out = torch.empty(cap, dtype=torch.int32, device=dev)
        d_count = torch.zeros(1, dtype=torch.int64, device=dev)
        my_bases, total_bases = we - wb, n
        scaling = "strong"
        kernel_name = "mm::fused_kernel<11, true, true, 0, false, false>"
        workload = (f"canonical minimizers k={k} w={w}, ONE {n} bp PackedSeq (G seed {SEED}) cut into {world} window "
                    f"ranges, absolute positions, exact seam; device-resident")

        def step():
            b.run_device(d_packed, n, out, win_begin=wb, win_end=we, sync=False, d_count=d_count)

        def outputs():
            return int(d_count.item())
    else:  # contigs: BASELINE config 4
        k, w = 31, 51
        b = sm.canonical_minimizers(k, w).workspace(ws)
        lengths = list(sharding.CHM13_CONTIG_LENGTHS)
        mine = sharding.assign_contigs(lengths, world)[rank]
        d_seqs = [generate(lengths[i], sharding.CHM13_CONTIG_SEED0 + i) for i in mine]
        my_lens = [lengths[i] for i in mine]
        my_bases, total_bases = sum(my_lens), sum(lengths)
        cap = int(my_bases * 2.3 / (w + 1)) + 4096
        out = torch.empty(cap, dtype=torch.int32, device=dev)
        offs = [0]
        scaling = "strong"
        kernel_name = "mm::fused_kernel<51, true, true, 0, false, false> (batch mode)"
        workload = (f"canonical minimizers k={k} w={w}, 24 CHM13-like contigs ({total_bases} bp, G seed 100+contig) "
                    f"placed greedily on {world} GPUs, one batch launch per GPU and step (mm_run_batch_device, "
                    f"contig-local positions), RCCL gather of the position buffers to rank 0 timed separately")

        def step():
            offs[:] = sm.run_batch_device(b, d_seqs, my_lens, out)

        def outputs():
            return int(offs[-1])
    torch.cuda.synchronize(dev)

    # Ramp (round 5; untimed, BEFORE the W warm-up steps, disclosed as config.clock_ramp_ms).  The set-up above
    # (allocation, synthetic input) leaves the chip idle; W = 5 steps of a 1.5 ms kernel do not bring it back to its
    # sustained state - the same kernel read 1.56 ms per step behind five warm-up steps and 1.49 ms behind 200 ms of
    # steps on one box, 4.4 %, with the shader clock reading no lower in the slow case (profiles/r05_clock_ramp.txt: it is
    # not only the shader clock that comes up under load).  `value` is a sustained rate, so the chip runs this step back
    # to back for MM_BENCH_RAMP_MS (default 200) milliseconds first; the W warm-up steps and the K timed steps follow.
    ramp_ms = float(os.environ.get("MM_BENCH_RAMP_MS", "200"))
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < ramp_ms:
        for _ in range(8):
            step()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if args.warmup == 0:
        step()
        torch.cuda.synchronize(dev)
    n_out = outputs()
    assert 0 < n_out <= cap, (n_out, cap)

    ws.enable_timing(True)
    ws.kernel_time(True)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    kern_ms, launches = ws.kernel_time(True)
    ws.enable_timing(False)
    # Shader clock under this kernel, sampled OUTSIDE the timed region over a few more identical steps: 16 sleeping
    # waves on a stream of their own read the shader cycle counter against the 100 MHz real-time counter
    # (mm_clock_probe_*).  Not inside the timed loop: a second active queue alone slows the steps by ~7 %
    # (measured: 1.96 against 1.82 ms), whatever runs on it.
    link_trace("after the timed loop")
    live_clock = None
    if rank == 0 and world == 1 and hasattr(ws, "clock_probe_begin"):
        try:
            probe_steps = max(4, args.steps // 2)
            ws.clock_probe_begin(int(probe_steps * kern_ms / max(1, launches) * 1000 * 0.9))
            for _ in range(probe_steps):
                step()
            torch.cuda.synchronize(dev)
            live_clock = ws.clock_probe_end()
        except Exception:
            live_clock = None
    link_trace("after the clock probe")
    ws.check()  # no asynchronous run of the timed loop reported a look-back time-out / kernel error

    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # every rank must have produced a plausible result (density ~ 2/(w+1))
        ok = torch.tensor([1 if abs(n_out / my_bases - 2.0 / (w + 1)) < 0.01 else 0], dtype=torch.int32, device=rdev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        assert int(ok.item()) == 1, "a rank produced an implausible number of minimizers"

    multi_extra = []
    median5 = end_to_end = total_out = None
    emit_state = {"done": False}
    emit_lock = threading.Lock()

    def emit_line():
        """Rank 0 prints the ONE JSON line (once: the watchdog of the multi-GPU extras may get here first)."""
        with emit_lock:
            if emit_state["done"]:
                return
            emit_state["done"] = True
            emit_state["by_watchdog"] = threading.current_thread() is not threading.main_thread()
            if rank == 0:
                bases_done = float(total_bases) * args.steps
                # PackedSeq read + u32 positions written, for the units ONE launch of this rank processes
                alg_bytes = (my_bases + 3) // 4 + 4 * n_out
                kern_s = kern_ms / 1e3 / max(1, launches)
                achieved = alg_bytes / kern_s / 1e9
                traffic, valu, prov = (None, None, "not recorded for this workload")
                if args.workload == "headline" and n == N_BASES:
                    traffic, valu, prov = recorded_counters(kern_s * 1e3, live_clock)
                config = {"workload": workload, "k": k, "w": w, "bases_per_gpu": my_bases, "outputs_per_gpu": n_out,
                          "kernel": kernel_name, "parallelism": f"shard{world}", "clock_ramp_ms": ramp_ms}
                if args.workload == "strong":
                    config["windows_per_rank"] = [e - a for a, e in plan_s["ranges"]]
                    config["bases_total"] = plan_s["total_bases"]
                if gather_ms is not None:
                    config["gather_ms"] = round(gather_ms, 3)
                    config["gather"] = ("all-reduce of the per-contig counts + dist.gather of the device-resident position "
                                        f"buffers to rank 0 ({backend}); {total_out} positions in total; median of 3")
                roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                            "kernel_ms": round(kern_s * 1e3, 4), "algorithmic_bytes": alg_bytes,
                            "traffic_source": prov,
                            "note": "HBM is the nominal bound (SURVEY.md 8d); the kernel is VALU-issue bound, see `valu` "
                                    "(second bound: VALU-pipe time of one launch / kernel time) and DESIGN.md 4.1"}
                if valu is not None:
                    roofline["valu"] = valu
                line = {
                    "metric": METRIC,
                    "value": round(bases_done / dt / 1e9, 3),
                    "unit": "Gbases/s",
                    "n_gpus": world,
                    "steps": args.steps,
                    "warmup": args.warmup,
                    "ms_per_step": round(dt / args.steps * 1e3, 4),
                    "higher_is_better": True,
                    "scaling": scaling,
                    "vs_baseline": None,
                    "dtype": "u32",
                    "data": "synthetic",
                    "config": config,
                    "roofline": roofline,
                }
                if median5 is not None:
                    line["median_of_5"] = median5
                if end_to_end is not None:
                    line["end_to_end"] = end_to_end
                if extras or multi_extra:
                    line["extra"] = extras + multi_extra
                if not args.no_cpu_baseline and world == 1:
                    line["cpu_baseline"] = cpu_baseline()
                sys.stdout.flush()
                os.dup2(saved_stdout, 1)
                print(json.dumps(line), flush=True)
                os.dup2(2, 1)


    # ---------------------------------------------------------------- N > 1, default workload: the other two figures
    # (untimed region; the same barrier + max-over-ranks protocol, a handful of steps each)
    if distributed and args.workload == "strong" and not args.no_extra:
        # The two extra figures run collectives that the default line does not need (barriers around a second and a
        # third workload, point-to-point sends of the gather).  A rank that fails or hangs inside them must not take the
        # headline down: after MM_BENCH_EXTRA_TIMEOUT seconds (default 240) rank 0 prints the line with what it has and
        # every rank leaves.
        extra_deadline = float(os.environ.get("MM_BENCH_EXTRA_TIMEOUT", "240"))

        def watchdog():
            time.sleep(extra_deadline)
            if emit_state["done"]:
                return
            if rank == 0:
                multi_extra.append({"config": "multi-GPU extras", "error": f"not finished after {extra_deadline:g} s (a rank "
                                    "failed or hung in a collective); the headline figures of this line are complete"})
                emit_line()
            os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()

        def timed_all_ranks(fn, steps_x):
            for _ in range(3):
                fn()
            torch.cuda.synchronize(dev)
            dist.barrier()
            tx = time.perf_counter()
            for _ in range(steps_x):
                fn()
            torch.cuda.synchronize(dev)
            dist.barrier()
            tt = torch.tensor([time.perf_counter() - tx], dtype=torch.float64, device=rdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item()) / steps_x

        steps_x = max(3, min(10, args.steps))
        # weak: every rank walks the whole 3.1 Gbp sequence by itself (independent genomes; linear by construction)
        try:
            out_full = torch.empty(int(n * 2.3 / (w + 1)) + 4096, dtype=torch.int32, device=dev)
            t_weak = timed_all_ranks(lambda: b.run_device(d_packed, n, out_full, sync=False, d_count=d_count), steps_x)
            del out_full
            multi_extra.append({"config": f"weak: one {n} bp sequence per GPU, {world} GPUs", "scaling": "weak",
                                "ms_per_step": round(t_weak * 1e3, 4), "Gbases_per_s": round(n * world / t_weak / 1e9, 1)})
        except Exception as e:  # must never take the default line down
            multi_extra.append({"config": "weak", "error": str(e)[:200]})
        # contigs (BASELINE config 4): 24 CHM13-like contigs placed greedily, one batch launch per rank, then the gather
        try:
            if n != N_BASES:
                raise RuntimeError("skipped: reduced --bases")
            kc, wc = 31, 51
            bc = sm.canonical_minimizers(kc, wc).workspace(ws)
            lengths_c = list(sharding.CHM13_CONTIG_LENGTHS)
            mine_c = sharding.assign_contigs(lengths_c, world)[rank]
            d_c = [generate(lengths_c[i], sharding.CHM13_CONTIG_SEED0 + i) for i in mine_c]
            lens_c = [lengths_c[i] for i in mine_c]
            out_c = torch.empty(int(sum(lens_c) * 2.3 / (wc + 1)) + 4096, dtype=torch.int32, device=dev)
            offs_c = [0]

            def step_c():
                offs_c[:] = sm.run_batch_device(bc, d_c, lens_c, out_c)
            t_c = timed_all_ranks(step_c, steps_x)
            dist.barrier()
            tg = time.perf_counter()
            _, _, _, counts_c, gathered_c = sharding.run_contig_batch_sharded(lambda idx: (out_c, offs_c), lengths_c, gather_to=0)
            torch.cuda.synchronize(dev)
            g_ms = (time.perf_counter() - tg) * 1e3
            multi_extra.append({"config": f"C4 canonical minimizers k=31 w=51, 24 CHM13-like contigs on {world} GPUs, one "
                                          "batch launch per GPU and step", "scaling": "strong",
                                "ms_per_step": round(t_c * 1e3, 4), "Gbases_per_s": round(sum(lengths_c) / t_c / 1e9, 1),
                                "gather_ms": round(g_ms, 3), "positions": int(sum(counts_c)),
                                "gather": f"per-contig counts all-reduced, position buffers sent point-to-point to rank 0 ({backend})"})
            del d_c, out_c, gathered_c
        except Exception as e:  # must never take the default line down
            multi_extra.append({"config": "C4 contigs", "error": str(e)[:200]})

    # ---------------------------------------------------------------- gather (config 4), separately timed
    if args.workload == "contigs":
        times = []
        for _ in range(3):
            torch.cuda.synchronize(dev)
            if distributed:
                dist.barrier()
            tg = time.perf_counter()
            _, _, _, counts, gathered = sharding.run_contig_batch_sharded(
                lambda idx: (out, offs), lengths, gather_to=0)
            torch.cuda.synchronize(dev)
            times.append((time.perf_counter() - tg) * 1e3)
            if rank == 0:
                assert gathered is not None and all(int(g.numel()) == int(c) for g, c in zip(gathered, counts))
            del gathered
        gather_ms = statistics.median(times)
        total_out = int(sum(counts))

    # ---------------------------------------------------------------- untimed extras (rank 0, N = 1)
    if rank == 0 and world == 1 and not args.no_extra:
        med, allms = timed_kernel_ms(step)
        median5 = {"kernel_ms": round(med, 4), "Gbases_per_s": round(my_bases / med / 1e6, 1),
                   "all_ms": [round(x, 4) for x in allms],
                   "protocol": "warm-up + 5 repeats, median (bench/src/bin/paper.rs:536-556); kernel time by HIP events"}
        if args.workload == "headline":
            # One rank's share of the strong split at N = 8, timed ALONE on this GPU exactly like the timed loop above
            # (wall clock around `steps` asynchronous calls): shard_efficiency = t_full / (8 x t_shard) is what a
            # strong 1 -> 8 scaling curve of this kernel can reach at best - the part of it one GPU can measure.
            try:
                sb, se = strong_plan(n, 8, k, w)["ranges"][3]

                def shard_step():
                    b.run_device(d_packed, n, out, win_begin=sb, win_end=se, sync=False, d_count=d_count)
                for _ in range(max(5, args.warmup)):
                    shard_step()
                torch.cuda.synchronize(dev)
                ts = time.perf_counter()
                for _ in range(args.steps):
                    shard_step()
                torch.cuda.synchronize(dev)
                t_shard = (time.perf_counter() - ts) / args.steps
                med_s, _ = timed_kernel_ms(shard_step)
                shard_row = {"config": f"1/8 window range of the headline sequence ([{sb}, {se}): rank 3 of 8), alone on one GPU",
                             "windows": se - sb, "outputs": int(d_count.item()), "ms_per_step": round(t_shard * 1e3, 4),
                             "kernel_ms": round(med_s, 4), "Gbases_per_s": round((se - sb) / t_shard / 1e9, 1),
                             "shard_efficiency": round((dt / args.steps) / (8 * t_shard), 4),
                             "what": "shard_efficiency = ms_per_step of the full sequence / (8 x ms_per_step of the shard)"}
                extras.append(shard_row)
            except Exception as e:
                extras.append({"config": "1/8 window range", "error": str(e)[:200]})
            try:
                import ctypes as C

                import numpy as np
                hp, hp_owner = sm.pinned_array(((n + 3) // 4 + 64,), np.uint8)
                hp[:] = d_packed.cpu().numpy()
                ho, ho_owner = sm.pinned_array((n_out + 1024,), np.uint32)
                ho[:] = 0
                cnt = C.c_uint64()
                plan = b.plan()
                # Ten calls back to back; `ms` = the median of the last five.  The first calls of a process whose link has
                # been idle run far below the link's rate - 71-72 ms for calls 2-4 of this script in rounds 4 and 5, 41 ms
                # from the fifth on, in the same process, on the same buffers (profiles/r05_host_path.txt) - so every call's
                # time is printed (`calls_ms`) and the steady state is what is compared with the link's floor.
                e2e = []
                for i in range(10):
                    if i == 1 and os.environ.get("MM_ENV_DYNAMIC"):
                        os.environ["MM_PIPE_TRACE"] = "1"  # (diagnostics: the chunk milestones of an early call, to stderr)
                    te = time.perf_counter()
                    sm._check(L.mm_run_host(plan.h, ws.h, hp.ctypes.data_as(C.POINTER(C.c_uint8)), 0, n,
                                            ho.ctypes.data_as(C.POINTER(C.c_uint32)), None, n_out + 1024, C.byref(cnt)))
                    e2e.append((time.perf_counter() - te) * 1e3)
                    os.environ.pop("MM_PIPE_TRACE", None)
                assert cnt.value == n_out
                calls_ms = [round(x, 2) for x in e2e]
                m = statistics.median(e2e[5:])
                host_ab = None
                # (diagnostics: the call's mechanisms in THIS process state - on request, or when the call took far longer
                # than it does in a fresh process: 39.5 ms on every box of round 5, 71-72 ms in one run of this script in three)
                if os.environ.get("MM_BENCH_HOST_AB") or (m > 55.0 and os.environ.get("MM_ENV_DYNAMIC")):
                    host_ab = {}
                    for om, im in (("engine", "engine"), ("blit", "engine"), ("direct", "engine"), ("blit", "blit")):
                        os.environ["MM_HOST_OUT"], os.environ["MM_HOST_IN"] = om, im
                        tt = []
                        for _ in range(3):
                            te = time.perf_counter()
                            sm._check(L.mm_run_host(plan.h, ws.h, hp.ctypes.data_as(C.POINTER(C.c_uint8)), 0, n,
                                                    ho.ctypes.data_as(C.POINTER(C.c_uint32)), None, n_out + 1024, C.byref(cnt)))
                            tt.append((time.perf_counter() - te) * 1e3)
                        host_ab[f"out={om},in={im}"] = round(min(tt), 2)
                    os.environ.pop("MM_HOST_OUT", None)
                    os.environ.pop("MM_HOST_IN", None)
                    os.environ["MM_PIPE_TRACE"] = "1"  # (the chunks' milestones of one more call, to stderr)
                    sm._check(L.mm_run_host(plan.h, ws.h, hp.ctypes.data_as(C.POINTER(C.c_uint8)), 0, n,
                                            ho.ctypes.data_as(C.POINTER(C.c_uint32)), None, n_out + 1024, C.byref(cnt)))
                    os.environ.pop("MM_PIPE_TRACE", None)
                    try:
                        want = {f"{hp.ctypes.data:x}", f"{ho.ctypes.data:x}"}
                        for ln in open("/proc/self/numa_maps"):
                            if ln.split(" ", 1)[0] in want:
                                print("[numa_maps] " + ln.strip(), file=sys.stderr, flush=True)
                    except Exception:
                        pass
                # the link itself, measured beside it (this figure moved 39.6 -> 71.4 ms between two driver boxes of round
                # 4): 512 MiB each way between the same page-locked buffers and the device - each direction alone and,
                # since round 5, BOTH AT ONCE (mm_link_probe: the copy engines on two streams).  The two directions do not
                # add up (97 GB/s together against 57.6 + 57.1 alone on round 5's boxes), so the call's floor is priced with
                # the rate they reach together: both run at half of it until the upload is through, the rest of the
                # positions then leaves at the one-way rate.
                link = {}
                try:
                    nb = min(512 << 20, hp.nbytes, ho.nbytes)
                    rates = (C.c_double * 3)()
                    sm._check(L.mm_link_probe(ws.h, C.c_void_p(hp.ctypes.data), C.c_void_p(ho.ctypes.data), nb, rates))
                    link["h2d_GBps"], link["d2h_GBps"] = round(rates[0], 1), round(rates[1], 1)
                    link["both_directions_GBps"] = round(rates[2], 1)
                    b_in, b_out = n / 4.0, 4.0 * n_out
                    half = rates[2] * 1e9 / 2.0
                    t_overlap = min(b_in, b_out) / half
                    rest = (b_out - b_in) / (rates[1] * 1e9) if b_out > b_in else (b_in - b_out) / (rates[0] * 1e9)
                    link["floor_ms_at_these_rates"] = round((t_overlap + rest) * 1e3, 2)
                    link["floor_ms_if_full_duplex"] = round(max(b_in / (rates[0] * 1e9), b_out / (rates[1] * 1e9)) * 1e3, 2)
                    link["floor_what"] = ("both directions at half of both_directions_GBps until the smaller transfer is through, the "
                                          "rest at its one-way rate; floor_ms_if_full_duplex = max of the two one-way times, the "
                                          "figure of rounds 3-4, which no call can reach on a link that does not run 2 x one-way")
                    # where this process sits relative to the GPU: a page-locked buffer on the OTHER socket's memory makes
                    # every transfer cross the inter-socket link as well (the same code measured 39.6 and 71.5 ms on boxes
                    # of this pool whose one-way rates were both 57 GB/s)
                    try:
                        pr = torch.cuda.get_device_properties(dev)
                        bdf = f"{int(pr.pci_domain_id):04x}:{int(pr.pci_bus_id):02x}:{int(pr.pci_device_id):02x}.0"
                    except Exception:
                        bdf = None
                    try:
                        gpu_node = int(open(f"/sys/bus/pci/devices/{bdf.lower()}/numa_node").read()) if bdf else None
                    except Exception:
                        gpu_node = None
                    try:
                        # (the CPU this thread last ran on: field 39 of /proc/self/stat, counted behind the ")" of the name)
                        cpu = int(open("/proc/self/stat").read().rsplit(")", 1)[1].split()[36])
                        import glob as _glob
                        nodes = _glob.glob(f"/sys/devices/system/cpu/cpu{cpu}/node*")
                        cpu_node = int(os.path.basename(nodes[0])[4:]) if nodes else None
                    except Exception:
                        cpu_node = None
                    link["numa"] = {"gpu_node": gpu_node, "cpu_node_of_this_thread": cpu_node}
                except Exception as e:
                    link = {"error": str(e)[:120]}
                end_to_end = {"ms": round(m, 2), "Gbases_per_s": round(n / m / 1e6, 1), "link": link,
                              "what": "mm_run_host: H2D of the packed bytes + kernel + D2H of the positions, "
                                      "page-locked caller buffers (mm_host_alloc), pipelined in 16 chunks (at most two "
                                      "uploads and two downloads in the runtime's hands at a time, counts polled from "
                                      "page-locked words the kernels store); ten calls back to back, `ms` = median of the "
                                      "last five, every call in `calls_ms`; PCIe-bound, never part of `value`",
                              "ms_over_floor": (round(m / link["floor_ms_at_these_rates"], 3)
                                                if isinstance(link.get("floor_ms_at_these_rates"), float) else None)}
                end_to_end["calls_ms"] = calls_ms
                if host_ab:
                    end_to_end["mechanisms_ms"] = host_ab
                del hp, ho, hp_owner, ho_owner
            except Exception as e:  # host memory limits of the box
                end_to_end = {"error": str(e)[:200]}

    emit_line()
    if emit_state.get("by_watchdog"):
        time.sleep(60.0)  # (the watchdog thread is printing the line and ends the process)
    if distributed:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
