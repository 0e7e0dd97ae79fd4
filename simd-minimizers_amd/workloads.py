"""Synthetic workloads of the rows either side of the hot path (reads mode, super-k-mer indices,
skip-ambiguous windows, k-mer values, ASCII and FASTA packing), shared by ``bench.py`` (its ``extra`` list, outside
the timed region) and ``tools/run_config.py`` (the command rocprofv3 wraps for ``profiles/``).

``component(name, ws, dev)`` returns a dict: ``step`` (one invocation, asynchronous on the workspace's stream, which
must be torch's current stream), ``units`` / ``unit`` (what one step processes), ``alg_bytes()`` (algorithmic HBM
bytes of one step, callable after a step has run: inputs read once + outputs written once), ``what`` and
``kernels`` (substrings of the kernel names one step launches, for the profiler summaries)."""
import ctypes as C
import os

from . import (Builder, _check, canonical_minimizers, lib, run_reads_device)

COMPONENTS = ("READS", "READS_SK", "SKIP", "VALUES", "PACK", "FASTA", "FASTQ", "READS_VAR", "LONGREADS", "BATCH10K")

# The reference's `short` experiment (bench/src/bin/paper.rs:62-115): sequences of random length in [n, 2n), one
# Builder::run per sequence; its published forward (w=11, k=21) figures in ns per base (bench/results-neon.json, BASELINE.md
# section 1 - ARM NEON, not measured here).
LADDER_N = (16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192)
LADDER_REF_NS_PER_BASE = (20.4, 20.5, 16.8, 9.5, 5.9, 3.9, 2.9, 2.5, 2.2, 2.1)


def _generate(ws, dev, n, seed):
    import torch
    t = torch.zeros((n + 3) // 4 + 64, dtype=torch.uint8, device=dev)
    _check(lib().mm_generate_device_async(ws.h, seed, 0, n, t.data_ptr()))
    return t


def component(name, ws, dev):
    import torch
    L = lib()
    k, w = int(os.environ.get("MM_WL_K", "21")), int(os.environ.get("MM_WL_W", "11"))  # (the bench rows: k=21 w=11; tools override)
    if name in ("READS", "READS_SK"):
        n_reads, rl = 8_000_000, 150
        n = n_reads * rl
        b = canonical_minimizers(k, w).workspace(ws)
        d = _generate(ws, dev, n, 7)
        out = torch.empty(int(n * 0.2), dtype=torch.int32, device=dev)
        sk = torch.empty_like(out) if name == "READS_SK" else None
        offs = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)

        def step():
            run_reads_device(b, d, n_reads, rl, rl, out, offs, d_count=cnt, sync=False, out_sk=sk)

        def alg():
            c = int(cnt.item())
            return (n + 3) // 4 + 4 * c * (2 if sk is not None else 1) + 8 * (n_reads + 1)
        return {"step": step, "units": n, "unit": "bases", "alg_bytes": alg, "keep": (d, out, sk, offs, cnt),
                "what": f"reads mode: {n_reads} reads x {rl} bp, canonical minimizers k={k} w={w}"
                        + (" with super-k-mer indices" if sk is not None else "") + ", one launch (src/lib.rs:378 per read)",
                "kernels": ["fused_kernel"]}
    if name in ("READS_VAR", "LONGREADS") or name.startswith("LADDER_"):
        # reads of MIXED lengths packed back to back (the FASTQ packer's layout), one call of mm_run_packed_reads_device:
        #   READS_VAR   8 M reads of 100 .. 200 bp (one lane per read: every lane of a wave walks its longest read)
        #   LONGREADS   200 k reads, lengths log-uniform in 1 .. 50 kbp - the HiFi / ONT regime; ONE lane-table launch (round 6)
        #   LADDER_<n>_<F|C>  a rung of the reference's `short` experiment: lengths uniform in [n, 2n), 2^30 bases in all,
        #               forward / canonical minimizers
        g = torch.Generator(device=dev)
        g.manual_seed(6)
        canonical = True
        if name == "READS_VAR":
            lens = torch.randint(100, 201, (8_000_000,), device=dev, generator=g)
            label = "8 M reads of 100 .. 200 bp"
        elif name == "LONGREADS":
            import math
            u = torch.rand(200_000, device=dev, generator=g, dtype=torch.float64)
            lens = torch.exp(math.log(1000.0) + u * (math.log(50_000.0) - math.log(1000.0))).to(torch.int64)
            label = "200 k reads, lengths log-uniform in 1 .. 50 kbp (long reads)"
        else:
            _, ln, fl = name.split("_")
            ln = int(ln)
            canonical = fl == "C"
            lens = torch.randint(ln, 2 * ln, ((1 << 30) * 2 // (3 * ln),), device=dev, generator=g)
            label = f"{lens.numel()} sequences of {ln} .. {2 * ln - 1} bp (the reference's `short` experiment, bench/src/bin/paper.rs:62-115)"
        n_reads = int(lens.numel())
        starts = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
        starts[1:] = torch.cumsum(lens, 0)
        n = int(starts[-1].item())
        mx = int(lens.max().item())
        del lens
        b = Builder(k, w, canonical, 0).workspace(ws)
        d = _generate(ws, dev, n, 7)
        out = torch.empty(int(n * 0.19) + 4096, dtype=torch.int32, device=dev)
        offs = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)

        def step():
            _check(L.mm_run_packed_reads_device_async(b.plan().h, ws.h, C.c_void_p(d.data_ptr()), d.numel(), 0, n_reads,
                                                      C.c_void_p(starts.data_ptr()), n, mx, C.c_void_p(out.data_ptr()), None,
                                                      out.numel(), C.c_void_p(offs.data_ptr()), C.c_void_p(cnt.data_ptr())))

        def alg():  # the packed bases, the starts in, positions and offsets out
            return (n + 3) // 4 + 4 * int(cnt.item()) + 16 * (n_reads + 1)
        return {"step": step, "units": n, "unit": "bases", "alg_bytes": alg, "keep": (d, out, offs, cnt, starts),
                "what": f"{label}, {'canonical' if canonical else 'forward'} minimizers k={k} w={w}, packed back to back, one call of "
                        "mm_run_packed_reads_device (src/lib.rs:378 per read)",
                "kernels": ["fused_kernel", "seg_"], "lane_table": lambda: ws.last_lane_table(), "reads": n_reads}
    if name == "BATCH10K":
        # 20 000 contigs of 10 kbp through mm_run_batch_device: one lane-table launch (round 6; before: every contig tiles of its own)
        n_seqs, ln = 20_000, 10_000
        n = n_seqs * ln
        b = canonical_minimizers(k, w).workspace(ws)
        d = _generate(ws, dev, n, 9)
        out = torch.empty(int(n * 0.19) + 4096, dtype=torch.int32, device=dev)
        ptrs = (C.c_void_p * n_seqs)(*[d.data_ptr() + (i * ln) // 4 for i in range(n_seqs)])
        nbytes = (C.c_uint64 * n_seqs)(*[d.numel() - (i * ln) // 4 for i in range(n_seqs)])
        lens = (C.c_uint64 * n_seqs)(*([ln] * n_seqs))
        boffs = (C.c_uint64 * n_seqs)(*([0] * n_seqs))
        out_offsets = (C.c_uint64 * (n_seqs + 1))()

        def step():  # (synchronous: the offsets come back to the host)
            _check(L.mm_run_batch_device(b.plan().h, ws.h, n_seqs, ptrs, nbytes, boffs, lens, C.c_void_p(out.data_ptr()), None,
                                         out.numel(), out_offsets))
        return {"step": step, "units": n, "unit": "bases", "alg_bytes": lambda: (n + 3) // 4 + 4 * int(out_offsets[n_seqs]) + 8 * (n_seqs + 1),
                "keep": (d, out), "kernel_timing": True,
                "what": f"{n_seqs} contigs x {ln} bp, canonical minimizers k={k} w={w}, one call of mm_run_batch_device (kernel time by HIP "
                        "events; the call itself is synchronous and returns the offsets to the host)",
                "kernels": ["fused_kernel", "seg_"], "lane_table": lambda: ws.last_lane_table(), "reads": n_seqs}
    if name == "SKIP" or name.startswith(("SKIP_W", "PLAIN_W")):
        # (SKIP: the bench row, k=21 w=11.  SKIP_W33 / _W51 and their PLAIN_ twins: tools/prof_head.py stalls:<name> - the dirty
        # walk of the large windows beside the plain walk on the same sequence, profiles/r05_skip_dirty_walk.txt)
        if name != "SKIP":
            w = int(name.split("_W")[1])
            k = 31 if w % 2 else 30  # (canonical windows need odd k + w - 1)
        n = 1_000_000_000
        d = _generate(ws, dev, n, 2)
        amb = torch.zeros((n + 7) // 8 + 64, dtype=torch.uint8, device=dev)
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        amb[torch.randint(0, n // 8, (n // 8000,), device=dev, generator=g)] = 1 << 3   # 0.1 % isolated Ns
        for s in torch.randint(0, n // 8 - 7000, (200,), generator=g, device=dev).tolist():
            amb[s:s + 6250] = 0xFF                                                      # 200 gaps of 50 kbp
        out = torch.empty(int(n * 0.2), dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        b = canonical_minimizers(k, w).workspace(ws)

        def step():
            if name.startswith("PLAIN"):
                b.run_device(d, n, out, sync=False, d_count=cnt)
            else:
                b.run_skip_ambiguous_device(d, amb, n, out, sync=False, d_count=cnt)

        def alg():
            return (n + 3) // 4 + (0 if name.startswith("PLAIN") else (n + 7) // 8) + 4 * int(cnt.item())
        return {"step": step, "units": n, "unit": "bases", "alg_bytes": alg, "keep": (d, amb, out, cnt),
                "what": f"skip-ambiguous windows (PackedNSeq, src/lib.rs:451-496): canonical k={k} w={w} on {n} bp with "
                        "0.1 % isolated Ns and 200 gaps of 50 kbp; window-ambiguity prepass + walk",
                "kernels": ["window_ambiguity_kernel", "fused_kernel"]}
    if name == "VALUES":
        n = 1_000_000_000
        d = _generate(ws, dev, n, 5)
        pos = torch.empty(int(n * 0.2), dtype=torch.int32, device=dev)
        c = canonical_minimizers(k, w).workspace(ws).run_device(d, n, pos)
        vals = torch.empty(c, dtype=torch.int64, device=dev)

        def step():
            _check(L.mm_values_u64_device_async(ws.h, C.c_void_p(d.data_ptr()), d.numel(), 0, n, k, 1,
                                                C.c_void_p(pos.data_ptr()), c, C.c_void_p(vals.data_ptr())))
        return {"step": step, "units": c, "unit": "values", "alg_bytes": lambda: (n + 3) // 4 + 12 * c,
                "keep": (d, pos, vals),
                "what": f"Output::values_u64 (src/lib.rs:584-612): canonical {k}-mer values of the {c} minimizer "
                        f"positions of {n} bp",
                "kernels": ["values_u64_kernel"]}
    if name == "PACK":
        n = 1 << 30
        g = torch.Generator(device=dev)
        g.manual_seed(3)
        asc = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (n,), device=dev, generator=g)]
        packed = torch.empty(n // 4 + 64, dtype=torch.uint8, device=dev)

        def step():
            _check(L.mm_pack_ascii_device_async(ws.h, C.c_void_p(asc.data_ptr()), n, C.c_void_p(packed.data_ptr())))
        return {"step": step, "units": n, "unit": "bases", "alg_bytes": lambda: n + n // 4, "keep": (asc, packed),
                "what": f"PackedSeqVec::from_ascii on the device, {n} ASCII bases ((c >> 1) & 3, 4 per byte)",
                "kernels": ["pack_ascii"]}
    if name == "FASTA":
        n, width, n_rec = 1 << 30, 60, 24
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        t = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (n,), device=dev, generator=g)]
        i = torch.arange(n, device=dev)
        t[i % (width + 1) == width] = 10
        del i
        for r in range(n_rec):
            p = (n // n_rec) * r
            hdr = b">record%d\n" % r
            if p:
                t[p - 1] = 10
            t[p: p + len(hdr)] = torch.tensor(list(hdr), dtype=torch.uint8, device=dev)
        packed = torch.empty(n // 4 + 64, dtype=torch.uint8, device=dev)
        rb = torch.zeros(1025, dtype=torch.int64, device=dev)
        rp = torch.zeros(1024, dtype=torch.int64, device=dev)
        cnt = torch.zeros(2, dtype=torch.int64, device=dev)

        def step():
            _check(L.mm_fasta_pack_device_async(ws.h, C.c_void_p(t.data_ptr()), n, C.c_void_p(packed.data_ptr()),
                                                packed.numel() // 4 * 4, C.c_void_p(rb.data_ptr()),
                                                C.c_void_p(rp.data_ptr()), 1024, C.c_void_p(cnt.data_ptr())))
        return {"step": step, "units": n, "unit": "text bytes", "alg_bytes": lambda: n + (int(cnt[0].item()) + 3) // 4,
                "keep": (t, packed, rb, rp, cnt),
                "what": f"FASTA text -> packed records on the device (needletail + from_ascii, bench/src/lib.rs:51-82): "
                        f"{n} bytes of text, {width}-base lines, {n_rec} records",
                "kernels": ["fasta"]}
    if name == "FASTQ":
        # 150 bp reads: '@' + 19 name bytes, the sequence, '+', the qualities = 324 bytes per record
        rl, name_len = 150, 19
        rec_bytes = (1 + name_len + 1) + (rl + 1) + 2 + (rl + 1)
        n_rec = (1 << 30) // rec_bytes
        n = n_rec * rec_bytes
        g = torch.Generator(device=dev)
        g.manual_seed(2)
        t = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (n,), device=dev, generator=g)]
        i = torch.arange(n, device=dev) % rec_bytes
        t[i == 0] = ord("@")
        for p in (name_len + 1, name_len + 2 + rl, name_len + 2 + rl + 2, rec_bytes - 1):
            t[i == p] = 10
        t[i == name_len + 2 + rl + 1] = ord("+")
        del i
        packed = torch.empty(n // 4 + 64, dtype=torch.uint8, device=dev)
        rb = torch.zeros(n_rec + 1, dtype=torch.int64, device=dev)
        rp = torch.zeros(n_rec, dtype=torch.int64, device=dev)
        cnt = torch.zeros(2, dtype=torch.int64, device=dev)

        def step():
            _check(L.mm_fastq_pack_device_async(ws.h, C.c_void_p(t.data_ptr()), n, C.c_void_p(packed.data_ptr()),
                                                packed.numel() // 4 * 4, C.c_void_p(rb.data_ptr()),
                                                C.c_void_p(rp.data_ptr()), n_rec, C.c_void_p(cnt.data_ptr())))

        def alg():
            assert int(cnt[1].item()) == n_rec and int(cnt[0].item()) == n_rec * rl
            return n + (n_rec * rl + 3) // 4 + 16 * n_rec
        return {"step": step, "units": n, "unit": "text bytes", "alg_bytes": alg, "keep": (t, packed, rb, rp, cnt),
                "what": f"FASTQ text -> packed reads on the device (needletail + from_ascii, bench/src/lib.rs:51-82): {n} bytes "
                        f"of text, {n_rec} records of {rl} bases",
                "kernels": ["fastq"]}
    raise ValueError("unknown component " + name)


def measure(name, ws, dev, warm=3, reps=5):
    """Median device time of one step (torch events on the current stream = the workspace's stream) and the
    derived rates; everything the component allocated is released afterwards."""
    import statistics

    import torch
    import time
    c = component(name, ws, dev)
    # warm-up by TIME as well as by count (round 5): a component's set-up - allocation, synthetic input - leaves the chip
    # idle long enough for its clocks to drop, and three steps of a 0.7 ms kernel do not bring them back: the driver read
    # READS_SK at 0.884 ms in rounds 3 and 4 where a warmed-up run reads 0.74 (profiles/r05_reads_sk.txt)
    # (round 6: the FIRST step of a component may compile its kernel - READS_SK, the syncmer and lane-table flavours are
    # specialised at first use, 4-10 s with a cold code-object cache, i.e. on every fresh box - and that time used to count
    # as warm-up: the loop below ended after three steps with the chip idle for seconds, and the five timed steps ran on
    # cold clocks.  That was the driver's 0.88-0.90 ms for READS_SK in rounds 3-5 against 0.73-0.76 on boxes whose cache was
    # warm: profiles/r06_reads_sk_fresh.txt.  The first step now runs before the clock starts; MM_BENCH_OLD_WARMUP=1 is the A/B.)
    if not os.environ.get("MM_BENCH_OLD_WARMUP"):
        c["step"]()
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    while True:
        for _ in range(warm):
            c["step"]()
        torch.cuda.synchronize(dev)
        if time.perf_counter() - t0 > 0.06:
            break
    ms = []
    for _ in range(reps):
        if c.get("kernel_timing"):  # (a synchronous entry point: the walk kernel's own time, HIP events on the workspace's stream)
            ws.enable_timing(True)
            ws.kernel_time(True)
            c["step"]()
            t, launches = ws.kernel_time(True)
            ws.enable_timing(False)
            ms.append(t)
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        c["step"]()
        e1.record()
        torch.cuda.synchronize(dev)
        ms.append(e0.elapsed_time(e1))
    ws.check()
    med = statistics.median(ms)
    alg = int(c["alg_bytes"]())
    rec = {"component": name, "what": c["what"], "ms": round(med, 4), "units": c["units"], "unit": c["unit"],
           "G_units_per_s": round(c["units"] / med / 1e6, 1), "algorithmic_bytes": alg,
           "GB_per_s": round(alg / med / 1e6, 1), "frac": round(alg / (med * 1e-3) / 8e12, 4)}
    if "lane_table" in c:
        rec["lane_table"] = bool(c["lane_table"]())
        rec["reads"] = c["reads"]
    c.clear()
    torch.cuda.empty_cache()
    return rec


def ladder(ws, dev):
    """The reference's `short` experiment (bench/src/bin/paper.rs:62-115) as ONE call per rung: sequences of random length in
    [n, 2n) for n = 16 .. 8192, 2^30 bases per rung (the reference: 2^20, one Builder::run per sequence), forward and canonical
    minimizers k=21 w=11.  Whole-call device time (the lane table's kernels included), ns per base beside the reference's
    published NEON figure."""
    rows = []
    for n, ref in zip(LADDER_N, LADDER_REF_NS_PER_BASE):
        row = {"n": n, "lengths": f"{n}..{2 * n - 1}", "reference_neon_fwd_ns_per_base": ref}
        for fl, key in (("F", "forward"), ("C", "canonical")):
            try:
                r = measure(f"LADDER_{n}_{fl}", ws, dev)
                row[key] = {"ms": r["ms"], "Gbases_per_s": r["G_units_per_s"], "ns_per_base": round(r["ms"] * 1e6 / r["units"], 5),
                            "frac": r["frac"], "lane_table": r["lane_table"], "sequences": r["reads"]}
                row["bases"] = r["units"]
            except Exception as e:  # a rung must never take the bench line down
                row[key] = {"error": str(e)[:200]}
        rows.append(row)
    return {"component": "LADDER", "what": "the reference's short-sequence ladder (bench/src/bin/paper.rs:62-115; BASELINE.md section 1): "
            "lengths uniform in [n, 2n), 2^30 bases per rung packed back to back, minimizers k=21 w=11, one call of "
            "mm_run_packed_reads_device per rung (whole-call device time)", "rows": rows}
