"""Synthetic workloads of the rows either side of the hot path (reads mode, super-k-mer indices,
skip-ambiguous windows, k-mer values, ASCII and FASTA packing), shared by ``bench.py`` (its ``extra`` list, outside
the timed region) and ``tools/run_config.py`` (the command rocprofv3 wraps for ``profiles/``).

``component(name, ws, dev)`` returns a dict: ``step`` (one invocation, asynchronous on the workspace's stream, which
must be torch's current stream), ``units`` / ``unit`` (what one step processes), ``alg_bytes()`` (algorithmic HBM
bytes of one step, callable after a step has run: inputs read once + outputs written once), ``what`` and
``kernels`` (substrings of the kernel names one step launches, for the profiler summaries)."""
import ctypes as C

from . import (Builder, _check, canonical_minimizers, lib, run_reads_device)

COMPONENTS = ("READS", "READS_SK", "SKIP", "VALUES", "PACK", "FASTA", "FASTQ")


def _generate(ws, dev, n, seed):
    import torch
    t = torch.zeros((n + 3) // 4 + 64, dtype=torch.uint8, device=dev)
    _check(lib().mm_generate_device_async(ws.h, seed, 0, n, t.data_ptr()))
    return t


def component(name, ws, dev):
    import torch
    L = lib()
    k, w = 21, 11
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
    t0 = time.perf_counter()
    while True:
        for _ in range(warm):
            c["step"]()
        torch.cuda.synchronize(dev)
        if time.perf_counter() - t0 > 0.06:
            break
    ms = []
    for _ in range(reps):
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
    c.clear()
    torch.cuda.empty_cache()
    return rec
