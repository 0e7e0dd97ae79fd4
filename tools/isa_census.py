#!/usr/bin/env python3
"""Static census of the dominant kernel's main loop (runs on the build box, no GPU): compiles
mm::fused_kernel<W, CANON, ...> to gfx950 assembly with hipcc -S, finds the W-block loop of the plain
walk (fast emit, no range / ambiguity checks) and counts its VALU instructions by issue rate
(full / half rate as measured on MI355X in SHADER CYCLES, profiles/r03_valu_issue_rates.txt: 2.31 / 4.14 cycles
per wave64 instruction - tools/ubench/valu_rate.hip reads the shader clock under every loop since round 3, the
round-1 table assumed 2.4 GHz and read 2.5 / 4.3; any SGPR source operand makes an instruction half rate).  Writes profiles/head_isa_census.json,
which bench.py reads for roofline.valu.issue_clk.  usage: isa_census.py [W] [canon 0|1]"""
import collections
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from isa_loops import classify  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 11
CANON = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
csrc = os.path.join(ROOT, "simd-minimizers_amd", "csrc")
tmp = tempfile.mkdtemp(prefix="mm_census_")
src = os.path.join(tmp, "k.hip")
c = "true" if CANON else "false"
open(src, "w").write('#include "mm_fused_impl.h"\n'
                     f"template __global__ void mm::fused_kernel<{W}, {c}, {c}, 0, false, false>(const mm::FusedParams);\n")
asm = os.path.join(tmp, "k.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + csrc, "-S",
                "--cuda-device-only", "-o", asm, src], check=True, stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
labels, insts = {}, []
for ln in lines:
    s = ln.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        labels[m.group(1)] = len(insts)
        continue
    if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
        continue
    s = s.split(";")[0].strip()
    if s:
        insts.append(s)
best = None
NWIN = W  # windows one iteration of the main loop walks: W, or W x kWideGroup<W> since round 5 (whole load groups unrolled)
for i, s in enumerate(insts):
    m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", s) or re.match(r"s_branch\s+(\.LBB\d+_\d+)", s)
    if not m or m.group(1) not in labels or labels[m.group(1)] > i:
        continue
    body = insts[labels[m.group(1)]:i + 1]
    ops = collections.Counter(b.split()[0] for b in body)
    # (the emit compare is v_cmpx in the fast-emit body; canonical walks carry W more v_cmp_ne_u32_sdwa since round 3:
    # the lazy strand vote's position compare)
    emit_cmp = ops["v_cmpx_ne_u32_sdwa"] if ops["v_cmpx_ne_u32_sdwa"] else ops["v_cmp_ne_u32_sdwa"]
    writes = ops["ds_write_b16"] + ops["ds_write_b8"]
    if writes == 0 or writes % W or writes // W > 8 or emit_cmp != writes:
        continue
    if sum(v for o, v in ops.items() if o.startswith("ds_read")) != writes or not any(o.startswith("buffer_load") for o in ops):
        continue  # (the block loop proper: one table look-up and one list append per window, the sequence loads of a later group)
    if any(o.startswith(("v_cmp_lt_i32", "buffer_store", "global_store")) for o in ops):
        continue  # range-checked or direct-store walks
    # the hot loop is the one that walks the most windows per iteration (the group-unrolled one), then the shortest
    if best is None or writes > NWIN or (writes == NWIN and len(body) < len(best)):
        best, NWIN = body, writes
assert best, "main loop not found"


def wide_group(w):
    """kWideGroup<W> of mm_fused_impl.h (blocks per wide sequence load; 0 = per-block loads)."""
    def blocks(nd):
        if w % 16 == 0:
            return 0
        nsub, valid, m = (w + 15) // 16, 32 * nd - 24, 0
        while m < 8:
            if any((2 * w * m + 32 * g) % 32 > 24 or 2 * w * m + 32 * g + 6 + 2 * (min(16, w - 16 * g) + (1 if g == nsub - 1 else 0)) > valid
                   for g in range(nsub)):
                break
            m += 1
        return m
    m4, m5 = blocks(4), blocks(5)
    return m4 if (m4 > 0 and 2 * m4 >= m5) else m5


# With wide loads the block loop holds one view-extraction case per block of a group (a wave-uniform switch: ONE of
# them runs per iteration) and the group's start (buffer rotation, v_alignbyte, the loads: once per group).  The
# listing is split at its inner labels; pieces that are such a case or the group start count 1 / group.
MG = wide_group(W)
first = None
for i, s_ in enumerate(insts):
    if s_ is best[0] and insts[i:i + len(best)] == best:
        first = i
        break
inner = sorted(v - first for v in labels.values() if first is not None and first < v < first + len(best))
pieces, prev = [], 0
for cut in inner + [len(best)]:
    if cut > prev:
        pieces.append(best[prev:cut])
    prev = cut
cls = collections.Counter()
for piece in pieces:
    ops = [b.split()[0] for b in piece]
    valu = [o for o in ops if o.startswith("v_")]
    group_start = any(o.startswith(("v_alignbyte", "buffer_load")) for o in ops)
    view_case = bool(valu) and all(o.startswith("v_alignbit") for o in valu) and len(valu) <= 2 * ((W + 15) // 16)
    wgt = 1.0 / MG if (MG > 1 and NWIN == W and (group_start or view_case)) else 1.0  # (unrolled groups: every piece runs)
    for b in piece:
        cls[classify(b.split()[0], b)] += wgt
cls = collections.Counter({k: round(v, 2) for k, v in cls.items()})
if os.environ.get("CENSUS_HIST"):  # opcode histogram of the loop (unweighted), half-rate ones marked, + the listing
    hist = collections.Counter()
    for b in best:
        if b.startswith("v_"):
            hist[(b.split()[0], classify(b.split()[0], b))] += 1
    for (o, k_), v in sorted(hist.items(), key=lambda t: -t[1]):
        print(f"  {v:4d}  {'HALF' if k_ == 'valu_half' else 'full'}  {o}", file=sys.stderr)
    if os.environ["CENSUS_HIST"] == "2":
        print("\n".join(best), file=sys.stderr)
meta = {}
for ln in lines:
    for key in (".vgpr_count:", ".sgpr_count:", ".vgpr_spill_count:"):
        if key in ln:
            meta[key.strip(".:")] = int(ln.split(":")[1])
sys.path.insert(0, csrc)
from strip_comments import strip  # noqa: E402  (the hash is of the source without its comments, as in bench.py)
h = hashlib.sha256()
for f in ("mm_fused_impl.h", "mm_common.h"):
    h.update("\n".join(ln for ln in strip(open(os.path.join(csrc, f)).read()).split("\n") if ln.strip()).encode())  # (blank lines - stripped comment lines - do not count)
full, half = cls["valu_full"], cls["valu_half"]
# shader cycles per wave64 instruction (means over the instructions of each class in profiles/r03_valu_issue_rates.txt,
# "by wall time at that clock"); the architectural figures are 2 and 4 (MI355X_MICROARCH.md)
FULL, HALF = 2.31, 4.14
rec = {"kernel": f"mm::fused_kernel<{W}, {c}, {c}, 0, false, false>", "kernel_source_sha": h.hexdigest()[:16],
       "main_loop_windows": NWIN, "main_loop_instructions": len(best), "wide_group_blocks": MG,
       "valu_full_rate": full, "valu_half_rate": half,
       "salu": cls["salu"], "lds": cls["lds"], "vmem": cls["vmem"],
       "valu_per_window": round((full + half) / NWIN, 2),
       "issue_clk_per_valu": round((FULL * full + HALF * half) / (full + half), 3),
       "issue_clk_per_window": round((FULL * full + HALF * half) / NWIN, 1),
       "ideal_clk_per_valu": round((2.0 * full + 4.0 * half) / (full + half), 3),
       "rates": f"{FULL} / {HALF} shader cycles per full- / half-rate wave64 VALU instruction, measured with the shader "
                "clock read under each loop (profiles/r03_valu_issue_rates.txt); ideal_clk_per_valu uses the "
                "architectural 2 / 4",
       **meta}
out = os.path.join(ROOT, "profiles", "head_isa_census.json" if (W, CANON) == (11, True) else f"isa_census_w{W}_{'canon' if CANON else 'fwd'}.json")
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
