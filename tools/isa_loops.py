#!/usr/bin/env python3
"""Per-loop instruction census of a gfx950 assembly listing (hipcc -S --cuda-device-only).

usage: isa_loops.py kernel.s [min_instructions]
For every backward branch (loop) prints the number of instructions by class and, for the VALU
ones, the split into full-rate and half-rate opcodes as measured on MI355X
(profiles/r01_valu_issue_rates.txt).  A measurement aid, not part of the product."""
import collections
import re
import sys

HALF = ("v_min", "v_max", "v_alignbit", "v_bfe", "v_cmp", "v_cndmask", "v_and_or", "v_lshl_add", "v_lshl_or",
        "v_mad", "v_mul", "v_perm", "v_bfi", "v_lshlrev", "v_add3", "v_xad", "v_pk_", "v_readlane",
        "v_writelane", "v_readfirstlane", "v_or3", "v_med3", "v_sad", "v_bcnt", "v_mbcnt", "v_ffb",
        "v_cvt", "v_alignbyte", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")


def classify(op, text):
    if op.startswith("v_"):
        half = op.startswith(HALF) or "sdwa" in op or "_dpp" in op
        # an SGPR source operand halves the rate of any VALU instruction
        ops = text.split(None, 1)[1] if " " in text else ""
        srcs = ops.split(",")[1:]
        if any(re.match(r"\s*s\d+|\s*s\[", s) for s in srcs) and not op.startswith(("v_readlane", "v_writelane")):
            half = True
        return "valu_half" if half else "valu_full"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path = sys.argv[1]
    min_n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    lines = open(path).read().split("\n")
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
    for i, s in enumerate(insts):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", s) or re.match(r"s_branch\s+(\.LBB\d+_\d+)", s)
        if not m or m.group(1) not in labels:
            continue
        j = labels[m.group(1)]
        if j > i or i - j < min_n:
            continue
        body = insts[j:i + 1]
        cls = collections.Counter(classify(b.split()[0], b) for b in body)
        ops = collections.Counter(b.split()[0] for b in body if b.startswith("v_"))
        print(f"loop {m.group(1)}: {len(body)} instructions  " + "  ".join(f"{k}={v}" for k, v in sorted(cls.items())))
        print("   VALU by opcode: " + ", ".join(f"{k}:{v}" for k, v in ops.most_common()))
        clk = 2.5 * cls["valu_full"] + 4.3 * cls["valu_half"]
        print(f"   VALU issue estimate: {clk:.0f} clk per wave-iteration")


if __name__ == "__main__":
    main()
