"""Condense `rocprofv3 --memory-copy-trace --kernel-trace --output-format csv` of tools/gpu_host_trace.py into a timeline of
the LAST mm_run_host call: every copy (direction, agents, start, end), every fused-kernel launch, and how much of the time
the two copy directions were in flight together.  Usage: trace_host_path.py <dir with *_memory_copy_trace.csv> [calls]"""
import csv, glob, os, sys
d = sys.argv[1]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
def rows(pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
cp = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_", ""), r["Source_Agent_Id"], r["Destination_Agent_Id"]) for r in rows("*memory_copy_trace.csv")]
krows = rows("*kernel_trace.csv")
kn = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in krows if "fused_kernel" in r["Kernel_Name"]]
# (round 5: the default download is a copy KERNEL that stores into the caller's page-locked buffer - it shows among the kernels)
ck = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in krows if "copy_range_kernel" in r["Kernel_Name"]]
cp.sort(); kn.sort()
if not kn:
    sys.exit("no fused-kernel launches in the trace")
per_call = len(kn) // calls
last = kn[-per_call:]
t0 = min(last[0][0], min((c[0] for c in cp if c[0] >= last[0][0] - 5_000_000), default=last[0][0]))
t1 = max([k[1] for k in last] + [c[1] for c in cp if c[0] >= t0])

t1 = max([t1] + [c[1] for c in ck if c[0] >= t0])
ev = [(c[0], c[1], f"copy {c[2]:<16} {c[3]} -> {c[4]}") for c in cp if c[0] >= t0] + [(k[0], k[1], "fused kernel") for k in last] + \
     [(c[0], c[1], "copy kernel: positions -> the caller's page-locked buffer") for c in ck if c[0] >= t0]
ev.sort()
print(f"last call: {per_call} chunks, {(t1 - t0) / 1e6:.2f} ms from the first copy / kernel to the last end")
for s, e, what in ev:
    print(f"  {(s - t0) / 1e6:8.3f} .. {(e - t0) / 1e6:8.3f} ms  ({(e - s) / 1e6:7.3f})  {what}")
def busy(direction):
    iv = sorted((c[0], c[1]) for c in cp if c[0] >= t0 and c[2] == direction)
    return iv
def total(iv):
    return sum(e - s for s, e in iv)
def overlap(a, b):
    o = 0
    for s1, e1 in a:
        for s2, e2 in b:
            o += max(0, min(e1, e2) - max(s1, s2))
    return o
h2d, d2h = busy("HOST_TO_DEVICE"), busy("DEVICE_TO_HOST")
ckb = sorted((c[0], c[1]) for c in ck if c[0] >= t0)
if ckb:
    print(f"copy kernels (device -> host): {total(ckb) / 1e6:.2f} ms in {len(ckb)} launches, in flight together with the uploads "
          f"{overlap(h2d, ckb) / 1e6:.2f} ms; first download starts at {(ckb[0][0] - t0) / 1e6:.2f} ms, last ends at {(ckb[-1][1] - t0) / 1e6:.2f} ms")
print(f"copy engines busy: H2D {total(h2d) / 1e6:.2f} ms in {len(h2d)} copies, D2H {total(d2h) / 1e6:.2f} ms in {len(d2h)} copies, "
      f"both directions in flight together {overlap(h2d, d2h) / 1e6:.2f} ms")
