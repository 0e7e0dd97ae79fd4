"""Per-kernel durations and the walk / expand overlap from a rocprofv3 kernel trace (rocpd .db; last launches)."""
import sqlite3, sys, collections, glob, os
db = sys.argv[1]
if os.path.isdir(db):
    db = sorted(glob.glob(os.path.join(db, "**", "*.db"), recursive=True))[0]
cur = sqlite3.connect(db).cursor()
by = collections.defaultdict(list)
for name, s, e in cur.execute("select name, start, end from kernels order by start"):
    key = "walk" if "walk_kernel" in name else "expand" if "expand_kernel" in name else "fused" if "fused_kernel" in name else name[:30]
    by[key].append((s, e))
for name, v in by.items():
    tail = v[len(v) // 2:]
    print(f"{name:30s} n={len(v):3d} mean {sum(e - s for s, e in tail) / len(tail) / 1e3:9.1f} us (second half of the launches)")
if "walk" in by and "expand" in by:
    w, e = by["walk"], by["expand"]
    for (ws, we), (es, ee) in list(zip(w, e))[-4:]:
        print(f"  walk {(we - ws) / 1e3:8.1f} us | expand starts {(es - ws) / 1e3:7.1f} us after the walk starts, ends {(ee - we) / 1e3:7.1f} us after it ends | span {(max(we, ee) - ws) / 1e3:8.1f} us")
