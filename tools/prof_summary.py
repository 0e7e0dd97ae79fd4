"""Condense rocprofv3 sqlite outputs (gpurun_out/prof_*/..db) into a small text summary for profiles/."""
import sqlite3, sys, glob, os

def summarize(path, out):
    db = sqlite3.connect(path); cur = db.cursor()
    out.write(f"== {path}\n")
    try:
        out.write("-- top_kernels (name, calls, total_us, avg_us, pct)\n")
        for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            out.write(f"{r[0][:90]:90s} calls={r[1]:4d} total_us={r[2]:12.1f} avg_us={r[3]:10.2f} pct={r[4]:5.1f}\n")
    except Exception as e:
        out.write(f"(no top_kernels: {e})\n")
    try:
        # the stats average includes the untimed warm-up launches (clock ramp after idle): list them
        d = [r[0] / 1e3 for r in cur.execute("select (end - start) from kernels where name like '%fused_kernel%' order by start")]
        if len(d) >= 4:
            half = d[len(d) // 2:]
            out.write("-- fused_kernel launches in order, us: " + " ".join(f"{x:.0f}" for x in d) + "\n")
            out.write(f"-- average of the second half ({len(half)} launches): {sum(half) / len(half):.1f} us\n")
            if len(d) >= 60:  # the stats pass runs 40 warm-up + 20 timed steps
                last = d[-20:]
                out.write(f"-- average of the last 20 launches (the timed steps of the stats pass): {sum(last) / len(last):.1f} us\n")
    except Exception as e:
        out.write(f"(no per-launch durations: {e})\n")
    try:
        rows = list(cur.execute("select kernel_name,counter_name,value,duration,grid_size,workgroup_size,lds_block_size,vgpr_count,sgpr_count from counters_collection"))
        if rows:
            out.write("-- counters (kernel, counter, value, duration_ns, grid, wg, lds, vgpr, sgpr)\n")
            for r in rows:
                if 'mm::' in r[0] and 'generate' not in r[0]:
                    out.write(f"{r[0][:60]:60s} {r[1]:24s} {r[2]:16.3f} dur_ns={r[3]} grid={r[4]} wg={r[5]} lds={r[6]} vgpr={r[7]} sgpr={r[8]}\n")
    except Exception as e:
        out.write(f"(no counters: {e})\n")

if __name__ == "__main__":
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    for p in sorted(glob.glob("gpurun_out/prof_*/*.db")):
        summarize(p, out)
