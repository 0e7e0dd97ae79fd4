"""mm_run_batch_device: lane table (MM_LANE_TABLE=1) against per-sequence tiles (=0) over the contig length, 1 Gbp in all,
canonical k=21 w=11 and k=31 w=51; kernel ms by HIP events.  The policy takes the lane table below 8 default tiles per contig."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import torch
import simd_minimizers_amd as sm
ws = sm.default_workspace(0)
total = 1 << 30
d = sm.generate_device(total, 9)
out = torch.empty(int(total * 0.19), dtype=torch.int32, device="cuda")
for (k, w) in ((21, 11), (31, 51)):
    b = sm.canonical_minimizers(k, w)
    for ln in (10_000, 50_000, 200_000, 500_000, 1_000_000, 2_000_000, 8_000_000):
        n = total // ln
        seqs = [d[(i * ln) // 4:] for i in range(n)]
        lens = [ln] * n
        row = []
        for pol in ("1", "0", None):
            if pol is None: os.environ.pop("MM_LANE_TABLE", None)
            else: os.environ["MM_LANE_TABLE"] = pol
            for _ in range(3): sm.run_batch_device(b, seqs, lens, out)
            ws.enable_timing(True); ws.kernel_time(True)
            for _ in range(5): sm.run_batch_device(b, seqs, lens, out)
            ms, l = ws.kernel_time(True); ws.enable_timing(False)
            row.append(f"{'lane table' if pol == '1' else 'tiles' if pol == '0' else 'policy'} {ms / l:.4f} ms{' (lt)' if ws.last_lane_table() else ''}")
        os.environ.pop("MM_LANE_TABLE", None)
        print(f"k={k} w={w} {n:6d} x {ln:8d} bp: " + " | ".join(row), flush=True)
