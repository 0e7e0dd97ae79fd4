"""Quick on-GPU timing probe of the fused kernel (HIP events around the kernel)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import simd_minimizers_amd as sm

def run(n, k, w, canonical, mode, nblks, reps=5, seed=2):
    d = sm.generate_device(n, seed)
    ws = sm.default_workspace(0)
    out = torch.zeros(int(n * (2.2 / (w + 1))) + 1024, dtype=torch.int32, device="cuda")
    b = sm.Builder(k, w, canonical, mode)
    for nblk in nblks:
        ws.set_blocks_per_lane(nblk)
        try:
            cnt = b.run_device(d, n, out)  # warm-up
        except sm.MinimizerError as e:
            print(f"  nblk={nblk}: {e}"); continue
        ws.enable_timing(True); ws.kernel_time(True)
        for _ in range(reps):
            b.run_device(d, n, out, sync=False)
        ws.sync()
        ms, launches = ws.kernel_time(True)
        ws.enable_timing(False)
        t = ms / launches / 1e3
        bytes_alg = n / 4 + 4 * cnt
        print(f"k={k} w={w} canon={canonical} mode={mode} n={n} nblk={nblk} path={ws.last_path()} "
              f"count={cnt} {t*1e3:.3f} ms  {n/t/1e9:.1f} Gbase/s  {bytes_alg/t/1e9:.1f} GB/s alg", flush=True)
    ws.set_blocks_per_lane(0)
    del d, out

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 268435456
    run(n, 21, 11, False, 0, [0, 4, 8, 12, 16, 20, 24, 32])
    run(n, 21, 11, True, 0, [0, 4, 8, 12, 16, 20, 24, 32])
    run(n, 31, 51, True, 0, [0, 2, 3, 4, 6])
    run(n, 15, 17, True, 1, [0, 4, 8, 12])
