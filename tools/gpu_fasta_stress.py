"""Stress of the one-pass FASTA packer against the three-pass kernels: random texts made on the device (random line
widths, CRLF or LF, random header lines, control characters, random lengths around chunk multiples), both packers'
packed bytes, record tables and counts compared; repeated.  Prints the number of comparisons and any difference."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import simd_minimizers_amd as sm
dev = torch.device("cuda:0")
ws = sm.Workspace(0, torch.cuda.current_stream(dev).cuda_stream)
L = sm.lib()
g = torch.Generator(device="cuda"); g.manual_seed(int(os.environ.get("SEED", "7")))
def rnd(lo, hi): return int(torch.randint(lo, hi, (1,), device="cuda", generator=g).item())
def make(n):
    t = torch.tensor(list(b"ACGTacgtNn"), dtype=torch.uint8, device="cuda")[torch.randint(0, 10, (n,), device="cuda", generator=g)]
    i = torch.arange(n, device="cuda")
    width = [7, 20, 31, 60, 61, 70, 80, 150, 1000, 100000][rnd(0, 10)]
    crlf = rnd(0, 4) == 0
    step = width + (2 if crlf else 1)
    t[i % step == step - 1] = 10
    if crlf: t[i % step == step - 2] = 13
    every = [0, 1, 2, 3, 17, 1000][rnd(0, 6)]
    if every: t[i % (step * every) == 0] = ord(">")
    if rnd(0, 3) == 0:  # stray control characters and '>' inside lines
        m = torch.randint(0, 5000, (n,), device="cuda", generator=g) == 0
        t[m] = torch.tensor([9, 0, 11, 62, 13], dtype=torch.uint8, device="cuda")[torch.randint(0, 5, (int(m.sum().item()),), device="cuda", generator=g)]
    if rnd(0, 4): t[0] = ord(">")
    return t
def run(t, flav):
    os.environ["MM_FASTA_ONEPASS"] = flav
    n = t.numel()
    packed = torch.zeros(n // 4 + 64, dtype=torch.uint8, device=dev)
    cap = n // 2 + 2
    rb = torch.zeros(cap + 1, dtype=torch.int64, device=dev); rp = torch.zeros(cap, dtype=torch.int64, device=dev)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    out = (C.c_uint64 * 2)()
    sm._check(L.mm_fasta_pack_device(ws.h, C.c_void_p(t.data_ptr()), n, C.c_void_p(packed.data_ptr()), packed.numel() // 4 * 4,
                                     C.c_void_p(rb.data_ptr()), C.c_void_p(rp.data_ptr()), cap, C.c_void_p(cnt.data_ptr()), out))
    nb, nr = int(out[0]), int(out[1])
    return nb, nr, packed[: (nb + 3) // 4].clone(), rb[: nr + 1].clone(), rp[:nr].clone()
bad = 0
reps = int(os.environ.get("REPS", "150"))
for it in range(reps):
    base = [16384, 65536, 1 << 20, 1 << 24, 1 << 26][rnd(0, 5)]
    n = max(1, base * rnd(1, 4) + rnd(-40, 40))
    t = make(n)
    a, b = run(t, "0"), run(t, "1")
    same = a[0] == b[0] and a[1] == b[1] and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    if not same:
        bad += 1
        print(f"DIFFERENT at iteration {it}: n={n} bases {a[0]} / {b[0]} records {a[1]} / {b[1]}", flush=True)
print(f"{reps} texts, {bad} differences")
