"""Lazy strand vote against the per-window vote (-DMM_VOTE_EAGER) on sequences that tie often: two-letter alphabets,
short tandem repeats with mutations, homopolymer runs; canonical minimizers and syncmers of several (k, w), window ranges
that end inside tiles (partial walks).  Both kernels through the run-time specialisation; outputs must be identical."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import simd_minimizers_amd as sm
dev = torch.device("cuda:0")
ws = sm.default_workspace(0)
L = sm.lib()
g = torch.Generator(device="cuda"); g.manual_seed(3)
def pack(ascii_t):
    n = ascii_t.numel()
    p = torch.zeros((n + 3) // 4 + 64, dtype=torch.uint8, device=dev)
    sm._check(L.mm_pack_ascii_device_async(ws.h, C.c_void_p(ascii_t.data_ptr()), n, C.c_void_p(p.data_ptr())))
    ws.sync()
    return p
def seqs(n):
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    yield "two letters A/C", torch.tensor(list(b"AC"), dtype=torch.uint8, device=dev)[torch.randint(0, 2, (n,), device=dev, generator=g)]
    yield "two letters A/T", torch.tensor(list(b"AT"), dtype=torch.uint8, device=dev)[torch.randint(0, 2, (n,), device=dev, generator=g)]
    unit = acgt[torch.randint(0, 4, (37,), device=dev, generator=g)]
    rep = unit.repeat(n // 37 + 1)[:n].clone()
    mut = torch.randint(0, 200, (n,), device=dev, generator=g) == 0
    rep[mut] = acgt[torch.randint(0, 4, (int(mut.sum().item()),), device=dev, generator=g)]
    yield "37-base repeat, 0.5 % mutations", rep
    r = acgt[torch.randint(0, 4, (n,), device=dev, generator=g)]
    i = torch.arange(n, device=dev)
    r[(i // 5000) % 3 == 0] = ord("A")
    yield "random with 5 kbp poly-A runs", r
n = int(os.environ.get("MM_N", "200000000"))
os.environ["MM_JIT_FORCE"] = "1"
bad = 0
for name, a in seqs(n):
    d = pack(a)
    out = torch.zeros(n // 2 + 1024, dtype=torch.int32, device=dev)
    for (k, w, mode) in ((21, 11, 0), (20, 12, 0), (15, 17, 1), (15, 17, 2), (31, 33, 0), (31, 51, 0)):
        b = sm.Builder(k, w, True, mode)
        res = []
        for defs in ("-DMM_VOTE_EAGER", "-DMM_X=1"):
            os.environ["MM_JIT_DEFS"] = defs
            sig = []
            for (wb, we) in ((0, None), (12345, n // 3 + 777)):
                out.zero_()
                try:
                    c = b.run_device(d, n, out, win_begin=wb, win_end=we) if we else b.run_device(d, n, out)
                    v = out[:c].to(torch.int64)
                    sig.append((c, int((v * torch.arange(1, c + 1, device=dev)).sum().item())))
                except sm.MinimizerError as e:
                    sig.append(("error", str(e)[:40]))
            res.append(sig)
        ok = res[0] == res[1]
        bad += 0 if ok else 1
        print(f"{name}: k={k} w={w} mode={mode}: {'same' if ok else 'DIFFERENT'} {res[1]}", flush=True)
print(f"{bad} differences")
