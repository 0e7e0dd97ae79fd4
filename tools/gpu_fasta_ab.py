"""FASTA packer A/B on one box: the three-pass kernels (MM_FASTA_ONEPASS=0) against the one-pass kernel over lines (=1),
device time of mm_fasta_pack_device_async by events, on several shapes of 1 GiB of text made on the device; both
flavours must return the same packed bytes and record table."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import simd_minimizers_amd as sm
dev = torch.device("cuda:0")
ws = sm.Workspace(0, torch.cuda.current_stream(dev).cuda_stream)
L = sm.lib()
N = int(os.environ.get("MM_N", str(1 << 30)))

def make(n, width, crlf, rec_every):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    t = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")[torch.randint(0, 4, (n,), device="cuda", generator=g)]
    i = torch.arange(n, device="cuda")
    step = width + (2 if crlf else 1)
    t[i % step == step - 1] = 10
    if crlf: t[i % step == step - 2] = 13
    if rec_every: t[i % (step * rec_every) == 0] = ord(">")
    del i
    t[0] = ord(">")
    return t

def run(t, flav, reps=7):
    os.environ["MM_FASTA_ONEPASS"] = flav
    n = t.numel()
    packed = torch.zeros(n // 4 + 64, dtype=torch.uint8, device=dev)
    cap = 1 << 24
    rb = torch.zeros(cap + 1, dtype=torch.int64, device=dev); rp = torch.zeros(cap, dtype=torch.int64, device=dev)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    def step():
        sm._check(L.mm_fasta_pack_device_async(ws.h, C.c_void_p(t.data_ptr()), n, C.c_void_p(packed.data_ptr()),
                                               packed.numel() // 4 * 4, C.c_void_p(rb.data_ptr()), C.c_void_p(rp.data_ptr()),
                                               cap, C.c_void_p(cnt.data_ptr())))
    for _ in range(3): step()
    ms = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(reps):
        torch.cuda.synchronize(); e0.record(); step(); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    ws.check()
    nb, nr = int(cnt[0].item()), int(cnt[1].item())
    sig = (nb, nr, int(packed[: (nb + 3) // 4].to(torch.int64).sum().item()), int(rb[: nr + 1].sum().item()), int(rp[:nr].sum().item()),
           int((packed[: (nb + 3) // 4][:: 4099].to(torch.int64) * torch.arange(1, (((nb + 3) // 4) + 4098) // 4099 + 1, device=dev)).sum().item()))
    return sorted(ms)[len(ms) // 2], sig

shapes = (("60-base lines, LF, one record", (N, 60, False, 0)), ("60-base lines, CRLF", (N, 60, True, 0)),
          ("80-base lines, LF", (N, 80, False, 0)),
          ("reads: 150-base lines, every second line a header", (N, 150, False, 2)),
          ("one line", (N, N + 5, False, 0)), ("20-base lines (dense)", (N, 20, False, 0)))
for name, args in shapes:
    t = make(*args)
    a, sa = run(t, "0")
    b, sb = run(t, "1")
    print(f"{name}: three-pass {a:.3f} ms ({t.numel() / a / 1e6:.0f} GB/s) | one-pass over lines {b:.3f} ms ({t.numel() / b / 1e6:.0f} GB/s) | "
          f"{'SAME' if sa == sb else 'DIFFERENT ' + str(sa) + ' ' + str(sb)} bases {sa[0]} records {sa[1]}", flush=True)
    del t
