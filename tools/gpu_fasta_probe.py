"""FASTA text -> packed records on the device (mm_fasta_pack_device_async): time for a 1 GB multi-record text with
60-base lines, and for the same bases on one line per record."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm
L = sm.lib(); ws = sm.default_workspace(0)
dev = "cuda"
def make(n, width, n_rec):
    g = torch.Generator(device=dev); g.manual_seed(1)
    t = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (n,), device=dev, generator=g)]
    i = torch.arange(n, device=dev)
    if width: t[i % (width + 1) == width] = 10
    for r in range(n_rec):
        p = (n // n_rec) * r
        hdr = b">record%d\n" % r
        if p: t[p - 1] = 10
        t[p: p + len(hdr)] = torch.tensor(list(hdr), dtype=torch.uint8, device=dev)
    return t
for name, width in (("60-base lines", 60), ("one line per record", 0)):
    n = 1 << 30
    t = make(n, width, 24)
    packed = torch.empty(n // 4 + 64, dtype=torch.uint8, device=dev)
    rb = torch.zeros(1025, dtype=torch.int64, device=dev); rp = torch.zeros(1024, dtype=torch.int64, device=dev)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    def step():
        sm._check(L.mm_fasta_pack_device_async(ws.h, C.c_void_p(t.data_ptr()), n, C.c_void_p(packed.data_ptr()),
                                               packed.numel() // 4 * 4, C.c_void_p(rb.data_ptr()), C.c_void_p(rp.data_ptr()),
                                               1024, C.c_void_p(cnt.data_ptr())))
    for _ in range(3): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    b, r = [int(x) for x in cnt.cpu()]
    print(f"{name}: {n / 2**30:.0f} GiB text, {b} bases, {r} records: {ms:.3f} ms = {n / ms / 1e6:.0f} GB/s of text, "
          f"{b / ms / 1e6:.0f} Gbases/s", flush=True)
    del t, packed
