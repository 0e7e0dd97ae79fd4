"""FASTA text -> PackedSeq records on the device (mm_fasta_pack_device, the loader step in front of the path:
needletail::parse_fastx_file + PackedSeqVec::from_ascii per record in the reference's harness,
bench/src/lib.rs:51-82), bit-exact against the oracle's restatement of that reader, and end to end through
mm_run_batch_device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["three-pass", "one-pass", "two-pass"])
def packer_flavour(request, monkeypatch):
    """Every test runs with all three packers: the two passes of mask arithmetic (mm_fasta2.hip, the default since late
    round 4: no environment switch), the one-pass kernel over lines (MM_FASTA_ONEPASS=1: the text read once, one decoupled
    look-back between chunks) and the three-pass kernels (MM_FASTA_ONEPASS=0; also what the one-pass kernel falls back
    to for texts whose lines are shorter than 16 bytes on average - DESIGN.md 4.3a)."""
    if request.param == "two-pass":
        monkeypatch.delenv("MM_FASTA_ONEPASS", raising=False)
        monkeypatch.delenv("MM_FASTA_KERNEL", raising=False)
    else:
        # (round 5: the two older packers are cross-checks in the EXPERIMENTS build only - these flavours run in the
        # child pytest of tests/test_gpu_round5.py::test_cross_checks_in_the_experiments_build)
        request.getfixturevalue("exp_build")
        monkeypatch.setenv("MM_FASTA_ONEPASS", "1" if request.param == "one-pass" else "0")


def expect(oracle, text):
    recs = oracle.fasta_records(text)
    seq = b"".join(s for _, _, s in recs)
    base = np.cumsum([0] + [len(s) for _, _, s in recs]).astype(np.uint64)
    return recs, seq, base, oracle.pack_ascii(seq)[: (len(seq) + 3) // 4]


def check(sm, oracle, text, **kw):
    recs, seq, base, packed = expect(oracle, text)
    got = sm.fasta_pack_device(text, **kw)
    assert len(got) == len(recs), (len(got), len(recs))
    assert np.array_equal(got.base, base)
    assert [int(p) for p in got.text_pos] == [p for p, _, _ in recs]
    assert np.array_equal(got.packed[: len(packed)].cpu().numpy(), packed)
    for i, (_, h, _) in enumerate(recs):
        assert got.header(bytes(text), i) == h
    return got


def test_fasta_known_cases(sm, oracle, gpu):
    for text in [
        b">chr1 test\nACGT\nAC\n",
        b">a\r\nACGT\r\nGG\r\n>b\r\nTT",            # CRLF, no newline at the end
        b"junk before\n>chr1\nACGTNNNN\n\n\nacgt\n>c2\n>c3\nGG>T\nA",  # blank lines, empty record, '>' inside a line
        b"",
        b"\n",
        b"ACGT\nACGT\n",                            # no header at all: nothing
        b">only a header",
        b">h\n",
        b">\n>\n>\nA",
        b">x\n" + b"ACGT" * 5000,                   # one long line
        b">" + b"h" * 100000 + b"\nACGT\n",         # a header longer than three chunks
        b">x\n" + b"\n".join(b"ACGTTGCA"[: 1 + i % 8] for i in range(9000)) + b"\n>y\nGATTACA\n",
        b">abcdefghijklmnopqrst\n" * 3000 + b"ACGT\n",   # more record starts in a chunk than the one-pass table holds
        b">r\n" + (b"ACGTACGTACGTACGTACGTACGTACGTAC\r\n" * 4000),  # CRLF, 30-base lines: ~1000 separators per chunk
    ]:
        check(sm, oracle, text)


def test_fasta_control_characters(sm, oracle, gpu):
    """Bytes below 0x0E that are neither '\\n' nor '\\r' (TAB, NUL, VT ...) are candidates for the one-pass packer's
    separator list but stay what they are for the reader: header text or sequence bytes."""
    rng = np.random.default_rng(99)
    for text in [
        b">a\tb\x00c\nAC\tGT\x0bAC\n\x0cGG\n>b\n\t\n",
        b">x\n" + b"".join(bytes([int(c)]) for c in rng.choice(list(b"ACGT\t\x00\x01\x0b\x0c\x0d\n"), size=40000)),
        b"\t\n>r\x0d\x0d\nA\x0dC\x0d\x0aG\n",
        b">x\n" + (b"ACGT" * 300 + b"\x0b") * 40 + b"\n",
    ]:
        check(sm, oracle, text)


def test_fasta_short_reads(sm, oracle, gpu):
    """A reads file: every record one header line and one sequence line (two separators per ~200 bytes)."""
    rng = np.random.default_rng(5)
    parts = []
    for i in range(6000):
        m = int(rng.integers(30, 160))
        parts.append(b">read%d len=%d\n" % (i, m) + rng.choice(list(b"ACGT"), size=m).astype(np.uint8).tobytes() + b"\n")
    got = check(sm, oracle, b"".join(parts))
    assert len(got) == 6000


def random_fasta(rng, n_target):
    out = bytearray()
    if rng.random() < 0.2:
        out += b"preamble\n"
    while len(out) < n_target:
        out += b">" + bytes(rng.choice(list(b"abc XYZ>09|"), size=int(rng.integers(0, 40))).astype(np.uint8)) + (
            b"\r\n" if rng.random() < 0.2 else b"\n")
        m = int(rng.choice([0, 1, 5, 70, 4000, 20000, 70000]))
        m = int(rng.integers(0, m + 1))
        seq = rng.choice(list(b"ACGTacgtN"), size=m).astype(np.uint8).tobytes()
        width = int(rng.choice([1, 7, 60, 61, 80, 1000, 10 ** 6]))
        nl = b"\r\n" if rng.random() < 0.2 else b"\n"
        lines = [seq[i:i + width] for i in range(0, m, width)]
        out += nl.join(lines)
        if rng.random() < 0.9:
            out += nl
        if rng.random() < 0.1:
            out += b"\n\n"
    return bytes(out)


def test_fasta_state_across_groups(sm, oracle, gpu):
    """The header / record state carried over more than a GROUP of 256 chunks (4 MB; mm_fasta2.hip composes the chunks'
    state functions per group and then the groups): a header line, a sequence line and the text in front of the first
    header longer than that, a text that is one header without a newline, and 8.8 MB of 40-base records."""
    rng = np.random.default_rng(31)
    acgt = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)

    def seq(n):
        return acgt[rng.integers(0, 9, n)].tobytes()
    big = 4_500_000
    check(sm, oracle, b">" + b"h" * big + b"\n" + seq(1000) + b"\n>b\n" + seq(77) + b"\n")
    check(sm, oracle, b">a\n" + seq(big) + b"\n>b desc\n" + seq(3) + b"\n")
    check(sm, oracle, seq(big) + b"\n>a\n" + seq(100) + b"\n")
    check(sm, oracle, b">a\n" + seq(big))
    check(sm, oracle, b">" + b"h" * big)
    check(sm, oracle, (b">r\n" + seq(40) + b"\n") * 200_000, max_records=1 << 18)


def test_fasta_random(sm, oracle, gpu):
    rng = np.random.default_rng(2024)
    sizes = [1, 15, 16, 17, 4095, 4096, 4097, 16383, 16384, 16385, 32767, 32768, 32769, 65536 + 3, 200_000, 1_000_000]
    for i in range(64):
        text = random_fasta(rng, sizes[i % len(sizes)])
        cut = int(rng.integers(0, 3))
        if cut == 1 and len(text) > 8:  # end somewhere inside
            text = text[: int(rng.integers(1, len(text)))]
        check(sm, oracle, text)


def test_fasta_unaligned_text_and_limits(sm, oracle, gpu):
    import torch
    rng = np.random.default_rng(7)
    text = random_fasta(rng, 100_000)
    buf = torch.zeros(len(text) + 64, dtype=torch.uint8, device="cuda")
    recs, seq, base, packed = expect(oracle, text)
    for off in (1, 3, 8, 15):
        buf[off: off + len(text)] = torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).cuda()
        got = sm.fasta_pack_device(buf[off: off + len(text)])
        assert np.array_equal(got.base, base)
        assert np.array_equal(got.packed[: len(packed)].cpu().numpy(), packed)
    with pytest.raises(sm.MinimizerError):
        sm.fasta_pack_device(b">a\nA\n>b\nC\n>c\nG\n", max_records=2)


def test_fasta_to_minimizers(sm, oracle, gpu):
    """FASTA -> packed records -> one batch launch: record-local canonical minimizers == the oracle per record."""
    import torch
    rng = np.random.default_rng(11)
    lens = [250_000, 0, 1, 40, 41, 99_999, 1_000_003, 7]
    parts = []
    for i, m in enumerate(lens):
        seq = rng.choice(list(b"ACGT"), size=m).astype(np.uint8).tobytes()
        parts.append(b">contig%d some text\n" % i + b"\n".join(seq[j:j + 60] for j in range(0, m, 60)) + b"\n")
    text = b"".join(parts)
    recs = sm.fasta_pack_device(text)
    assert recs.lengths() == lens
    k, w = 21, 11
    b = sm.canonical_minimizers(k, w)
    out = torch.zeros(sum(lens) // 4 + 1024, dtype=torch.int32, device="cuda")
    offs = sm.run_fasta_device(b, recs, out)
    host = out.cpu().numpy().view(np.uint32)
    for i, (_, _, seq) in enumerate(oracle.fasta_records(text)):
        exp = oracle.run(oracle.pack_ascii(seq), len(seq), k, w, canonical=True)
        assert np.array_equal(host[offs[i]: offs[i + 1]], exp), i


def test_fasta_large_against_ascii_pack(sm, oracle, gpu, request):
    """300 MB of text (9 000 chunks: the chunk scans run over many groups per wave): the packed records equal
    mm_pack_ascii of the sequence bytes selected on the device with torch, and the record table matches the
    construction.  The default packer takes 1.15 GiB instead: more than 256 groups of 256 chunks, the second round of
    its resolve step."""
    import ctypes as C
    import torch
    two_pass = "two-pass" in request.node.name
    n, width, n_rec = (1_234_567_891 if two_pass else 300_000_000), 70, 24
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    t = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")[torch.randint(0, 4, (n,), device="cuda", generator=g)]
    i = torch.arange(n, device="cuda")
    t[i % (width + 1) == width] = 10
    del i
    keep = t != 10
    starts = []
    for r in range(n_rec):
        p = (n // n_rec) * r + (r * 7919) % 50
        hdr = b">record %d\n" % r
        if p:
            t[p - 1] = 10
            keep[p - 1] = False
        t[p: p + len(hdr)] = torch.tensor(list(hdr), dtype=torch.uint8, device="cuda")
        keep[p: p + len(hdr)] = False
        starts.append(p)
    keep &= t != 10
    seq = t[keep]
    m = int(seq.numel())
    exp = torch.zeros((m + 3) // 4 + 64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    sm._check(sm.lib().mm_pack_ascii_device_async(gpu.h, C.c_void_p(seq.data_ptr()), m, C.c_void_p(exp.data_ptr())))
    gpu.sync()
    got = sm.fasta_pack_device(t)
    assert len(got) == n_rec and int(got.base[-1]) == m
    assert [int(p) for p in got.text_pos] == starts
    before = torch.cumsum(keep.to(torch.int64), 0)
    assert [int(b) for b in got.base[:-1]] == [int(before[p].item()) for p in starts]
    assert torch.equal(got.packed[: (m + 3) // 4], exp[: (m + 3) // 4])
