import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["MM_ENV_DYNAMIC"] = "1"
import torch
import simd_minimizers_amd as sm
import mm_oracle as oracle
rng = np.random.default_rng(123)
def expect(text):
    recs = oracle.fasta_records(text)
    seq = b"".join(s for _, _, s in recs)
    base = np.cumsum([0] + [len(s) for _, _, s in recs]).astype(np.uint64)
    return recs, seq, base, oracle.pack_ascii(seq)[: (len(seq) + 3) // 4]
def check(text, tag):
    recs, seq, base, packed = expect(text)
    got = sm.fasta_pack_device(text, max_records=max(16, len(recs) + 4))
    assert len(got) == len(recs), (tag, len(got), len(recs))
    assert np.array_equal(np.asarray(got.base), base), tag
    assert [int(x) for x in got.text_pos] == [p for p, _, _ in recs], tag
    gp = got.packed.cpu().numpy()[: len(packed)]
    if len(seq) % 4:
        gp = gp.copy(); gp[-1] &= (1 << (2 * (len(seq) % 4))) - 1
    assert np.array_equal(gp, packed), tag
acgt = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)
def seqbytes(n): return acgt[rng.integers(0, 9, n)].tobytes()
# 1. header longer than a group (4 MB), sequence line longer than a group, text before the first header longer than a group
big = 5_000_000
check(b">" + b"h" * big + b"\n" + seqbytes(1000) + b"\n>b\n" + seqbytes(77) + b"\n", "long header")
check(b">a\n" + seqbytes(big) + b"\n>b desc\n" + seqbytes(3) + b"\n", "long line")
check(seqbytes(big) + b"\n>a\n" + seqbytes(100) + b"\n", "long preamble")
check(b">a\n" + seqbytes(big), "no trailing newline, long line")
check(b">" + b"h" * big, "only a header, no newline")
check((b">r\n" + seqbytes(40) + b"\n") * 200000, "many short records (8.8 MB)")
# 2. random soup
alpha = np.frombuffer(b"ACGT>\n\r xN", dtype=np.uint8)
for it in range(300):
    n = int(rng.choice([1, 31, 32, 33, 8191, 8192, 8193, 16383, 16384, 16385, 40000, 300000, 4_200_000]))
    n = max(1, n + int(rng.integers(-3, 4)))
    w = rng.random(len(alpha)); w[4] *= rng.choice([0.01, 0.2, 1]); w[5] *= rng.choice([0.02, 0.3, 2]); w[6] *= rng.choice([0, 0.3])
    t = alpha[rng.choice(len(alpha), n, p=w / w.sum())].tobytes()
    check(t, ("soup", it, n))
print("fasta stress ok")
