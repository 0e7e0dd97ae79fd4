"""CPU models of two device-side algorithms of round 3, checked against brute force (no GPU, no oracle):

* the two-stacks sliding minimum with a SPARSE suffix stack (``ring_turn`` / ``ring_step`` in
  simd-minimizers_amd/csrc/mm_fused_impl.h; reference semantics src/sliding_min.rs:86-212): suffix minima kept at
  every second ring element only, every regrouping of the three-operand minima returns the window minimum;
* the LAZY strand vote (``decide`` in the same file; src/canonical.rs:12-31): the T|G count of a window rebuilt from the
  count at the block's start and the T|G bits of the block's entering / leaving bases below the step.

The models follow the device code statement by statement, so that a change there has a place to be mirrored here."""
import random

import pytest


def ring_turn(ring, W, op):
    for e in range(W - 3, 0, -2):
        ring[e] = op(ring[e], ring[e + 1], ring[e + 2])


def ring_step(ring, state, key, J, W, op):
    """One step of a block: returns the window minimum; state = [prefix minimum]."""
    kept_next = ((J ^ W) & 1) == 0
    if kept_next:
        sel = op(key, ring[J + 1]) if J == 0 else op(state[0], key, ring[J + 1])
    else:
        if J == 0:
            state[0] = key
        elif J == 1:
            state[0] = op(ring[0], key)
        else:
            state[0] = op(state[0], ring[J - 1], key)
        sel = op(state[0], ring[J + 1], ring[J + 2]) if J + 1 < W else state[0]
    ring[J] = key
    return sel


@pytest.mark.parametrize("W", list(range(1, 41)) + [51, 64])
@pytest.mark.parametrize("right", [False, True])
def test_sparse_suffix_sliding_minimum(W, right):
    rng = random.Random(W * 2 + right)
    op = max if right else min
    n_blocks = 9
    # keys carry their position in the low bits, as on the device: (hash16 << 16) | position
    keys = [(rng.randrange(0, 8) << 16) | i for i in range(W * n_blocks)]  # (few hash values: many ties of the hash)
    if right:
        keys = [k ^ 0xFFFF0000 for k in keys]
    # warm-up: the first block fills the ring, then it is turned
    ring = keys[:W]
    ring_turn(ring, W, op)
    got, exp = [], []
    for b in range(1, n_blocks):
        state = [None]
        for J in range(W):
            e = b * W + J
            got.append(ring_step(ring, state, keys[e], J, W, op))
            exp.append(op(keys[e - W + 1: e + 1]))
        ring_turn(ring, W, op)
    assert got == exp


@pytest.mark.parametrize("W,k", [(11, 21), (12, 20), (17, 15), (25, 21), (33, 31), (35, 31), (51, 31), (64, 33)])
def test_lazy_strand_vote_count(W, k):
    rng = random.Random(W)
    l = k + W - 1
    n = l + W * 7 + 3
    seq = [rng.randrange(4) for _ in range(n)]
    tg = [(c >> 1) & 1 for c in seq]  # T|G = the high bit of the 2-bit code (A C T G = 0 1 2 3)
    thr = l // 2

    def eager(i):  # window i covers bases [i, i + l): canonical iff more than half are T|G
        return sum(tg[i: i + l]) - thr - 1

    # the device keeps dn for the first window of a block and, per block, the T|G bits of the bases that enter
    # (base i + l for the step from window i to i + 1) and leave (base i); view words hold 16 bases, bit 2j+1 = T|G
    nsub = (W + 15) // 16
    dn = eager(0)
    for b in range(0, (n - l) // W - 1):
        w0 = b * W
        xt, yt = [0] * nsub, [0] * nsub
        for j in range(W):
            g, jj = j >> 4, j & 15
            xt[g] |= tg[w0 + j + l] << (2 * jj + 1)
            yt[g] |= tg[w0 + j] << (2 * jj + 1)
        for j in range(W):
            d = dn
            for g in range((j >> 4) + 1):
                nb = j - 16 * g
                m = 0xAAAAAAAA if nb >= 16 else (0xAAAAAAAA & ((1 << (2 * max(nb, 0))) - 1))
                d += bin(xt[g] & m).count("1") - bin(yt[g] & m).count("1")
            assert d == eager(w0 + j), (b, j)
        dn += sum(bin(x).count("1") for x in xt) - sum(bin(y).count("1") for y in yt)
