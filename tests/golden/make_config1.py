"""Regenerates tests/golden/config1.json: BASELINE.json config 1 on its literal input - forward minimizer_positions,
k=5 w=7, the first 1 000 bases of generator G (seed 1) as the ASCII string "ACTG"[code].

Like model_anchors.json this is NOT a reference output (the Rust reference cannot run here): it pins the oracle's
answer for the literal configuration so that the scalar CPU path, the ASCII entry points and the HIP kernel are all
checked against one committed vector.  Run from the repo root:  python tests/golden/make_config1.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mm_oracle as o  # noqa: E402

N, SEED, K, W = 1000, 1, 5, 7
g = o.gen_packed(SEED, N)
ascii_seq = "".join("ACTG"[(g[i >> 2] >> (2 * (i & 3))) & 3] for i in range(N))
naive = o.run(g, N, K, W, canonical=False, flavour=o.NAIVE)
stream = o.run(g, N, K, W, canonical=False, flavour=o.STREAMING)
assert list(naive) == list(stream)
json.dump({"config": "BASELINE.json configs[0]: forward minimizer_positions, k=5 w=7, 1 kb ASCII sequence",
           "generator": {"seed": SEED, "n": N}, "k": K, "w": W, "ascii": ascii_seq,
           "positions": [int(x) for x in naive]},
          open(os.path.join(ROOT, "tests", "golden", "config1.json"), "w"), indent=1)
print(len(naive), "positions")
