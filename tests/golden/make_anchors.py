"""Regenerates tests/golden/model_anchors.json with the CPU oracle (oracle/mm_oracle.c).

The anchors are NOT reference outputs (the Rust reference cannot run in this image); they
pin the oracle against the independent numpy model of SURVEY.md §8c, whose numbers are
recorded in SURVEY.md ("Synthetic-input anchors [MODEL]"), so that any later divergence of
the restatement is caught.  Run from the repo root:  python tests/golden/make_anchors.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mm_oracle as o  # noqa: E402

N, SEED = 1_000_000, 1
CASES = [
    ("fwd k=21 w=11", 21, 11, False, 0),
    ("canonical k=21 w=11", 21, 11, True, 0),
    ("canonical k=31 w=51", 31, 51, True, 0),
    ("fwd k=5 w=7", 5, 7, False, 0),
    ("canonical closed syncmers k=15 w=17", 15, 17, True, 1),
    ("canonical open syncmers k=15 w=17", 15, 17, True, 2),
]


def main():
    g = o.gen_packed(SEED, N)
    out = {"generator": {"seed": SEED, "n": N,
                         "first32": "".join("ACTG"[(g[i >> 2] >> (2 * (i & 3))) & 3] for i in range(32))},
           "cases": []}
    for name, k, w, canon, mode in CASES:
        r = o.run(g, N, k, w, canonical=canon, mode=mode)
        cw, cp = o.checksum(r)
        out["cases"].append({"name": name, "k": k, "w": w, "canonical": canon, "mode": mode,
                             "count": int(len(r)), "first8": [int(x) for x in r[:8]],
                             "checksum_weighted": cw, "checksum_plain": cp})
    h = o.default_hasher(False)
    seq = b"ACGTGCTCAGAGACTCAG"
    out["fwd_hashes_k5"] = {"seq": seq.decode(),
                            "hashes": ["%08x" % x for x in o.hash_kmers(o.pack_ascii(seq), len(seq), 5, h)]}
    h = o.default_hasher(True)
    seq = b"ACGTGCTCAGAGACTCAGAGGA"
    out["canonical_hashes_k5"] = {"seq": seq.decode(),
                                  "hashes": ["%08x" % x for x in o.hash_kmers(o.pack_ascii(seq), len(seq), 5, h)]}
    with open(os.path.join(ROOT, "tests", "golden", "model_anchors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote model_anchors.json")


if __name__ == "__main__":
    main()
