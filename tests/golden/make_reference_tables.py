"""Extracts reference-held CONSTANTS (data, not code) into tests/golden/reference_tables.json.  Dev container only
(reads /root/reference); the JSON travels with the repo.

  * numVertsTable[256]           src/kfusion/marching_cubes.cpp:358   -- vertices emitted per marching-cubes case
  * sha256 of triTable[256][16]  src/kfusion/marching_cubes.cpp:81-355 -- pins every entry of the case table without
                                 carrying it (the hash is over the 4096 entries as comma-joined decimal text, -1 = end)
  * the raw (un-normalised) Sobolev filter taps of every (s, lambda) the reference knows
                                 src/sobfu/solver.cpp:160-251, as the decimal literals written there

    python tests/golden/make_reference_tables.py
"""
import hashlib
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    mc = open(os.path.join(REF, "src/kfusion/marching_cubes.cpp")).read()
    m = re.search(r"numVertsTable\s*\[256\]\s*=\s*\{(.*?)\};", mc, re.S)
    num_verts = [int(v) for v in m.group(1).replace("\n", " ").split(",") if v.strip()]
    assert len(num_verts) == 256
    m = re.search(r"triTable\s*\[256\]\s*\[16\]\s*=\s*\{(.*?)\};", mc, re.S)
    rows = re.findall(r"\{([^{}]*)\}", m.group(1))
    tri = [int(v) for r in rows for v in r.replace("\n", " ").split(",") if v.strip()]
    assert len(tri) == 4096
    tri_sha = hashlib.sha256(",".join(str(v) for v in tri).encode()).hexdigest()

    sv = open(os.path.join(REF, "src/sobfu/solver.cpp")).read()
    body = sv[sv.index("decompose_sobolev_filter(SolverParams& params, float* h_S_i) {"):sv.index("/* normalise filter to unit sum */")]
    filters = []
    for ms in re.finditer(r"if \(params\.s == (\d+)\) \{(.*?)\n    \}\n", body, re.S):
        s = int(ms.group(1))
        for ml in re.finditer(r"if \(params\.lambda == ([0-9.]+)f\) \{(.*?)\}", ms.group(2), re.S):
            taps = [None] * s
            for ma in re.finditer(r"h_S_i\[(\d+)\]\s*=\s*([^;]+);", ml.group(2)):
                i, rhs = int(ma.group(1)), ma.group(2).strip()
                mr = re.fullmatch(r"h_S_i\[(\d+)\]", rhs)
                taps[i] = taps[int(mr.group(1))] if mr else rhs.rstrip("f")
            assert all(t is not None for t in taps)
            filters.append({"s": s, "lambda": ml.group(1), "raw_taps": taps})
    assert len(filters) == 8
    out = {"source": {"numVertsTable": "src/kfusion/marching_cubes.cpp:358", "triTable": "src/kfusion/marching_cubes.cpp:81-355",
                      "filters": "src/sobfu/solver.cpp:160-251"},
           "numVertsTable": num_verts, "triTable_sha256": tri_sha, "sobolev_filters": filters}
    with open(os.path.join(HERE, "reference_tables.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print("wrote reference_tables.json:", len(num_verts), "vertex counts,", len(filters), "filters, triTable sha", tri_sha[:16])


if __name__ == "__main__":
    main()
