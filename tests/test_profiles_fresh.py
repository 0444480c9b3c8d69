"""profiles/pmc_latest.json -- the counter file bench.py quotes as roofline.traffic_from_profiles -- must have been collected on the
kernels that are in the tree: tools/profile_round.sh stamps it with the SHA-256 of sobfu_amd/csrc/solver_kernels.hip + the solver_*.inl parts it includes, and this test
fails when the source has changed since (re-run tools/profile_round.sh on a GPU box and commit the new profiles/)."""
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pmc_profile_matches_kernel_source():
    with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
        pj = json.load(f)
    d = os.path.join(ROOT, "sobfu_amd", "csrc")
    h = hashlib.sha256()  # solver_kernels.hip and the parts it includes (the same recipe as bench.kernel_source_sha256 / tools/profile_round.sh)
    for name in ["solver_kernels.hip"] + sorted(f for f in os.listdir(d) if f.startswith("solver_") and f.endswith(".inl")):
        with open(os.path.join(d, name), "rb") as f:
            h.update(f.read())
    sha = h.hexdigest()
    assert pj.get("kernel_source_sha256") == sha, "profiles/pmc_latest.json was collected on different kernels: re-run tools/profile_round.sh"
    assert pj.get("pass_b_hbm_bytes_per_launch", 0) > 0 and pj.get("pass_a_hbm_bytes_per_launch", 0) > 0
