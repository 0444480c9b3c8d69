"""Do the known-answer tests have teeth?  

The hot path of the solver has no reference-held vectors (reference test/solver_test.cpp:109-208 asserts nothing): the oracle's pin
for it is the closed-form suite (tests/closed_form.py) plus the reference's six value-pinning gtest cases (tests/test_oracle_pins.py,
test_ref_*).  This file builds the oracle with ONE deliberate deviation at a time (-DSO_MUTANT=k, oracle/sobfu_oracle.c) -- each a
plausible misreading of the reference -- and asserts that this suite, WITHOUT SURVEY Appendix B's recorded numbers, fails on every one
of them.  A mutant that survives is a missing closed-form case.  The same closed-form cases run on the HIP kernels through the C ABI
(tests/test_gpu_closed_form.py), so a HIP kernel with one of these deviations fails there too.

    k  deviation                                                          reference lines the correct behaviour follows
    1  the three 1-D passes COMPOSED instead of summed                    src/sobfu/cuda/solver.cu:155-160,290,366,443
    2  tap S[3+j] instead of S[3-j]                                       solver.cu:283-288
    3  lerp(v_lower, v_upper, t): operands swapped                        include/sobfu/cuda/utils.hpp:33-36
    4  upper index g+1 also at coordinate exactly 0                       utils.hpp:61-72
    5  psi += u                                                           solver.cu:66
    6  Laplacian sign                                                     src/sobfu/cuda/vector_fields.cu:291-337
    7  TSDF gradient clamps at a face instead of mirroring                vector_fields.cu:165-191
    8  sqrtf for __fsqrt_rd                                               utils.hpp:279-281
    9  (phi_global - phi_n o psi) instead of (phi_n o psi - phi_global)   solver.cu:28-31
   10  zero padding instead of clamp-to-edge in the convolutions          solver.cu:246-271
   11  warp weight from the upper corner instead of phi(floor(psi)).y     utils.hpp:78-85
   12  Laplacian mirrors the missing neighbour instead of using the centre vector_fields.cu:299-331
"""
import numpy as np
import pytest

import closed_form as cf
import test_oracle_pins as pins
from test_closed_form_oracle import OracleApi

MUTANTS = {1: "three passes composed", 2: "tap index S[3+j]", 3: "lerp operands swapped", 4: "upper index g+1 at coordinate 0",
           5: "psi += u", 6: "Laplacian sign", 7: "gradient clamps at a face", 8: "sqrtf for sqrt_rd", 9: "sign of (F - G)",
           10: "zero padding", 11: "weight from the upper corner", 12: "Laplacian mirrors at a face"}
GTESTS = [pins.test_ref_ClearTest_identity, pins.test_ref_TsdfGradientTest, pins.test_ref_UniformFieldJacobianTest,
          pins.test_ref_JacobianTestSimple, pins.test_ref_JacobianLaplacianTestComplicated, pins.test_ref_DataTermTest]


def kill_suite(O):
    """names of the checks that FAIL on the oracle build currently loaded (closed-form suite + the six reference gtest cases)"""
    failed = []
    try:
        cf.check_all(OracleApi(O), O.sobolev_filter(7, 0.1))
    except AssertionError as e:
        failed.append("closed_form: " + ((str(e).splitlines() or ["assert"])[0][:90]))
    for t in GTESTS:
        try:
            t(O)
        except AssertionError:
            failed.append(t.__name__)
    return failed


def test_unmutated_oracle_passes_the_kill_suite(oracle):
    assert kill_suite(oracle) == []


@pytest.mark.parametrize("k", sorted(MUTANTS))
def test_mutant_is_killed(oracle, k, tmp_path):
    path = oracle.build_mutant(k, str(tmp_path))
    with oracle.use_library(path):
        failed = kill_suite(oracle)
    assert failed, f"mutant {k} ({MUTANTS[k]}) SURVIVES the closed-form suite and the reference gtests: a known-answer case is missing"
    print(f"mutant {k:2d} ({MUTANTS[k]}): killed by {failed}")
