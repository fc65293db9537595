"""GPU: a fixed-seed slice of the differential fuzzer (tools/fuzz.py) -- every C-ABI entry point
against the oracle on randomised shapes, parameters and error cases.  The full fuzzer ran
72 168 cases over the final kernels without a mismatch (profiles/r01_fuzz_v3.txt, incl. the chunked host path; 59 633 on the
earlier ones, profiles/r01_fuzz.txt)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("block", range(4))
def test_fuzz_slice(block):
    from tools import fuzz
    for k in range(150):
        seed = 7_000_003 * (block + 1) + k
        fuzz.CASES[k % len(fuzz.CASES)](np.random.default_rng(seed))
    fuzz.case_umi_large(np.random.default_rng(11 * (block + 1)))
