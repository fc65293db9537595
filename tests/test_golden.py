"""Golden vectors of the reference (tests/golden/reference_known_answers.json): the CPU oracle
must reproduce them (CPU run), and so must the HIP path through the C ABI (GPU run)."""
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


def _check(impl, enc, err_type):
    for case in GOLD["adaptor_align"]:
        reads = case["reads"]
        quals = [case["qual_char"] * len(r) for r in reads]
        out = impl.adaptor_align(reads, quals, enc, case["gapopen"], case["gapext"], case["adaptor"],
                                 case.get("sec_starts", []), case.get("sec_ends", []))
        if "scores" in case:
            assert np.asarray(out[0]).tolist() == case["scores"], case["source"]
        assert np.asarray(out[1]).tolist() == case["starts"] if "starts" in case else True, case["source"]
        assert np.asarray(out[2]).tolist() == case["ends"] if "ends" in case else True, case["source"]
        if "sec_start" in case:
            assert [np.asarray(x).tolist() for x in out[3]] == case["sec_start"], case["source"]
            assert [np.asarray(x).tolist() for x in out[4]] == case["sec_width"], case["source"]
    for case in GOLD["compute_lev_masked"]:
        assert np.asarray(impl.compute_lev_masked(case["seqs"])).tolist() == case["dist"], case["source"]
    for case in GOLD["fast_levdist_test"]:
        got = impl.fast_levdist_test(case["seqs"], case["limit"])
        assert [sorted(np.asarray(x).tolist()) for x in got] == case["sorted_neighbours"], case["source"]
    for case in GOLD["umi_group"]:
        got = impl.umi_group(case["umi1"], case["threshold1"], case.get("umi2"), case.get("threshold2", case["threshold1"]),
                             case["groups"])
        assert [np.asarray(c).tolist() for c in got] == case["clusters"], case["source"]
    for case in GOLD["create_consensus_basic"]:
        assert impl.create_consensus_basic(case["aln"], case["min_cov"], case["pseudo"])[0] == case["consensus"], case["source"]
    for case in GOLD["unmask_alignment"]:
        assert impl.unmask_alignment(case["alignments"], case["originals"]) == case["unmasked"], case["source"]
    for case in GOLD["unmask_errors"]:
        with pytest.raises(err_type, match=case["message"]):
            impl.unmask_alignment(case["alignments"], case["originals"])
    for case in GOLD["errors"]:
        with pytest.raises(err_type, match=case["message"]):
            impl.create_consensus_quality(case["aln"], 0.6, case["quals"], enc)


def test_oracle_reproduces_golden_vectors(oracle, oenc):
    _check(oracle, oenc, oracle.OracleError)


@pytest.mark.gpu
def test_hip_path_reproduces_golden_vectors(enc):
    import sarlacc_amd
    from sarlacc_amd import calls
    _check(calls, enc, sarlacc_amd.SarlaccError)
