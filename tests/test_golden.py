"""Golden vectors of the reference (tests/golden/reference_known_answers.json): the CPU oracle
must reproduce them (CPU run), and so must the HIP path through the C ABI (GPU run)."""
import json
import os

import numpy as np
import pytest

_HERE = os.path.join(os.path.dirname(__file__), "golden")
_KNOWN = json.load(open(os.path.join(_HERE, "reference_known_answers.json")))
RULE = json.load(open(os.path.join(_HERE, "reference_test_rule_cases.json")))
SECTIONS = ("reference_held", "survey_recorded", "constructed")


def _cases(group):
    """every case of a group, whatever its evidence section"""
    return [c for sec in SECTIONS for c in _KNOWN[sec].get(group, [])]


class _Gold:
    def __getitem__(self, group):
        return _cases(group)


GOLD = _Gold()


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
    for case in GOLD["create_consensus_quality"]:
        out = impl.create_consensus_quality(case["aln"], case["min_cov"], case["quals"], enc)
        assert out[0] == case["consensus"] and np.asarray(out[1]).tolist() == case["lerr"], case["source"]
    # expected values from the rule the reference's test states next to its literal inputs
    for case in RULE["create_consensus_basic"]:
        out = impl.create_consensus_basic(case["aln"], case["min_cov"], case["pseudo"])
        assert out[0] == case["consensus"], case["source"]
        # natural-log errors: identical up to the last bits of log / log1p (the reference test compares R's
        # log(error) with the C++ log1p(-p) through expect_identical on these very inputs)
        assert np.allclose(np.asarray(out[1]), case["lerr"], rtol=1e-13, atol=0), case["source"]
    for case in RULE["create_consensus_basic_loop"]:
        out = impl.create_consensus_basic_loop(case["alns"], case["min_cov"], case["pseudo"])
        assert list(out[0]) == case["consensus"] and list(out[1]) == case["phred"], case["source"]
    for case in RULE["mask_bad_bases"]:
        assert list(impl.mask_bad_bases(case["seqs"], case["quals"], enc, case["threshold"])) == case["masked"], case["source"]


def test_oracle_reproduces_golden_vectors(oracle, oenc):
    _check(oracle, oenc, oracle.OracleError)


@pytest.mark.gpu
def test_hip_path_reproduces_golden_vectors(enc):
    import sarlacc_amd
    from sarlacc_amd import calls
    _check(calls, enc, sarlacc_amd.SarlaccError)


def test_golden_sections_are_labelled_honestly():
    """reference_held cites only the reference's tests / Rd pages; survey_recorded says so; nothing is in two sections."""
    for group, cases in _KNOWN["reference_held"].items():
        for c in cases:
            assert ("tests/testthat/" in c["source"] or "man/" in c["source"]) and "SURVEY" not in c["source"] \
                and "recorded" not in c["source"], (group, c["source"])
    for group, cases in _KNOWN["survey_recorded"].items():
        for c in cases:
            assert "SURVEY" in c["source"], (group, c["source"])
    for group, cases in _KNOWN["constructed"].items():
        for c in cases:
            assert "NOT a pin" in c["source"], (group, c["source"])
    assert set(_KNOWN) == set(SECTIONS) | {"_provenance"}


def test_host_logic_literals_of_the_reference_tests():
    """.setup_subseqs (tests/testthat/test-adaptor-align.R:125-127) and .tied_overlap (test-tuning.R:54-58)."""
    from sarlacc_amd import generics
    for case in GOLD["setup_subseqs"]:
        out = generics._setup_subseqs(case["adaptor"])
        assert out["starts"].tolist() == case["starts"] and out["ends"].tolist() == case["ends"], case["source"]
    for case in GOLD["tied_overlap"]:
        got = generics._tied_overlap(np.asarray(case["real"], float), np.asarray(case["fake"], float))
        assert abs(got - case["value"]) < 1e-12, case["source"]


def test_rule_cases_file_is_what_its_script_writes(tmp_path):
    """the committed fixture is reproducible from the committed script"""
    import subprocess
    import sys
    src = os.path.join(_HERE, "make_test_rule_cases.py")
    code = open(src).read().replace('os.path.dirname(os.path.abspath(__file__))', repr(str(tmp_path)))
    subprocess.run([sys.executable, "-c", code], check=True, stdout=subprocess.DEVNULL)
    assert json.load(open(tmp_path / "reference_test_rule_cases.json")) == RULE
