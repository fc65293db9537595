"""Writes tests/golden/reference_test_rule_cases.json: cases whose INPUTS are literals of the reference's
own tests and whose expected OUTPUTS follow from the independent rule the same test states next to them
(the test then asserts expect_identical(rule(inputs), .Call(...)), so the rule's value is what the
reference is held to).  Each rule is restated here in numpy from its description in the cited lines:

  basic consensus   tests/testthat/test-consensus.R:21-42 (column counts -> keep columns covered by
                    >= min.coverage of the rows -> first-maximum base among A,C,G,T -> error
                    1 - (n_chosen + pseudo/4) / (n_ACGT + pseudo), natural log)
  error -> Phred    tests/testthat/test-consensus.R:71-77 (round(-10 * lerr / ln 10), clamped to the
                    range of the Phred encoding, as characters)
  quality masking   tests/testthat/test-masking.R:5-14 (base -> 'N' where the error probability decoded
                    from its quality character exceeds the threshold); literal inputs :16-22, where
                    PhredQuality(numeric) of Biostrings (absent here; any version) encodes p as the
                    character 33 + round(-10 log10 p)

Run from the repository root:  python tests/golden/make_test_rule_cases.py
No part of the reference is imported or executed; /root/reference is not read.
"""
import json
import math
import os

import numpy as np

TEST_ALIGN = ["AAAAGAAAAA-AAATAAAA", "ACACA-AAAA--AAT-AGA", "GA-AG-C-A-T-AAT-AAA",
              "AT-AG-T-AGTAAGA-AGA", "-AAAGAT-AGTCAGA-AGA", "AGAAAAT-AGAAATA-AGA"]      # test-consensus.R:4-9
N_ALIGN = ["NAAAAANNN", "NNAANA---", "NNNANNN--", "NNNN--NN-"]                              # test-consensus.R:11-14
NUC = "ACGT"


def basic_rule(aln, min_cov, pseudo):
    rows = np.array([list(r) for r in aln])
    cons, lerr = [], []
    for col in rows.T:
        counts = [int((col == b).sum()) for b in NUC]
        if sum(counts) + int((col == "N").sum()) < min_cov * len(aln):
            continue
        k = int(np.argmax(counts))              # first maximum, A before C before G before T
        cons.append(NUC[k])
        lerr.append(math.log(1.0 - (counts[k] + pseudo / 4.0) / (sum(counts) + pseudo)))
    return "".join(cons), lerr


def phred_string(lerr):
    q = np.rint(np.asarray(lerr) / math.log(10.0) * -10.0)
    return "".join(chr(33 + int(v)) for v in np.clip(q, 0, 93))


def mask_rule(seq, qual, threshold):
    err = [10.0 ** (-(ord(c) - 33) / 10.0) for c in qual]
    return "".join("N" if e > threshold else b for b, e in zip(seq, err))


def phred_chars(probs):
    return "".join(chr(33 + int(round(-10.0 * math.log10(p)))) for p in probs)


def main():
    out = {"_provenance": __doc__.split("\n\n")[0].replace("\n", " "), "create_consensus_basic": [],
           "create_consensus_basic_loop": [], "mask_bad_bases": []}
    for name, aln, lines in (("test.align", TEST_ALIGN, "46-53"), ("n.align", N_ALIGN, "58-65")):
        for mc, pc in ((0.6, 1), (0.6, 2), (0.2, 2), (0.9, 1)):
            cons, lerr = basic_rule(aln, mc, pc)
            out["create_consensus_basic"].append({
                "source": "tests/testthat/test-consensus.R:%s (%s, min.coverage=%g, pseudo.count=%g; expected = BASICFUN :21-42)"
                          % (lines, name, mc, pc),
                "aln": aln, "min_cov": mc, "pseudo": pc, "consensus": cons, "lerr": lerr})
    loop = [N_ALIGN, TEST_ALIGN]
    res = [basic_rule(a, 0.6, 1) for a in loop]
    out["create_consensus_basic_loop"].append({
        "source": "tests/testthat/test-consensus.R:79-88 (list(n.align, test.align), 0.6, 1; expected = per-alignment results, "
                  "errors through errorToPhred :71-77)",
        "alns": loop, "min_cov": 0.6, "pseudo": 1, "consensus": [r[0] for r in res], "phred": [phred_string(r[1]) for r in res]})
    seqs = ["AAAATTTTCCCCGGGG", "GGGGTTTTCCCCAAAA", "AAAACCCCTTTTGGGG"]                       # test-masking.R:16-18
    probs = [np.repeat(10.0 ** np.array([-1., -2., -3., -4.]), 4),                             # :19-21
             np.repeat(10.0 ** np.array([-4., -3., -2., -1.]) / 2, 4),
             np.repeat(10.0 ** np.array([-3., -1., -2., -4.]) * 2, 4)]
    quals = [phred_chars(p) for p in probs]
    for thr in (0.001, 0.01, 0.05, 0.1):                                                      # :23-25
        out["mask_bad_bases"].append({
            "source": "tests/testthat/test-masking.R:16-25 (toy example, threshold %g; expected = CHECKFUN :5-14)" % thr,
            "seqs": seqs, "quals": quals, "threshold": thr, "masked": [mask_rule(s, q, thr) for s, q in zip(seqs, quals)]})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_test_rule_cases.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
