"""Adapter exposing the CPU oracle under the same interface as sarlacc_amd.calls, so
that the generic-level functions (sarlacc_amd.generics) can be run end-to-end against
the oracle in tests.  Test infrastructure only."""
from oracle import oracle as O
from sarlacc_amd.strset import StringSet


def _l(x):
    return x.to_strings() if isinstance(x, StringSet) else list(x)


def _e(enc):
    return (enc.errors, enc.names) if hasattr(enc, "errors") else enc


def adaptor_align(seq, qual, enc, go, ge, adaptor, ss, se):
    return list(O.adaptor_align(_l(seq), _l(qual), _e(enc), go, ge, adaptor, ss, se))


def barcode_align(seq, qual, enc, go, ge, ref):
    return O.barcode_align(_l(seq), _l(qual), _e(enc), go, ge, ref)


def general_align(seq, qual, enc, go, ge, ref, edit_only):
    return list(O.general_align(_l(seq), _l(qual), _e(enc), go, ge, ref, edit_only))


def mask_bad_bases(seq, qual, enc, thr):
    return O.mask_bad_bases(_l(seq), _l(qual), _e(enc), thr)


def compute_lev_masked(seqs):
    return O.compute_lev_masked(_l(seqs))


def umi_group(u1, t1, u2, t2, groups):
    return O.umi_group(_l(u1), t1, None if u2 is None else _l(u2), t2, groups, fast=True)


def quick_msa(groups, seqs, ma, mm, gx, go, bw):
    return O.quick_msa(groups, _l(seqs), ma, mm, gx, go, bw)


def create_consensus_basic_loop(aln, cov, pc):
    return list(O.create_consensus_basic_loop(aln, cov, pc))


def create_consensus_quality_loop(aln, cov, quals, enc):
    return list(O.create_consensus_quality_loop(aln, cov, quals, _e(enc)))
