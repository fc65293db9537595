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


def _trie_rank(umis):
    order = sorted(range(len(umis)), key=lambda i: ([{"A": 0, "C": 1, "G": 2, "T": 3, "N": 4}[c] for c in umis[i]], i))
    rank = [0] * len(umis)
    for r, i in enumerate(order):
        rank[i] = r
    return order, rank


def umi_pairs_shard(umi, limit, shard_index, shard_count):
    """CPU stand-in with the same contract: pairs (rank_i << 32 | rank_j) whose row rank falls in this
    shard's block of 256-wide row tiles (block boundaries need not match the GPU's)."""
    import numpy as np
    umis = _l(umi)
    order, rank = _trie_rank(umis)
    nbrs = O.fast_levdist_test(umis, limit)
    n = len(umis)
    nt = (n + 255) // 256
    lo = (nt * shard_index) // shard_count * 256
    hi = (nt * (shard_index + 1)) // shard_count * 256
    out = []
    for i, lst in enumerate(nbrs):
        ri = rank[i]
        if not (lo <= ri < hi):
            continue
        for j in (lst - 1).tolist():
            if rank[j] > ri:
                out.append((ri << 32) | rank[j])
    return np.array(out, dtype=np.uint64)


def umi_group_from_pairs(umi, limit, pairs):
    umis = _l(umi)
    order, rank = _trie_rank(umis)
    n = len(umis)
    lists = [[] for _ in range(n)]
    for p in sorted(int(x) for x in pairs):
        a, b = p >> 32, p & 0xffffffff
        lists[order[a]].append(b)
        lists[order[b]].append(a)
    links = []
    for i in range(n):
        own = [rank[i]] if umis[i].count("N") <= 2 * limit else []
        links.append([order[r] + 1 for r in sorted(set(lists[i] + own))])
    return O.cluster_umis_test(links)
