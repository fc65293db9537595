"""Throughput of umiGroup -> multiReadAlign -> consensusReadSeq at BASELINE scale
(config 3 + config 4): G molecules x 10 reads x 2 kb, 12-bp UMIs, one pre-group.
Host-pointer C ABI (PCIe copies included), numpy-only marshalling."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sarlacc_amd
from sarlacc_amd import calls
from sarlacc_amd.strset import StringSet

NUC = np.frombuffer(b"ACGT", dtype=np.uint8)


def noisy_copies(truth, copies, rng, sub=0.05, indel=0.01, chunk=5000):
    """truth: (G, L) uint8 -> StringSet of G*copies noisy reads (mockReads error process), + quals.
    Generated `chunk` molecules at a time so that 10^6 x 2 kb stays within a few GB of host memory."""
    if truth.shape[0] > chunk:
        parts = [_noisy_copies(truth[i:i + chunk], copies, rng, sub, indel) for i in range(0, truth.shape[0], chunk)]
        chars = np.concatenate([p[0].chars[:p[0].total] for p in parts])
        qchars = np.concatenate([p[1].chars[:p[1].total] for p in parts])
        lens = np.concatenate([np.diff(p[0].off) for p in parts])
        off = np.zeros(lens.size + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        return StringSet(chars, off), StringSet(qchars, off.copy())
    return _noisy_copies(truth, copies, rng, sub, indel)


def _noisy_copies(truth, copies, rng, sub, indel):
    r = np.repeat(truth, copies, axis=0)
    s = rng.random(r.shape, dtype=np.float32) < sub
    r[s] = NUC[rng.integers(0, 4, int(s.sum()))]
    counts = np.ones(r.shape, np.int8)
    ind = rng.random(r.shape, dtype=np.float32) < indel
    ch = np.array([0, 2, 3, 4, 5])
    counts[ind] = ch[rng.integers(0, 5, int(ind.sum()))]
    flat = np.repeat(r.reshape(-1), counts.reshape(-1))
    lens = counts.sum(1)
    off = np.zeros(len(lens) + 1, np.int64)
    off[1:] = np.cumsum(lens)
    p = rng.random(flat.size, dtype=np.float32) * np.float32(0.06)
    q = (np.clip(np.round(-10 * np.log10(np.maximum(p, 1e-30))), 0, 93) + 33).astype(np.uint8)
    return StringSet(flat, off), StringSet(q, off.copy())


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    rng = np.random.default_rng(1000)
    t0 = time.perf_counter()
    umi_truth = NUC[rng.integers(0, 4, (G, 12))]
    body_truth = NUC[rng.integers(0, 4, (G, L))]
    umis, _ = noisy_copies(umi_truth, 10, rng)
    reads, quals = noisy_copies(body_truth, 10, rng)
    n = len(reads)
    print("generated %d reads (%.1f s)" % (n, time.perf_counter() - t0), flush=True)
    enc = sarlacc_amd.phred_encoding()

    for rep in range(2):
        t0 = time.perf_counter()
        coff, cmem = calls.umi_group_flat(umis, 1, None, 1, np.array([0, n], np.int64), np.arange(1, n + 1, dtype=np.int32))
        t1 = time.perf_counter()
        goff, gflat = calls.csr_select(coff, cmem, np.diff(coff) >= 2)   # clusters of >= 2 reads
        t2 = time.perf_counter()
        cons, phred = calls.msa_consensus_flat(goff, gflat, reads, 0, -1, -5, -1, 100, 0.6, quals=quals, encoding=enc)
        t5 = time.perf_counter()
        big = [gflat[goff[k]:goff[k + 1]] for k in range(len(goff) - 1)]
        print("rep %d: umi_group %.2fs | numpy glue %.2fs | msa+consensus (fused, rows stay in HBM) %.2fs | "
              "%d clusters>=2 covering %d reads | %.2f M reads/min end to end, consensus mean len %.0f"
              % (rep, t1 - t0, t2 - t1, t5 - t2, len(big), gflat.size,
                 n / (t5 - t0) * 60 / 1e6, np.mean(np.diff(cons.off))), flush=True)
    # accuracy of the consensus vs truth for pure clusters
    from tests.test_oracle_umi import lev2
    mol = np.repeat(np.arange(G), 10)
    errs = []
    cs = cons.to_strings()
    for k in range(0, len(big), max(1, len(big) // 200)):
        m = mol[big[k] - 1]
        if len(set(m.tolist())) == 1 and len(m) >= 8:
            t = body_truth[m[0]].tobytes().decode()
            errs.append(lev2(cs[k], t) / 2 / len(t))
    if errs:
        print("consensus error rate vs truth (pure clusters >= 8 reads, n=%d): mean %.5f max %.5f" % (len(errs), np.mean(errs), np.max(errs)))


if __name__ == "__main__":
    main()
