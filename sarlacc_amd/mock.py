"""Synthetic Nanopore-like reads following the reference's simulator
(/root/reference/R/mockReads.R:5-100): uniform bases, per-base substitution with a
uniform base (prob 0.05), per-base indel event (prob 0.01) replacing the base by k
copies of itself with k uniform in {0,2,..,max.insert}, made-up qualities with error
probability ~ U(0, sub.rate+indel.rate), half of the reads reverse-complemented.
numpy only; seeded; used by tests, smoke() and the CPU-baseline sample."""
import numpy as np

from .encoding import error_to_phred_char
from .strset import StringSet

NUC = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, np.uint8)
for a, b in zip(b"ACGTNMRWSYKVHDB-", b"TGCANKYWSRMBDHV-"):
    _COMP[a] = b


def revcomp(s):
    b = s.encode() if isinstance(s, str) else bytes(s)
    return _COMP[np.frombuffer(b, dtype=np.uint8)][::-1].tobytes().decode()


def mutate(ref, rng, sub_rate=0.05, indel_rate=0.01, max_insert=5):
    """One noisy copy of `ref` (uint8 array) (mockReads.R:69-80)."""
    r = ref.copy()
    sub = rng.random(r.size) < sub_rate
    r[sub] = NUC[rng.integers(0, 4, int(sub.sum()))]
    counts = np.ones(r.size, np.int64)
    ind = rng.random(r.size) < indel_rate
    choices = np.array([0] + list(range(2, max_insert + 1)))
    counts[ind] = choices[rng.integers(0, choices.size, int(ind.sum()))]
    return np.repeat(r, counts)


def mock_reads(adaptor1, adaptor2, nmolecules=10, nreads=10, seqlen=1000, seed=1000,
               sub_rate=0.05, indel_rate=0.01, max_insert=5, flip_strands=True):
    """Returns dict(reads=StringSet, quals=StringSet, molecule=int array, flipped=bool array,
    reference=list of str, umi=list of str).  adaptor1's first N-run is the barcode,
    the second the UMI (one run -> it is the barcode, as in mockReads.R:21-41; for the
    benchmark adaptors with a single N run we treat that run as the UMI)."""
    rng = np.random.default_rng(seed)
    a1 = np.frombuffer(adaptor1.encode(), dtype=np.uint8)
    rc2 = np.frombuffer(revcomp(adaptor2).encode(), dtype=np.uint8)
    npos = np.flatnonzero(a1 == ord("N"))
    runs = np.split(npos, np.flatnonzero(np.diff(npos) > 1) + 1) if npos.size else []
    seqs, quals, mol, flipped, refs, umis = [], [], [], [], [], []
    for m in range(nmolecules):
        body = NUC[rng.integers(0, 4, seqlen)]
        t1 = a1.copy()
        umi = b""
        for run in runs:
            fill = NUC[rng.integers(0, 4, run.size)]
            t1[run] = fill
            umi = fill.tobytes()
        ref = np.concatenate([t1, body, rc2])
        refs.append(ref.tobytes().decode())
        umis.append(umi.decode())
        for _ in range(nreads):
            r = mutate(ref, rng, sub_rate, indel_rate, max_insert)
            q = error_to_phred_char(rng.random(r.size) * (sub_rate + indel_rate))
            fl = bool(flip_strands and rng.random() < 0.5)
            if fl:
                r = _COMP[r][::-1]
                q = q[::-1]
            seqs.append(r.tobytes())
            quals.append(q.tobytes())
            mol.append(m)
            flipped.append(fl)
    return dict(reads=StringSet.from_strings(seqs), quals=StringSet.from_strings(quals),
                molecule=np.array(mol), flipped=np.array(flipped), reference=refs, umi=umis)


def random_reads(n, min_len, max_len, seed, alphabet=b"ACGT", qual_lo=33, qual_hi=126):
    """Plain random reads + random quality characters (for differential tests)."""
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(alphabet, dtype=np.uint8)
    seqs, quals = [], []
    for _ in range(n):
        L = int(rng.integers(min_len, max_len + 1))
        seqs.append(alpha[rng.integers(0, alpha.size, L)].tobytes().decode())
        quals.append(rng.integers(qual_lo, qual_hi + 1, L).astype(np.uint8).tobytes().decode())
    return seqs, quals
