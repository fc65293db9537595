"""On-device synthetic read generator (torch is used here only as plumbing: device
memory + RNG).  Same recipe as sarlacc_amd.mock / the reference's mockReads
(/root/reference/R/mockReads.R:53-93), vectorised so that BASELINE-sized batches
(10^6 x 2 kb) are produced directly in HBM:
  read = adaptor1 (N runs filled with random bases) + uniform body + revcomp(adaptor2),
  5 % substitutions by a uniform base, 1 % indel events (k in {0,2..5} copies),
  error-probability qualities ~ U(0, 0.06) as Phred+33, 50 % strand flips.
"""
import numpy as np
import torch

from .mock import revcomp


def _phred_from_uniform(u, upper):
    # q = round(-10*log10(p)), p = u*upper ; capped to [0, 93]
    p = torch.clamp(u * upper, min=1e-30)
    q = torch.round(-10.0 * torch.log10(p)).clamp_(0, 93)
    return (q + 33).to(torch.uint8)


def make_reads(n, read_len, adaptor1, adaptor2, seed, device, sub_rate=0.05, indel_rate=0.01,
               max_insert=5, flip=True, chunk=65536):
    """Returns (seq uint8[total], qual uint8[total], off int64[n+1], max_len) on `device`."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    a1 = torch.tensor(list(adaptor1.encode()), dtype=torch.uint8, device=device)
    rc2 = torch.tensor(list(revcomp(adaptor2).encode()), dtype=torch.uint8, device=device)
    body_len = read_len - a1.numel() - rc2.numel()
    assert body_len > 0
    nuc = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for x, y in zip(b"ACGT", b"TGCA"):
        comp[x] = y
    choices = torch.tensor([0] + list(range(2, max_insert + 1)), dtype=torch.int64, device=device)
    is_n = (a1 == ord("N"))

    seqs, quals, lens = [], [], []
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        body = nuc[torch.randint(0, 4, (m, body_len), generator=g, device=device)]
        pre = a1.repeat(m, 1)
        fill = nuc[torch.randint(0, 4, (m, a1.numel()), generator=g, device=device)]
        pre = torch.where(is_n.unsqueeze(0), fill, pre)
        r = torch.cat([pre, body, rc2.repeat(m, 1)], dim=1)
        sub = torch.rand(r.shape, generator=g, device=device) < sub_rate
        r = torch.where(sub, nuc[torch.randint(0, 4, r.shape, generator=g, device=device)], r)
        if flip:
            fl = torch.rand(m, generator=g, device=device) < 0.5
            r = torch.where(fl.unsqueeze(1), comp[r.long()].flip(1), r)
        counts = torch.ones(r.shape, dtype=torch.int64, device=device)
        ind = torch.rand(r.shape, generator=g, device=device) < indel_rate
        k = choices[torch.randint(0, choices.numel(), r.shape, generator=g, device=device)]
        counts = torch.where(ind, k, counts)
        flat = torch.repeat_interleave(r.reshape(-1), counts.reshape(-1))
        q = _phred_from_uniform(torch.rand(flat.numel(), generator=g, device=device), sub_rate + indel_rate)
        seqs.append(flat)
        quals.append(q)
        lens.append(counts.sum(dim=1))
    lens = torch.cat(lens)
    off = torch.zeros(n + 1, dtype=torch.int64, device=device)
    off[1:] = torch.cumsum(lens, 0)
    seq = torch.cat(seqs)
    qual = torch.cat(quals)
    return seq, qual, off, int(lens.max().item())


def to_host_strings(seq, qual, off, count):
    """Pull the first `count` reads back as Python strings (CPU baseline / checks)."""
    o = off[: count + 1].cpu().numpy()
    end = int(o[-1])
    s = seq[:end].cpu().numpy().tobytes()
    q = qual[:end].cpu().numpy().tobytes()
    return ([s[o[i]:o[i + 1]].decode() for i in range(count)],
            [q[o[i]:o[i + 1]].decode() for i in range(count)])


def make_molecule_reads(molecules, copies, read_len, seed, device, umi_len=12, sub_rate=0.05, indel_rate=0.01,
                        max_insert=5, chunk=8192):
    """BASELINE configs C3 + C4 generated in HBM: `molecules` random bodies of `read_len` bases and
    random `umi_len`-base UMIs, each observed `copies` times through the mockReads error process
    (5 % substitutions by a uniform base, 1 % indel events with k in {0,2..5} copies, qualities from
    error probabilities ~ U(0, 0.06)); reads are in molecule order (read r comes from molecule r // copies).
    Returns dict(seq, qual, off, umi, umi_off, max_len) of device tensors (uint8 / int64)."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    nuc = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    choices = torch.tensor([0] + list(range(2, max_insert + 1)), dtype=torch.int64, device=device)

    def noisy(truth):
        r = truth.repeat_interleave(copies, dim=0)
        sub = torch.rand(r.shape, generator=g, device=device) < sub_rate
        r = torch.where(sub, nuc[torch.randint(0, 4, r.shape, generator=g, device=device)], r)
        counts = torch.ones(r.shape, dtype=torch.int64, device=device)
        ind = torch.rand(r.shape, generator=g, device=device) < indel_rate
        k = choices[torch.randint(0, choices.numel(), r.shape, generator=g, device=device)]
        counts = torch.where(ind, k, counts)
        flat = torch.repeat_interleave(r.reshape(-1), counts.reshape(-1))
        return flat, counts.sum(dim=1)

    seqs, quals, lens, umis, ulens = [], [], [], [], []
    for lo in range(0, molecules, chunk):
        m = min(chunk, molecules - lo)
        body = nuc[torch.randint(0, 4, (m, read_len), generator=g, device=device)]
        flat, ln = noisy(body)
        seqs.append(flat)
        quals.append(_phred_from_uniform(torch.rand(flat.numel(), generator=g, device=device), sub_rate + indel_rate))
        lens.append(ln)
        u = nuc[torch.randint(0, 4, (m, umi_len), generator=g, device=device)]
        uf, ul = noisy(u)
        umis.append(uf)
        ulens.append(ul)
    lens = torch.cat(lens)
    ulens = torch.cat(ulens)
    n = lens.numel()
    off = torch.zeros(n + 1, dtype=torch.int64, device=device)
    off[1:] = torch.cumsum(lens, 0)
    uoff = torch.zeros(n + 1, dtype=torch.int64, device=device)
    uoff[1:] = torch.cumsum(ulens, 0)
    return {"seq": torch.cat(seqs), "qual": torch.cat(quals), "off": off, "umi": torch.cat(umis), "umi_off": uoff,
            "max_len": int(lens.max().item())}
