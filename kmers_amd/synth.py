"""Seeded synthetic FASTA/FASTQ generators for the BASELINE configs.

Files end with a newline, FASTA is strictly one sequence per line with header `>` + decimal taxon
id, FASTQ is 4-line records with header `@r<index>` and quality all `I` -- the shapes the
reference's extractor (OtherFiles/KmersFromFile.java) handles without its wrapped-line quirks.
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def random_genomes(seed, n_taxa, length, first_taxon=1001, shared_frac=0.0):
    """-> (list of taxon ids, list of uint8 arrays).  shared_frac of each genome (after the first)
    is copied from its predecessor so that multi-taxon k-mers exist (cfg3)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    taxa, seqs = [], []
    for i in range(n_taxa):
        g = _ACGT[rng.integers(0, 4, size=length, dtype=np.uint8)]
        if shared_frac > 0 and i > 0:
            m = int(length * shared_frac)
            s = int(rng.integers(0, length - m + 1))
            o = int(rng.integers(0, length - m + 1))
            g[s:s + m] = seqs[i - 1][o:o + m]
        taxa.append(first_taxon + i)
        seqs.append(g)
    return taxa, seqs


def fasta_bytes(taxa, seqs):
    parts = []
    for t, s in zip(taxa, seqs):
        parts.append(b">%d\n" % t)
        parts.append(s.tobytes())
        parts.append(b"\n")
    return b"".join(parts)


def random_reads(seed, genomes, n_reads, read_len, sub_rate=0.01, n_rate=0.0):
    """Reads sampled from a uniformly random genome/offset/strand with substitution errors.
    -> uint8 array [n_reads, read_len]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    which = rng.integers(0, len(genomes), size=n_reads)
    strand = rng.integers(0, 2, size=n_reads)
    for gi, g in enumerate(genomes):
        idx = np.nonzero(which == gi)[0]
        if idx.size == 0:
            continue
        off = rng.integers(0, len(g) - read_len + 1, size=idx.size)
        win = g[off[:, None] + np.arange(read_len)[None, :]]
        rcm = strand[idx] == 1
        win[rcm] = _COMP[win[rcm][:, ::-1]]
        out[idx] = win
    if sub_rate > 0:
        m = rng.random(out.shape) < sub_rate
        out[m] = _ACGT[rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8)]
    if n_rate > 0:
        m = rng.random(out.shape) < n_rate
        out[m] = ord("N")
    return out


def fastq_bytes(reads, first_index=0):
    n, L = reads.shape
    qual = b"I" * L
    parts = []
    for i in range(n):
        parts.append(b"@r%d\n" % (first_index + i))
        parts.append(reads[i].tobytes())
        parts.append(b"\n+\n")
        parts.append(qual)
        parts.append(b"\n")
    return b"".join(parts)


def batch_from_seqs(seqs):
    """Concatenate sequences into the boundary's batch form: (bases uint8, seq_off uint64)."""
    lens = np.fromiter((len(s) for s in seqs), dtype=np.uint64, count=len(seqs))
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    bases = np.concatenate(seqs) if len(seqs) else np.zeros(0, np.uint8)
    return np.ascontiguousarray(bases, dtype=np.uint8), off


def cfg1(scale=1.0):
    """BASELINE config 1/2: 10 taxa x 500 kbp reference, 1M x 150 bp reads, 1 % substitutions."""
    glen = max(1000, int(500_000 * scale))
    nreads = max(100, int(1_000_000 * scale))
    taxa, genomes = random_genomes(1, 10, glen)
    reads = random_reads(2, genomes, nreads, 150, 0.01)
    return taxa, genomes, reads


def cfg5(scale=1.0, n_taxa=40):
    """BASELINE config 5 (long-read stress): genome sizes Zipf(s = 1.2) from 50 Mbp x scale down to 10 kbp, 5 % of the
    bases in homopolymer / dinucleotide tracts of 100-250 bases (a few key prefixes hold most of those k-mers), and
    10 kbp reads with 15 % substitutions and 0.5 % N (40 000 x scale of them, drawn from a uniformly random taxon: the
    small genomes are covered hundreds of times, the large ones hardly).  -> (taxa, genomes, reads[n, 10000])"""
    rng = np.random.Generator(np.random.PCG64(7))
    sizes = [max(10_000, int(50_000_000 * scale / (i + 1) ** 1.2)) for i in range(n_taxa)]
    genomes = []
    for n in sizes:
        g = _ACGT[rng.integers(0, 4, size=n, dtype=np.uint8)]
        n_tracts = max(1, int(n * 0.05 / 175))
        starts = rng.integers(0, max(1, n - 250), size=n_tracts)
        lens = rng.integers(100, 250, size=n_tracts)
        kinds = rng.integers(0, 4, size=(n_tracts, 3))
        for s, ln, (mono, a, b) in zip(starts.tolist(), lens.tolist(), kinds.tolist()):
            e = min(n, s + ln)
            if mono < 2:
                g[s:e] = _ACGT[a]
            else:
                g[s:e:2] = _ACGT[a]
                g[s + 1:e:2] = _ACGT[b]
        genomes.append(g)
    n_reads = max(20, int(40_000 * scale))
    reads = np.concatenate([random_reads(700 + c, genomes, min(2000, n_reads - c), 10_000, 0.15, n_rate=0.005)
                            for c in range(0, n_reads, 2000)])
    return list(range(9000, 9000 + n_taxa)), genomes, reads
