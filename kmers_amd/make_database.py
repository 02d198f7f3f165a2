"""Driver wiring of Database/MakeDatabase.java (:127-185, :279-318) over the GPU FileCreator:
-a (FASTA -> reference DB, rc=True) and -q (FASTQ -> read DB + read-name map, rc=False)."""
import numpy as np

from . import parse
from ._lib import KMG_EUNSUPPORTED, KmersGpuError
from .api import FileCreator


def _check_eof(pf, mode, k, min_k):
    """A file that ends inside a sequence line: the reference hands its last k-mer out twice (KmersFromFile.java:115-123,
    193-200).  For FASTA with k-mers of one length the record splitter reproduces that; a FASTQ cut there, or any such
    file with minK < maxK (where suffixes follow the repeated k-mer), is refused rather than built differently."""
    if pf.high_bytes:
        # Sequence.java:15-33 decides by the sign of a window's first byte: windows of bytes >= 128 only are k-mers of
        # garbage codes there, invalid here -- refused rather than built differently
        raise KmersGpuError(KMG_EUNSUPPORTED, "sequence lines hold bytes >= 128: not supported on the GPU path")
    if mode == parse.FASTQ and pf.empty_header:
        # ReadIDMapping starts with curReadS = "" (ReadIDMapping.java:10-16): a first emitting read with an empty header
        # gets index -1 and no name line there
        raise KmersGpuError(KMG_EUNSUPPORTED, "a FASTQ record with an empty header: not supported on the GPU path")
    if pf.eof_in_kmer and (mode == parse.FASTQ or min_k != k):
        raise KmersGpuError(KMG_EUNSUPPORTED, "input ends inside a sequence line (no final newline): not supported "
                            "on the GPU path for FASTQ input or with minK < maxK")


def make_reference_db(ctx, fasta, k=32, key_length=6, dust=None, runofsame=None, key_range=None, min_k=None, id_map=None):
    """MakeDatabase -a -K k -k min_k -l key_length [-D dust] [-R runofsame] [-m id_map] (min_k: default k).  -> KmerDatabase
    fasta: the file's bytes, or what parse.parse made of them (a build in key ranges parses once).
    id_map: the bytes of the -m file (sequence id <tab> taxon id); headers are then names, not ints."""
    min_k = k if min_k is None else min_k
    pf = fasta if isinstance(fasta, parse.ParsedFile) else parse.parse(fasta, parse.FASTA, k, id_map=id_map)
    _check_eof(pf, parse.FASTA, k, min_k)
    fc = FileCreator(ctx, key_length, k, "ref", rc=True, dust=dust, runofsame=runofsame, key_range=key_range)
    any_bad = bool((pf.id_ok == 0).any())
    has = fc.add_kmers(pf.batch, min_length=min_k, max_length=k, flags=any_bad)
    if any_bad:
        # Integer.parseInt throws NumberFormatException (with -m: map.get is null) at the sequence's first k-mer and not
        # before -- new Kmer(..) is evaluated first, KmersFromFile.java:103,165 -- so a header that is no int only fails
        # for the sequences the build flags as having produced a k-mer
        bad = np.nonzero((pf.id_ok == 0) & (has >= 0))[0]
        if bad.size:
            fc.create().close()
            raise ValueError("FASTA header of sequence %d is not %s" % (int(bad[0]), "in the id map" if pf.named else "an int"))
    db = fc.create()
    db.n_windows = fc.n_windows
    return db


def make_read_db(ctx, fastq, k=32, key_length=6, dust=None, runofsame=None, key_range=None, min_k=None):
    """MakeDatabase -q -r map [-D dust] [-R runofsame].  -> (KmerDatabase, read-name map lines in index order)"""
    min_k = k if min_k is None else min_k
    pf = fastq if isinstance(fastq, parse.ParsedFile) else parse.parse(fastq, parse.FASTQ, k)
    _check_eof(pf, parse.FASTQ, k, min_k)
    fc = FileCreator(ctx, key_length, k, "read", rc=False, dust=dust, runofsame=runofsame, key_range=key_range)
    ridx = fc.add_kmers(pf.batch, min_length=min_k, max_length=k)
    db = fc.create()
    db.n_windows = fc.n_windows
    # ReadIDMapping.geNext prints a header each time a new index is handed out
    names = []
    tok = pf.batch.seq_tok
    last = -1
    for i in np.nonzero(ridx >= 0)[0].tolist():
        if ridx[i] != last:
            names.append(pf.names[int(tok[i])])
            last = int(ridx[i])
    return db, names
