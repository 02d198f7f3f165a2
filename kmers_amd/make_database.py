"""Driver wiring of Database/MakeDatabase.java (:127-185, :279-318) over the GPU FileCreator:
-a (FASTA -> reference DB, rc=True) and -q (FASTQ -> read DB + read-name map, rc=False)."""
import numpy as np

from . import parse
from .api import FileCreator


def make_reference_db(ctx, fasta, k=32, key_length=6, dust=None, runofsame=None, key_range=None):
    """MakeDatabase -a -K k -k k -l key_length [-D dust] [-R runofsame].  -> KmerDatabase
    fasta: the file's bytes, or what parse.parse made of them (a build in key ranges parses once)."""
    pf = fasta if isinstance(fasta, parse.ParsedFile) else parse.parse(fasta, parse.FASTA, k)
    lens = np.diff(pf.batch.seq_off.astype(np.int64))
    bad = np.nonzero((pf.id_ok == 0) & (lens >= k))[0]
    if bad.size:   # Integer.parseInt throws NumberFormatException at the sequence's first k-mer
        raise ValueError("NumberFormatException: FASTA header of sequence %d is not an int" % int(bad[0]))
    fc = FileCreator(ctx, key_length, k, "ref", rc=True, dust=dust, runofsame=runofsame, key_range=key_range)
    fc.add_kmers(pf.batch)
    db = fc.create()
    db.n_windows = fc.n_windows
    return db


def make_read_db(ctx, fastq, k=32, key_length=6, dust=None, runofsame=None, key_range=None):
    """MakeDatabase -q -r map [-D dust] [-R runofsame].  -> (KmerDatabase, read-name map lines in index order)"""
    pf = fastq if isinstance(fastq, parse.ParsedFile) else parse.parse(fastq, parse.FASTQ, k)
    fc = FileCreator(ctx, key_length, k, "read", rc=False, dust=dust, runofsame=runofsame, key_range=key_range)
    ridx = fc.add_kmers(pf.batch)
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
