"""File-level drivers mirroring the three reference programs on the hot path (their CLIs stay in Java in a
real deployment; these exist so the path can be run end to end, and cross-checked against Kmers.jar,
without a JVM):
  make_database  <- Database/MakeDatabase.java  (-a | -q, -i -o -K -k -l -z|-Z -r -D -R)
  matcher        <- Database/Matcher.java       (-i -d -o -z|-Z)
  taxa_counts    <- Utils/TaxaCounts.java       (-d -o)
"""
import gzip
import struct

import numpy as np

from . import container, parse
from .api import DB, KmerDatabase, KmersGpuError, READ_ID, REF_ID, match_lines, merge_databases, taxa_count_lines
from ._lib import KMG_EUNSUPPORTED, PAYLOAD_READPOS, PAYLOAD_TAXON
from .make_database import make_read_db, make_reference_db


def write_database(db: KmerDatabase, path, zip_level=5):
    """FileCreator.create (FileCreator.java:117-217): meta under key -1, then every key 0..4^l-1 in order
    (each gets a block, empty ones included, quirk C.12)."""
    blocks, off = db.encode_blocks()
    out = container.IndexedOutputFile(path, hr=False, zip_level=zip_level)
    out.write_all(db.meta(), -1)
    for key in range(4 ** db.key_length):
        out.set_current_key(key)
        out.write(blocks[int(off[key]):int(off[key + 1])])
    out.close()


def read_meta(f: container.IndexedInputFile):
    """KmerFile.getMetaData, binary branch (KmerFile.java:284-288)."""
    m = f.read_key(-1)
    min_k, max_k, key_len = m[0], m[1], m[2]
    (n,) = struct.unpack_from(">i", m, 3)
    ids = list(struct.unpack_from(">%di" % n, m, 7))
    return min_k, max_k, key_len, ids, m[7 + 4 * n] == 1


def open_database(ctx, path) -> KmerDatabase:
    """new KmerFile(file, dataType) + its decoded content uploaded to the GPU."""
    f = container.IndexedInputFile(path)
    if f.hr:
        raise KmersGpuError(KMG_EUNSUPPORTED, "human-readable (-h) databases are read by the Java path")
    min_k, max_k, key_len, ids, rc = read_meta(f)
    if min_k != max_k or max_k > 32:
        raise KmersGpuError(KMG_EUNSUPPORTED, "minK != maxK or k > 32: use the Java path")
    if ids == REF_ID:
        kind = PAYLOAD_TAXON
    elif ids == READ_ID:
        kind = PAYLOAD_READPOS
    else:
        raise ValueError("UnexpectedDataTypeException: %s" % path)       # KmerFile.java:30-33
    keys, val_off, val_a, val_b = parse.decode_blocks(f.read_keys(0, f.max_index()), kind)
    return KmerDatabase.load(ctx, max_k, key_len, rc, kind, keys, val_off, val_a, val_b)


def make_database(ctx, input_path, output_path, mode, k=32, min_k=24, key_length=6, zip_level=5, read_map_path=None,
                  dust=None, runofsame=None):
    """MakeDatabase -a (mode 'a') / -q (mode 'q').  Defaults as the reference's (-K 32 -k 24 -l 6 -z 5,
    MakeDatabase.java:106-109); minK != maxK is refused (GPU path limit)."""
    if min_k != k:
        raise KmersGpuError(KMG_EUNSUPPORTED, "minK != maxK is not supported on the GPU path")
    data = open(input_path, "rb").read()
    if mode == "a":
        db = make_reference_db(ctx, data, k, key_length, dust, runofsame)
    elif mode == "q":
        if read_map_path is None:      # new File(null) -> NullPointerException in the reference (:172-173)
            raise ValueError("-r <read map file> is required with -q")
        db, names = make_read_db(ctx, data, k, key_length, dust, runofsame)
        with gzip.open(read_map_path, "wt") as g:
            for nm in names:
                g.write(nm + "\n")
    else:
        raise ValueError("input mode must be 'a' (FASTA reference) or 'q' (FASTQ reads)")
    write_database(db, output_path, zip_level)
    return db


def matcher(ctx, search_path, db_path, output_path, plain=False):
    """Matcher with -n 0: one line per matched search k-mer, ascending (Matcher.java:235-250); output is
    gzip at the default level unless plain (-Z x)."""
    dbfiles = container.file_set(db_path)
    sfiles = container.file_set(search_path)
    if not dbfiles or not sfiles:
        raise FileNotFoundError(db_path if not dbfiles else search_path)
    db = DB([open_database(ctx, f) for f in dbfiles])          # name.1 .. name.N are merged (DB.java:99-109)
    searches = [open_database(ctx, f) for f in sfiles]
    search = searches[0] if len(searches) == 1 else merge_databases(searches)   # Matcher.java:111-128 via KmerUtils
    hit, _ = db.get_nearest_kmers(search)
    lines = match_lines(search, db.merged, hit)
    text = "".join(l + "\n" for l in lines).encode()
    with (open(output_path, "wb") if plain else gzip.open(output_path, "wb")) as f:
        f.write(text)
    return lines


def taxa_counts(ctx, db_path, output_path):
    """TaxaCounts default mode: `taxon \\t match \\t unique`, ascending taxon, plain text (TaxaCounts.java:102-150)."""
    dbfiles = container.file_set(db_path)
    if not dbfiles:
        raise FileNotFoundError(db_path)
    lines = taxa_count_lines(DB([open_database(ctx, f) for f in dbfiles]))
    with open(output_path, "w") as f:
        f.write("".join(l + "\n" for l in lines))
    return lines
