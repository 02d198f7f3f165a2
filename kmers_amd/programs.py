"""File-level drivers mirroring the three reference programs on the hot path (their CLIs stay in Java in a
real deployment; these exist so the path can be run end to end, and cross-checked against Kmers.jar,
without a JVM):
  make_database  <- Database/MakeDatabase.java  (-a | -q, -i -o -K -k -l -z|-Z -r -D -R)
  matcher        <- Database/Matcher.java       (-i -d -o -z|-Z)
  taxa_counts    <- Utils/TaxaCounts.java       (-d -o)
  collect_by_read <- Utils/CollectByRead.java   (-i -o)      read_classifier <- Utils/ReadClassifier.java (-i -o)
  read_calls     = MakeDatabase -q + Matcher + CollectByRead + ReadClassifier in one go, no intermediate files
"""
import gzip
import struct

import numpy as np

from . import container, parse
from .api import (DB, KmerDatabase, KmersGpuError, READ_ID, REF_ID, classify_reads, collect_by_read_blocks, kmer_strings,
                  match_lines, merge_databases, read_call_blocks, taxa_count_lines)
from ._lib import KMG_EINVAL, KMG_EUNSUPPORTED, InconsistentDataException, PAYLOAD_READPOS, PAYLOAD_TAXON
from .make_database import make_read_db, make_reference_db


def _hr_values(kind, a, b):
    """CountDataType.toString `TxC|TxC` (unsaturated counts, CountDataType.java:109-113) | SetDataType<ReadPos>
    with its default separator `r:p r:p` (DataCollector.java:67, ReadPosDataType.java:46-49)."""
    if kind == PAYLOAD_TAXON:
        return "|".join("%dx%d" % x for x in zip(a, b))
    return " ".join("%d:%d" % x for x in zip(a, b))


def write_database(db: KmerDatabase, path, zip_level=5, hr=False, out=None, key_range=None, with_meta=None):
    """FileCreator.create (FileCreator.java:117-217): meta under key -1, then every key 0..4^l-1 in order
    (each gets a block, empty ones included, quirk C.12).  hr = MakeDatabase -h: text meta, one text line per
    k-mer with the characters it shares with the previous k-mer of its key block dropped
    (FileCreator.java:166-179, 234-242, 349-370), text index.
    out / key_range: append the blocks of keys [first, end) to an open file instead (a database built in key
    ranges, one call per range in ascending order; the first call writes the meta block, the caller closes)."""
    if hr and db.min_length != db.max_length:
        # (FileCreator.diff, FileCreator.java:234-242, runs off the end of the shorter of two k-mers where one is a
        # prefix of the other: the reference itself throws on such a database)
        raise KmersGpuError(KMG_EUNSUPPORTED, "human-readable databases (-h) hold k-mers of one length on the GPU path")
    part = out is not None
    first, end = key_range if key_range is not None else (0, 4 ** db.key_length)
    if with_meta is None:
        with_meta = not part or first == 0
    if out is None:
        out = container.IndexedOutputFile(path, hr=hr, zip_level=zip_level)
    if not hr:
        blocks, off = db.encode_blocks()
        if with_meta:
            out.write_all(db.meta(), -1)
        for key in range(first, end):
            out.set_current_key(key)
            out.write(blocks[int(off[key]):int(off[key + 1])])
        if not part:
            out.close()
        return
    ids = " ".join(str(i) for i in db.data_id())                     # Compressor.getStringID
    if with_meta:
        out.write_all(("%d\n%d\n%d\n%s\n%s\n" % (db.k, db.k, db.key_length, ids, "1" if db.rc else "0")).encode(), -1)
    h = db.fetch()
    names = kmer_strings(h["keys"], db.k, h.get("keys_lo"))
    val_off, va, vb = h["val_off"].tolist(), h["val_a"].tolist(), h["val_b"].tolist()
    boff = h["bucket_off"].tolist()
    for key in range(first, end):
        out.set_current_key(key)
        last = " " * db.k
        lines = []
        for u in range(boff[key], boff[key + 1]):
            nxt = names[u]
            first = 0
            while last[first] == nxt[first]:
                first += 1
            lines.append("%s\t%s\n" % (nxt[first:], _hr_values(db.kind, va[val_off[u]:val_off[u + 1]], vb[val_off[u]:val_off[u + 1]])))
            last = nxt
        out.write("".join(lines).encode())
    if not part:
        out.close()


def _read_hr_database(ctx, f: container.IndexedInputFile, first=0, last=None):
    """KmerFile over a human-readable file: text meta (KmerFile.java:279-283) and UncompressedKmerSpliterator
    (KmerFile.java:132-160): k-mer = previous k-mer with its tail replaced by the line's first field
    (Kmer.java:26-44), values parsed by CountDataType.fromString / SetDataType<ReadPos>.fromString."""
    m = f.read_key(-1).decode().split("\n")
    min_k, max_k, key_len = int(m[0]), int(m[1]), int(m[2])
    ids = [int(x) for x in m[3].split()]
    rc = m[4] == "1"
    if min_k != max_k or max_k > 64:
        raise KmersGpuError(KMG_EUNSUPPORTED, "minK != maxK or k > 64: use the Java path")
    if ids == REF_ID:
        kind = PAYLOAD_TAXON
    elif ids == READ_ID:
        kind = PAYLOAD_READPOS
    else:
        raise ValueError("UnexpectedDataTypeException")
    code = {"A": 0, "C": 1, "T": 2, "G": 3}
    keys, keys_lo, val_off, va, vb = [], [], [0], [], []
    prev = None
    for line in f.read_keys(first, f.max_index() if last is None else last).decode().splitlines():
        part, vals = line.split("\t")
        cur = part if prev is None or len(part) >= len(prev) else prev[:len(prev) - len(part)] + part
        prev = cur
        v = 0
        for ch in cur:
            v = (v << 2) | code[ch.upper()]
        v <<= 128 - 2 * len(cur)                 # (two words: bases 0 .. 31, bases 32 ..)
        keys.append(v >> 64)
        keys_lo.append(v & (2 ** 64 - 1))
        if kind == PAYLOAD_TAXON:
            for item in vals.split("|"):
                t, c = item.split("x")
                va.append(int(t)); vb.append(min(int(c), 2 ** 31 - 1))
        else:
            for item in vals.split(" "):
                r, p = item.split(":")
                va.append(int(r)); vb.append(int(p))
        val_off.append(len(va))
    return KmerDatabase.load(ctx, max_k, key_len, rc, kind, np.array(keys, np.uint64), np.array(val_off, np.uint64),
                             np.array(va, np.int32), np.array(vb, np.int32), keys_lo=np.array(keys_lo, np.uint64))


def read_meta(f: container.IndexedInputFile):
    """KmerFile.getMetaData, binary branch (KmerFile.java:284-288)."""
    m = f.read_key(-1)
    min_k, max_k, key_len = m[0], m[1], m[2]
    (n,) = struct.unpack_from(">i", m, 3)
    ids = list(struct.unpack_from(">%di" % n, m, 7))
    return min_k, max_k, key_len, ids, m[7 + 4 * n] == 1


def open_database(ctx, path, key_range=None) -> KmerDatabase:
    """new KmerFile(file, dataType) + its decoded content uploaded to the GPU.  key_range = (first, last), both
    inclusive: only those keys' blocks (KmerFile.kmers(minKey, maxKey); Matcher -L)."""
    f = container.IndexedInputFile(path)
    first, last = (0, f.max_index()) if key_range is None else key_range
    if f.hr:
        return _read_hr_database(ctx, f, first, last)
    min_k, max_k, key_len, ids, rc = read_meta(f)
    if max_k > 64:
        raise KmersGpuError(KMG_EUNSUPPORTED, "k > 64: use the Java path")
    if ids == REF_ID:
        kind = PAYLOAD_TAXON
    elif ids == READ_ID:
        kind = PAYLOAD_READPOS
    else:
        raise ValueError("UnexpectedDataTypeException: %s" % path)       # KmerFile.java:30-33
    keys, val_off, val_a, val_b, lens, keys_lo = parse.decode_blocks(f.read_keys(first, last), kind, with_lens=True, with_lo=True)
    return KmerDatabase.load(ctx, max_k, key_len, rc, kind, keys, val_off, val_a, val_b, lens=lens, min_length=min_k, keys_lo=keys_lo)


_UNPACK = np.array([[65, 67, 84, 71][(b >> s) & 3] for b in range(256) for s in (6, 4, 2, 0)], np.uint8).reshape(256, 4)


def read_preprocessed(path, packed=False, limit=None):
    """The input of MakeDatabase -p (written by OtherFiles.SeqToTaxID): an IndexedFile keyed by String whose blocks
    hold (taxid, Sequence) records -- binary [int taxid][int length][2 bits per base] (IntCompressor +
    SequenceDataType.java:21-34) or human-readable lines `taxid ~ SEQUENCE`.  -> SequenceBatch of every record
    (MakeDatabase.java:192-237 hands every index to the same FileCreator); with packed=True and a binary file
    -> (packed bytes, seq_off, byte_off, seq_id) for FileCreator.add_packed, the bases left as they are."""
    from .api import SequenceBatch
    f = container.IndexedInputFile(path)
    wanted = sorted(f.index)
    if limit is not None:            # -L start[-end]: input indexes between the two names, inclusive (MakeDatabase.java:207-227)
        start, _, end = str(limit).partition("-")
        wanted = [x for x in wanted if start <= x <= (end or start)]
    if packed and not f.hr:
        chunks, ids, lens = [], [], []
        for key in wanted:
            data = f.read_key(key)
            off = 0
            while off < len(data):
                t, n = struct.unpack_from(">ii", data, off)
                nb = (n - 1) // 4 + 1 if n > 0 else 0
                chunks.append(data[off + 8:off + 8 + nb])
                ids.append(t)
                lens.append(n)
                off += 8 + nb
        lens = np.array(lens, np.uint64)
        seq_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        byte_off = np.concatenate([[0], np.cumsum((lens + np.uint64(3)) // np.uint64(4))]).astype(np.uint64)
        return np.frombuffer(b"".join(chunks), np.uint8), seq_off, byte_off, np.array(ids, np.int32)
    parts, ids, lens = [], [], []
    for key in wanted:
        data = f.read_key(key)
        if f.hr:
            for line in data.decode().splitlines():
                t, s = line.split(" ~ ", 1)
                ids.append(int(t))
                lens.append(len(s))
                parts.append(np.frombuffer(s.encode(), np.uint8))
        else:
            off = 0
            while off < len(data):
                t, n = struct.unpack_from(">ii", data, off)
                nb = (n - 1) // 4 + 1
                packed = np.frombuffer(data, np.uint8, nb, off + 8)
                parts.append(_UNPACK[packed].reshape(-1)[:n])
                ids.append(t)
                lens.append(n)
                off += 8 + nb
    bases = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    seq_off = np.concatenate([[0], np.cumsum(np.array(lens, np.uint64))]).astype(np.uint64)
    return SequenceBatch(bases, seq_off, seq_id=np.array(ids, np.int32))


def make_database(ctx, input_path, output_path, mode, k=32, min_k=24, key_length=6, zip_level=5, read_map_path=None,
                  dust=None, runofsame=None, hr=False, passes=1, limit=None, share=None, id_map_path=None):
    """MakeDatabase -a (mode 'a') / -q (mode 'q') / -p (mode 'p', preprocessed input).  Defaults as the reference's
    (-K 32 -k 24 -l 6 -z 5, MakeDatabase.java:106-109); min_k < k builds k-mers of every length min_k .. k.
    passes > 1 (-P, no counterpart in the reference; modes 'a' and 'q'): the database is built and written in that many
    key ranges, one after the other -- the same file, for inputs that do not fit one build; returns None then.
    share = (r, n) (-G r/n, no counterpart either): this process builds the r-th of n equal key ranges and writes it as
    <output>.<r+1>: n processes, one per GPU, no communication, and the files are a database set as the reference reads
    it (name.1 .. name.N, Matcher.java:80-89); the read map is written by share 0."""
    if min_k > k:
        raise ValueError("-k (shortest k-mer) above -K (longest)")
    if hr and min_k != k:
        raise KmersGpuError(KMG_EUNSUPPORTED, "human-readable databases (-h) hold k-mers of one length on the GPU path")
    # -m (with -a only, MakeDatabase.java:136-147): sequence id -> taxon id, tab separated, gzip or plain (ZipOrNot)
    id_map = open(id_map_path, "rb").read() if (id_map_path is not None and mode == "a") else None
    if (passes > 1 or share is not None) and mode in ("a", "q"):
        from . import parse
        if mode == "q" and read_map_path is None:
            raise ValueError("-r <read map file> is required with -q")
        pf = parse.parse(open(input_path, "rb").read(), parse.FASTA if mode == "a" else parse.FASTQ, k,
                         id_map=id_map if mode == "a" else None)
        nkeys = 4 ** key_length
        r, n = share if share is not None else (0, 1)
        r_lo, r_hi = nkeys * r // n, nkeys * (r + 1) // n
        passes = max(1, min(passes, r_hi - r_lo))
        if n > 1:
            output_path = "%s.%d" % (output_path, r + 1)
        out = container.IndexedOutputFile(output_path, hr=hr, zip_level=zip_level)
        for ps in range(passes):
            rng = (r_lo + (r_hi - r_lo) * ps // passes, r_lo + (r_hi - r_lo) * (ps + 1) // passes)
            if mode == "a":
                db = make_reference_db(ctx, pf, k, key_length, dust, runofsame, key_range=rng, min_k=min_k)
            else:
                db, names = make_read_db(ctx, pf, k, key_length, dust, runofsame, key_range=rng, min_k=min_k)
            write_database(db, output_path, zip_level, hr, out=out, key_range=rng, with_meta=ps == 0)
            db.close()
        out.close()
        if mode == "q" and r == 0:
            with gzip.open(read_map_path, "wt") as g:
                for nm in names:
                    g.write(nm + "\n")
        return None
    if mode == "p":
        # MakeDatabase.java:186-237: reference payload, reverse complements included, every index of the input
        from .api import FileCreator
        fc = FileCreator(ctx, key_length, k, "ref", rc=True, dust=dust, runofsame=runofsame)
        # (PreProcessedSpliterator with minK < maxK, MakeDatabase.java:393-404, hands out from every start the k-mer of
        # min(maxK, bases left) bases: every window plus the suffixes of length maxK-1 .. minK at the sequence's end --
        # what the text path yields for the same, all-valid, sequence)
        got = read_preprocessed(input_path, packed=True, limit=limit)
        if isinstance(got, tuple):                       # binary file: the packed bases go to the device as they are
            if len(got[3]):
                fc.add_packed(*got, min_length=min_k)
        else:
            fc.add_kmers(got, min_length=min_k, max_length=k)
        db = fc.create()
        write_database(db, output_path, zip_level, hr)
        return db
    data = open(input_path, "rb").read()
    if mode == "a":
        db = make_reference_db(ctx, data, k, key_length, dust, runofsame, min_k=min_k, id_map=id_map)
    elif mode == "q":
        if read_map_path is None:      # new File(null) -> NullPointerException in the reference (:172-173)
            raise ValueError("-r <read map file> is required with -q")
        db, names = make_read_db(ctx, data, k, key_length, dust, runofsame, min_k=min_k)
        with gzip.open(read_map_path, "wt") as g:
            for nm in names:
                g.write(nm + "\n")
    else:
        raise ValueError("input mode must be 'a' (FASTA reference) or 'q' (FASTQ reads)")
    write_database(db, output_path, zip_level, hr)
    return db


def matcher(ctx, search_path, db_path, output_path, plain=False, limit=None, max_diff=0, just=False, min_k=None, max_k=None):
    """Matcher with -n 0: one line per matched search k-mer, ascending (Matcher.java:235-250); output is
    gzip at the default level unless plain (-Z x).  min_k / max_k = -k / -K, by default the search file's lengths
    (Matcher.java:147-148): with one length L both sides go through KmerUtils.restrictedStream (KmerFile.java:80-83,
    DB.java:134; kmg_db_restrict) and take the exact join; a search over several lengths -- the default on files built
    with MakeDatabase's default -k 24 -K 32 -- is the trie (DB.java:83) in closed form (kmg_match_trie).  limit = "start" or "start-end" (-L, Matcher.java:159-171): only
    the search k-mers of keys start .. end; a k-mer and its match share their key, so the same slice of the
    reference is all that is loaded."""
    key_range = None
    if limit is not None:
        parts = str(limit).split("-")
        key_range = (int(parts[0]), int(parts[1]) if len(parts) == 2 else int(parts[0]))
    dbfiles = container.file_set(db_path)
    sfiles = container.file_set(search_path)
    if not dbfiles or not sfiles:
        raise FileNotFoundError(db_path if not dbfiles else search_path)
    # name.1 .. name.N are merged (DB.java:99-109); an inexact match can sit under any key close to the search
    # k-mer's (KmerUtils.getCloseKeys), so -L then limits the search side only
    db = DB([open_database(ctx, f, key_range if max_diff == 0 else None) for f in dbfiles])
    searches = [open_database(ctx, f, key_range) for f in sfiles]
    min_k = searches[0].min_length if min_k is None else int(min_k)
    max_k = searches[0].max_length if max_k is None else int(max_k)
    if min_k < db.min_length or max_k > db.max_length:          # DB.java:78-81
        raise InconsistentDataException(KMG_EINVAL, "Search kmers contain kmers of a length inconsistent with the database")
    if min_k != max_k:
        # a search over several lengths: the trie (DB.java:83,139-184) in closed form, kmg_match_trie
        from .api import match_trie, near_match_lines_var
        if any((s_.min_length, s_.max_length) != (min_k, max_k) for s_ in searches):
            searches = [s_.restrict(min_k, max_k) for s_ in searches]       # KmerFile.allRestrictedKmers (Matcher.java:241)
        # sets (name.1 .. name.N): files that share no key -- what -S / -G write -- are laid end to end (kmg_db_merge);
        # the reference's trie takes one key's k-mers file by file: for files that share keys the merged handle keeps them
        # and kmg_match_trie answers from the files in their order (search files that share keys: status 3)
        search = searches[0] if len(searches) == 1 else merge_databases(searches)
        if not db.merged.rc:
            raise ValueError("InconsistentDataException: Files contains different kmer parameters (min/max length, key length or reverse complement included")
        if max_diff > 0:         # -n [-j]: per length, the stored k-mers' prefixes within reach (near_match_lines_var)
            lines = near_match_lines_var(search, db.merged, max_diff, just, min_k, stored=True)
        else:
            rows, hit = match_trie(search, db.merged, min_k, stored=True)
            lines = match_lines(search, rows, hit, stored=True)
            rows.close()
        text = "".join(l + "\n" for l in lines).encode()
        with (open(output_path, "wb") if plain else gzip.open(output_path, "wb")) as f:
            f.write(text)
        return lines
    if any(s.min_length != s.max_length or s.k != min_k for s in searches):
        searches = [s.restrict(min_k) for s in searches]        # KmerFile.allRestrictedKmers (Matcher.java:241)
    if db.min_length != db.max_length or db.max_length != min_k:
        db = DB([f.restrict(min_k) for f in db.files])          # DB.java:134, per key; the files' k-mers merged after
    search = searches[0] if len(searches) == 1 else merge_databases(searches)   # Matcher.java:111-128 via KmerUtils
    if max_diff > 0:        # -n [-j]: DB.getNearestKmers' other branch, k-mers of one length (kmg_match_near)
        from .api import near_match_lines
        if not db.merged.rc:
            raise ValueError("InconsistentDataException: Files contains different kmer parameters (min/max length, key length or reverse complement included")
        lines = near_match_lines(search, db.merged, max_diff, just, stored=True)
    else:
        hit, _ = db.get_nearest_kmers(search)
        lines = match_lines(search, db.merged, hit, stored=True)
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


_CODE = {65: 0, 67: 1, 84: 2, 71: 3}        # A C T G (Base.java:17-41)


def _kmer_key(text):
    """Left-aligned 2-bit value of a k-mer string (Sequence.cBytes read big-endian)."""
    v = 0
    for ch in text.encode():
        v = (v << 2) | _CODE[ch]
    return v << (64 - 2 * len(text))


def _open_text(path):
    """Zip.ZipOrNot.getBufferedReader: gzip if the file starts with the gzip magic."""
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, "rt") if magic == b"\x1f\x8b" else open(path, "r")


def _write_text_blocks(path, blocks):
    """new IndexedOutputFile<>(f, new IntCompressor(), true, 5): human-readable flavour, zip level 5, one block
    per key in ascending order, no meta block (CollectByRead.java:66-67,116-119; ReadClassifier.java:52-53,157-160)."""
    out = container.IndexedOutputFile(path, hr=True, zip_level=5)
    for key in sorted(blocks):
        out.set_current_key(key)
        out.write(blocks[key])
    out.close()


def collect_by_read(ctx, input_path, output_path):
    """Utils.CollectByRead -i <Matcher output> -o <file>.  The match lines are loaded back as a read database and
    a reference database over the same k-mers (every k-mer hits itself) and transposed on the GPU."""
    keys, s_off, s_read, s_pos, r_off, r_tax, r_cnt = [], [0], [], [], [0], [], []
    special = {}
    k = None
    mixed = False
    with _open_text(input_path) as f:
        for line in f:
            line = line.rstrip("\n")
            if not line:
                continue
            kmer, reads_s, matches_s = line.split("\t", 2)
            ents = matches_s.split(" ")
            k = len(kmer) if k is None else k
            if len(kmer) != k or k > 32:
                mixed = True         # several lengths (Matcher's default on default databases): line numbers name the rows
            keys.append(0 if mixed else _kmer_key(kmer))
            for rp in reads_s.split("|"):
                r, p = rp.split(":")
                s_read.append(int(r))
                s_pos.append(int(p))
            s_off.append(len(s_read))
            if len(ents) == 1 and ents[0].startswith("-|"):          # one exact match: its taxa go to the GPU database
                for tc in ents[0][2:].split("|"):
                    t, c = tc.split("x")
                    r_tax.append(int(t))
                    r_cnt.append(int(c))
            else:
                # Matcher -n > 0: the taxa of all entries with the same number of differences added up, one map per
                # distance (CollectByRead.java:165-173); the line keeps an empty taxon list as its mark
                by_dist = {}
                for ent in ents:
                    diff_s, counts_s = ent.split("|", 1)
                    m = by_dist.setdefault(sum(1 for ch in diff_s if ch.isalpha()), {})
                    for tc in counts_s.split("|"):
                        t, c = tc.split("x")
                        m[int(t)] = m.get(int(t), 0) + int(c)
                special[len(keys) - 1] = " ".join("%d:%s" % (d, "|".join("%dx%d" % (t, min(c, 255)) for t, c in sorted(by_dist[d].items())))
                                                  for d in sorted(by_dist))
            r_off.append(len(r_tax))
    if k is None:
        _write_text_blocks(output_path, {})
        return {}
    if mixed:
        keys, k = list(range(len(keys))), 32
    kl = min(k, 6)
    search = KmerDatabase.load(ctx, k, kl, False, PAYLOAD_READPOS, np.array(keys, np.uint64), np.array(s_off, np.uint64),
                               np.array(s_read, np.int32), np.array(s_pos, np.int32))
    ref = KmerDatabase.load(ctx, k, kl, True, PAYLOAD_TAXON, np.array(keys, np.uint64), np.array(r_off, np.uint64),
                            np.array(r_tax + [0], np.int32), np.array(r_cnt + [0], np.int32))     # (+ a sentinel: never an empty array)
    if not special:
        blocks = collect_by_read_blocks(search, ref)
    else:
        from .api import collect_by_read as transpose
        read, pos, idx = transpose(search, ref)
        blocks, lines, cur, last = {}, [], None, None
        for rd, ps, h in zip(read.tolist(), pos.tolist(), idx.tolist()):
            if (rd, ps) == last:                                     # TreeSet.add keeps the element already present
                continue
            last = (rd, ps)
            key = (rd // 1000) * 1000
            if key != cur:
                if cur is not None:
                    blocks[cur] = "\n".join(lines).encode()
                cur, lines = key, []
            text = special.get(h)
            if text is None:
                text = "0:" + "|".join("%dx%d" % (r_tax[t], min(r_cnt[t], 255)) for t in range(r_off[h], r_off[h + 1]))
            lines.append("%d:%d\t%s" % (rd, ps, text))
        if cur is not None:
            blocks[cur] = "\n".join(lines).encode()
    search.close()
    ref.close()
    _write_text_blocks(output_path, blocks)
    return blocks


def read_classifier(ctx, input_path, output_path):
    """Utils.ReadClassifier -i <CollectByRead output> -o <file>.  Every line becomes one pseudo k-mer (its line
    number as a 32-mer) of a read database holding the line's (read, pos) and of a reference database holding its
    dist-0 taxa; kmg_classify_reads then does processRead for all reads at once."""
    f = container.IndexedInputFile(input_path)
    s_read, s_pos, r_off, r_tax, block_reads = [], [], [0], [], {}
    for key in sorted(f.index):
        text = f.read_key(key).decode()
        reads = block_reads.setdefault(key, [])
        for line in (text.split("\n") if text else []):
            rp_s, map_s = line.split("\t", 1)
            r, p = rp_s.split(":")
            if not reads or reads[-1] != int(r):
                reads.append(int(r))
            for ent in map_s.split(" "):
                dist, counts = ent.split(":")
                if int(dist) == 0:                            # ReadClassifier.java:109,125: only the dist-0 map
                    s_read.append(int(r))
                    s_pos.append(int(p))
                    for tc in counts.split("|"):
                        r_tax.append(int(tc.split("x")[0]))
                    r_off.append(len(r_tax))
    n = len(s_read)
    blocks = {key: b"" for key in block_reads}
    if n:
        keys = np.arange(n, dtype=np.uint64)
        one = np.arange(n + 1, dtype=np.uint64)
        search = KmerDatabase.load(ctx, 32, 6, False, PAYLOAD_READPOS, keys, one, np.array(s_read, np.int32), np.array(s_pos, np.int32))
        ref = KmerDatabase.load(ctx, 32, 6, True, PAYLOAD_TAXON, keys, np.array(r_off, np.uint64), np.array(r_tax, np.int32),
                                np.ones(len(r_tax), np.int32))
        n_reads = max(s_read) + 1
        call = classify_reads(search, ref, n_reads)
        search.close()
        ref.close()
        for key, reads in block_reads.items():
            blocks[key] = "".join("%d\t%d\n" % (r, int(call[r])) for r in reads if r < n_reads and call[r] >= 0).encode()
    _write_text_blocks(output_path, blocks)
    return blocks


def read_calls(ctx, reads_path, output_path, db_path=None, fasta_path=None, k=32, key_length=6, read_map_path=None, min_k=None):
    """FASTQ reads + a reference (database file set -d, or FASTA -a) -> the file the reference's four-stage chain
    MakeDatabase -q | Matcher | CollectByRead | ReadClassifier ends with, byte for byte, without the three files in
    between (a 1M-read sample's match text alone is ~600 MB gzipped): both databases stay on the GPU, the join
    feeds kmg_classify_reads_hits, and only `read \t call` lines leave.  No counterpart program in the reference."""
    if (db_path is None) == (fasta_path is None):
        raise ValueError("exactly one of -d (database) / -a (FASTA reference)")
    min_k = k if min_k is None else min_k
    if fasta_path is not None:
        ref = make_reference_db(ctx, open(fasta_path, "rb").read(), k, key_length, min_k=min_k)
    else:
        files = container.file_set(db_path)
        if not files:
            raise FileNotFoundError(db_path)
        ref = DB([open_database(ctx, f) for f in files]).merged
        k, key_length, min_k = ref.k, ref.key_length, ref.min_length
    if not ref.rc:       # DB.java:57-64
        raise ValueError("InconsistentDataException: Files contains different kmer parameters (min/max length, key length or reverse complement included")
    # the reads are built with the database's lengths, as MakeDatabase -q with the same -K / -k would build them
    rdb, names = make_read_db(ctx, open(reads_path, "rb").read(), k, key_length, min_k=min_k)
    if read_map_path is not None:
        with gzip.open(read_map_path, "wt") as g:
            for nm in names:
                g.write(nm + "\n")
    if min_k != k:
        # k-mers of several lengths: Matcher's search is the trie's (DB.java:83,139-184), the calls come from its answers
        from .api import classify_reads_rows, match_trie
        rows, hit = match_trie(rdb, ref, min_k, stored=db_path is not None)
        call, seen = classify_reads_rows(rdb, rows, hit, len(names))
        rows.close()
    else:
        call, seen = classify_reads(rdb, ref, len(names), with_hits=True)
    blocks = read_call_blocks(call, seen)
    rdb.close()
    ref.close()
    _write_text_blocks(output_path, blocks)
    return blocks
