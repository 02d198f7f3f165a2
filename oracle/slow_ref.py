"""Literal, object-by-object CPU restatement of the k-mer hot path of danielmoney/kmers.

TEST INFRASTRUCTURE ONLY.  Nothing under kmers_amd/ may import this module; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.

PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors for this path and
there is no JVM in the build container, so Kmers.jar could not be executed.  This restatement
is pinned only by the hand-derived known-answer vectors of SURVEY.md Appendix B
(tests/test_oracle_kat.py) and by agreement with the independent C restatement
(oracle/kmers_oracle.c).  tools/jar_crosscheck.sh diffs against Kmers.jar when `java` exists.

Every function cites the reference file:line it follows (paths relative to /root/reference/src,
`jar:` = behaviour recovered from out/Kmers.jar bytecode).
Pure-Python loops: use for small inputs only.
"""
import struct
from collections import OrderedDict

# ----------------------------------------------------------------------------- Kmers/Base.java
# Base.java:7-10  A=0 C=1 T=2 G=3 ; Base.java:22-41 fromCharacterByte accepts upper and lower case
_CHAR2CODE = {ord('A'): 0, ord('a'): 0, ord('C'): 1, ord('c'): 1,
              ord('T'): 2, ord('t'): 2, ord('G'): 3, ord('g'): 3}
_CODE2CHAR = "ACTG"


class InvalidBaseException(Exception):
    pass


class InconsistentDataException(Exception):
    pass


def _sbyte(c):
    """Java (byte) cast of an int."""
    c &= 0xFF
    return c - 256 if c >= 128 else c


def _sshort(v):
    """Java (short) cast of an int (KmersFromFile.java:165)."""
    v &= 0xFFFF
    return v - 65536 if v >= 32768 else v


# ------------------------------------------------------------------------- Kmers/Sequence.java
class Kmer:
    """Kmers/Kmer.java + Kmers/Sequence.java.  `chars` = list of base codes 0..3."""

    def __init__(self, chars):
        # Sequence.java:15-35: first byte > 4 => ASCII letters, convert each (may throw);
        # otherwise the bytes are codes already and any byte > 4 throws.  Bytes are signed.
        chars = [_sbyte(c) for c in chars]
        if chars[0] > 4:
            out = []
            for c in chars:
                code = _CHAR2CODE.get(c & 0xFF) if c >= 0 else None
                if code is None:
                    raise InvalidBaseException()
                out.append(code)
            chars = out
        else:
            for c in chars:
                if c > 4:
                    raise InvalidBaseException()
        self.chars = chars

    def copy(self):
        k = Kmer.__new__(Kmer)
        k.chars = list(self.chars)
        return k

    def length(self):
        return len(self.chars)

    def key(self, num):  # Kmer.java:47-55
        key = 0
        for i in range(num):
            key = key * 4 + self.chars[i]
        return key

    def cbytes(self):  # Sequence.java:90-115
        n = len(self.chars)
        l = (n - 1) // 4 + 1
        r = bytearray(l)
        c = 0
        cb = 0
        cc = 0
        for i in range(n):
            cb = ((cb << 2) + self.chars[i]) & 0xFF
            cc += 1
            if cc == 4:
                r[c] = cb
                c += 1
                cc = 0
                cb = 0
        if cc != 0:
            cb = (cb << ((4 - cc) * 2)) & 0xFF
            r[c] = cb
        return bytes(r)

    def compressed_bytes(self):  # Kmer.java:90-97  [len byte][packed]
        return bytes([len(self.chars) & 0xFF]) + self.cbytes()

    @staticmethod
    def from_compressed_seq(seq_bytes, length):  # Sequence.java:236-257
        b = [0] * length
        l = (length - 1) // 4 + 1
        for i in range(l):
            cb = seq_bytes[i]
            for j in range(4):
                p = i * 4 + (3 - j)
                if p < length:
                    b[p] = cb & 3
                cb >>= 2
        return b

    @staticmethod
    def create_from_compressed(data):  # Kmer.java:326-345
        length = data[0]
        return Kmer(Kmer.from_compressed_seq(data[1:], length)) if length > 0 else None

    def inplace_rc(self):  # Kmer.java:142-208: A<->T, C<->G  == code ^ 2, reversed
        comp = {0: 2, 1: 3, 2: 0, 3: 1}
        ch = self.chars
        n = len(ch)
        for i in range(n // 2):
            temp = comp[ch[i]]
            ch[i] = comp[ch[n - i - 1]]
            ch[n - i - 1] = temp
        if n % 2 == 1:
            ch[n // 2] = comp[ch[n // 2]]

    def is_own_rc(self):  # Kmer.java:210-225: a *palindrome* test, even lengths only
        n = len(self.chars)
        if n % 2 == 1:
            return False
        for i in range(n // 2):
            if self.chars[i] != self.chars[n - i - 1]:
                return False
        return True

    def compare_to(self, other):  # Sequence.java:165-187
        a, b = self.chars, other.chars
        for i in range(min(len(a), len(b))):
            if a[i] < b[i]:
                return -1
            if a[i] > b[i]:
                return 1
        if len(a) < len(b):
            return -1
        if len(a) > len(b):
            return 1
        return 0

    def sort_key(self):
        # tuple order == compare_to order (element-wise, shorter-prefix first)
        return tuple(self.chars)

    def __str__(self):  # Sequence.java:117-134
        return "".join(_CODE2CHAR[c] for c in self.chars)

    def __eq__(self, o):
        return isinstance(o, Kmer) and self.chars == o.chars

    def __hash__(self):
        return hash(tuple(self.chars))


# ------------------------------------------------- OtherFiles/KmersFromFileStateChanger.java
ID, KMER, OTHER = "ID", "KMER", "OTHER"


def fa_state_changer():  # KmersFromFileStateChanger.java:46-58
    return OTHER, {(ID, ord(' ')): OTHER, (ID, ord('\n')): KMER, (OTHER, ord('\n')): KMER,
                   (KMER, ord('\n')): OTHER, (OTHER, ord('>')): ID}


def fq_state_changer():  # KmersFromFileStateChanger.java:60-68
    return OTHER, {(OTHER, ord('@')): ID, (KMER, ord('@')): ID, (ID, ord('\n')): KMER,
                   (KMER, ord('+')): OTHER}


class ReadIDMapping:  # OtherFiles/ReadIDMapping.java:5-28
    def __init__(self):
        self.names = []          # what `out.println(id)` writes, in order
        self.cur_s = ""
        self.cur_i = -1

    def ge_next(self, rid):      # ReadIDMapping.java:14-23
        if rid != self.cur_s:
            self.names.append(rid)
            self.cur_s = rid
            self.cur_i += 1
        return self.cur_i


def kmers_from_file(data, state_changer, min_k, max_k, mapper):
    """KmersFromFile.KmersFromFileSpliterator (KmersFromFile.java:65-243) driven to exhaustion.

    `data`: bytes of the (already gunzipped) file.  `mapper(id_str, short_pos)` as the
    BiFunction at KmersFromFile.java:39-52.  Returns the list of (Kmer, payload) in stream order.
    One loop iteration below == one iteration of the do/while at KmersFromFile.java:88-192;
    the outer `while True` == repeated tryAdvance calls.
    """
    start_state, table = state_changer
    out = []
    kmerbytes = [0] * max_k
    state = start_state
    idb = []
    ending = False
    end_file = False
    cur_k = 0
    oldid = ""
    pos = 0
    kwd = None
    ip = 0
    n = len(data)
    while True:                                    # one tryAdvance per outer iteration
        while True:                                # do { ... } while (kwd == null ...)
            broke = False
            if ending:                             # :90-111
                if cur_k < min_k:
                    ending = False
                    kwd = None
                else:
                    newbytes = kmerbytes[max_k - cur_k:max_k]
                    try:
                        kwd = (Kmer(newbytes), mapper(oldid, _sshort(pos - cur_k)))
                    except InvalidBaseException:
                        pass
                    cur_k -= 1
            else:
                if ip >= n:                        # in.read() == -1  :116-123
                    ending = True
                    oldid = "".join(idb)
                    cur_k = min(pos - 1, max_k - 1)
                    end_file = True
                    broke = True
                else:
                    c = data[ip]
                    ip += 1
                    new_state = table.get((state, c))
                    if new_state is not None:      # :126-148
                        if state == KMER and pos >= min_k:
                            oldid = "".join(idb)
                            ending = True
                            cur_k = min(pos, max_k - 1)
                        state = new_state
                        if state == KMER:
                            pos = 0
                        elif state == ID:
                            idb = []
                            kwd = None
                        else:
                            kwd = None
                    else:                          # :149-189
                        if state == KMER:
                            if c > 32:
                                pos += 1
                                kmerbytes = kmerbytes[1:] + [c]
                                if pos >= max_k:
                                    try:
                                        k = Kmer(list(kmerbytes))
                                        kwd = (k, mapper("".join(idb), _sshort(pos - max_k)))
                                    except InvalidBaseException:
                                        kwd = None
                                        if pos > min_k:
                                            ending = True
                                            oldid = "".join(idb)
                                            cur_k = min(pos - 1, max_k - 1)
                            else:
                                kwd = None
                        elif state == ID:
                            idb.append(chr(c))
            if broke:
                break
            if kwd is not None:                    # while ((kwd == null) && (true))
                break
        if kwd is None:
            return out
        out.append((kwd[0].copy(), kwd[1]))
        # NB: kwd is a *field* and survives into the next tryAdvance (quirk C.7: a file that
        # lacks its final newline emits its last k-mer twice) -- so it is not reset here.


def fa_to_refdb_mapper(id_str, pos):  # KmersFromFile.java:39-42
    return int(id_str)  # Integer.parseInt: raises ValueError like NumberFormatException


def make_fq_mapper(read_map):  # KmersFromFile.java:49-52
    return lambda s, i: (read_map.ge_next(s), i)


# -------------------------------------------------------------------- codecs (App. A.4, src/)
def int_compress(v):  # jar:Compression.IntCompressor  4 B big-endian
    return struct.pack(">i", v)


def readpos_compress(rp):  # Reads/ReadPosDataType.java:22-28
    return struct.pack(">ih", rp[0], rp[1])


def count_compress(cmap):  # Counts/CountDataType.java:31-46  (cmap: dict taxon->count)
    b = struct.pack(">h", _sshort(len(cmap)))
    for t in sorted(cmap):
        b += struct.pack(">iB", t, min(cmap[t], 255))
    return b


def count_decompress(buf, off):  # CountDataType.java:83-95 + :48-67
    (n,) = struct.unpack_from(">h", buf, off)
    off += 2
    m = OrderedDict()
    for _ in range(n):
        t, c = struct.unpack_from(">iB", buf, off)
        off += 5
        m[t] = c
    return m, off


def readset_compress(rps):  # jar:Compression.SetCompressor [int n] + n x child
    b = struct.pack(">i", len(rps))
    for rp in rps:
        b += readpos_compress(rp)
    return b


def readset_decompress(buf, off):
    (n,) = struct.unpack_from(">i", buf, off)
    off += 4
    out = []
    for _ in range(n):
        out.append(struct.unpack_from(">ih", buf, off))
        off += 6
    return out, off


def byte_id(ids):  # jar:Compression.Compressor.getByteID
    return struct.pack(">i", len(ids)) + b"".join(struct.pack(">i", i) for i in ids)


REF_ID = [2048]          # CountDataType.java:115-121
READ_ID = [257, 2052]    # jar SetCompressor [257] ++ ReadPosDataType.java:57-62


def kwd_compress(kmer, payload_bytes):  # Kmers/KmerWithDataDataType.java:62-72
    return kmer.compressed_bytes() + payload_bytes


# ------------------------------------------------------------------ KmerFiles/FileCreator.java
class FileCreator:
    """FileCreator.java:31-438 without the disk: the temp bucket file is a dict key -> bytes.

    kind = 'ref'  : DataCollector.getCountInstance  (IntDataType, CountDataType, TreeCountMap)
    kind = 'read' : DataCollector.getReadInstance   (ReadPosDataType, SetDataType, toSet)
    (DataTypes/DataCollector.java:63-81)
    """

    def __init__(self, key_length, kind, rc):
        self.key_length = key_length
        self.maxkey = 4 ** key_length            # FileCreator.java:40-44
        self.kind = kind
        self.rc = rc
        self.min_k = -1
        self.max_k = -1
        self.buckets = {}

    def _payload(self, d):
        return int_compress(d) if self.kind == "ref" else readpos_compress(d)

    def add_kmers(self, kwds, min_k, max_k):     # FileCreator.java:65-110
        if self.min_k == -1:
            self.min_k, self.max_k = min_k, max_k
        if self.min_k != min_k or self.max_k != max_k:
            raise InconsistentDataException(
                "New stream does not have the same min or max kmer length as a previous stream")
        for kmer, d in kwds:
            self.buckets.setdefault(kmer.key(self.key_length), bytearray()).extend(
                kwd_compress(kmer, self._payload(d)))
            if self.rc and not kmer.is_own_rc():  # :83-87
                kmer.inplace_rc()
                self.buckets.setdefault(kmer.key(self.key_length), bytearray()).extend(
                    kwd_compress(kmer, self._payload(d)))

    def meta(self):                              # FileCreator.java:153-164
        ids = REF_ID if self.kind == "ref" else READ_ID
        return bytes([self.min_k & 0xFF, self.max_k & 0xFF, self.key_length & 0xFF]) + \
            byte_id(ids) + bytes([1 if self.rc else 0])

    def make_key(self, index):
        """MakeAndWriteKey.call (FileCreator.java:271-406), non-hr.  Returns
        (block_bytes, entries) with entries = [(Kmer, collection)] ascending."""
        buf = bytes(self.buckets.get(index, b""))
        kmers = {}
        off = 0
        while off < len(buf):                    # :280-303
            length = buf[off]
            l = (length - 1) // 4 + 1
            k = Kmer.create_from_compressed(buf[off:off + l + 1])
            off += l + 1
            if self.kind == "ref":
                (d,) = struct.unpack_from(">i", buf, off)
                off += 4
            else:
                d = struct.unpack_from(">ih", buf, off)
                off += 6
            sk = k.sort_key()
            cm = kmers.get(sk)
            if cm is None:
                cm = (k, {} if self.kind == "ref" else [])
                kmers[sk] = cm
            if self.kind == "ref":               # jar:TreeCountMap.add
                cm[1][d] = cm[1].get(d, 0) + 1
            else:                                # HashSet<ReadPos>.add, identity-hashed objects
                cm[1].append(d)
        entries = []
        out = bytearray()
        prev = bytes(1)                          # :315
        for sk in sorted(kmers):                 # TreeMap ascending, Sequence.compareTo
            k, coll = kmers[sk]
            kc = k.compressed_bytes()
            s = 0                                # shared(): FileCreator.java:244-252
            while s < len(prev) and s < len(kc) and prev[s] == kc[s]:
                s += 1
            prev = kc
            out.append(s & 0xFF)
            out.extend(kc if s == 0 else kc[s:])
            if self.kind == "ref":
                out.extend(count_compress(coll))
            else:
                # Java order is HashSet iteration order (nondeterministic); canonical = ascending
                coll.sort()
                out.extend(readset_compress(coll))
            entries.append((k, coll))
        return bytes(out), entries

    def create(self):
        """FileCreator.create (:117-217) minus the container: {-1: meta, key: block} for all keys."""
        blocks = {-1: self.meta()}
        for i in range(self.maxkey):             # every key gets a (possibly empty) block
            blocks[i] = self.make_key(i)[0]
        return blocks


# ---------------------------------------------------------------------- KmerFiles/KmerFile.java
def parse_meta(meta):  # KmerFile.java:275-289 (binary branch)
    min_k, max_k, key_len = meta[0], meta[1], meta[2]
    (n,) = struct.unpack_from(">i", meta, 3)
    ids = list(struct.unpack_from(">%di" % n, meta, 7))
    rc = meta[7 + 4 * n] == 1
    return min_k, max_k, key_len, ids, rc


def decode_block(block, kind):
    """CompressedKmerSpliterator.tryAdvance (KmerFile.java:200-250) over one key's bytes."""
    out = []
    off = 0
    prevkb = b""
    while off < len(block):
        copy = _sbyte(block[off])
        off += 1
        if copy == 0:
            length = block[off]
            off += 1
        else:
            length = prevkb[0]
        l = (length - 1) // 4 + 1
        if copy > 0:
            kb = prevkb[:copy] + block[off:off + l - copy + 1]
            off += l - copy + 1
        else:
            kb = bytes([length]) + block[off:off + l]
            off += l
        prevkb = kb
        k = Kmer.create_from_compressed(kb)
        if kind == "ref":
            data, off = count_decompress(block, off)
        else:
            data, off = readset_decompress(block, off)
        out.append((k, data))
    return out


# ---------------------------------------------------------------------------- Database/DB.java
class DB:
    """Database/DB.java over one or more decoded reference DBs (each = blocks dict)."""

    def __init__(self, files):
        metas = [parse_meta(f[-1]) for f in files]
        for m in metas:                          # DB.java:36-42
            if m[3] != metas[0][3]:
                raise InconsistentDataException("Files contain different datatypes")
        self.min_k, self.max_k, self.key_length = metas[0][0], metas[0][1], metas[0][2]
        for m in metas:                          # DB.java:57-64
            if (m[0], m[1], m[2]) != (self.min_k, self.max_k, self.key_length) or not m[4]:
                raise InconsistentDataException(
                    "Files contains different kmer parameters (min/max length, key length or "
                    "reverse complement included")
        self.files = files
        self.max_key = 4 ** self.key_length

    def kmers(self, key):
        """DB.kmers (DB.java:99-109): k-way merge (jar:StreamUtils.mergeSortedStreams, ties by
        source index) + groupAndReduce with merger addAll (counts add as longs, quirk C.9)."""
        merged = []
        for si, f in enumerate(self.files):
            for k, data in decode_block(f.get(key, b""), "ref"):
                merged.append((k.sort_key(), si, k, data))
        merged.sort(key=lambda t: (t[0], t[1]))
        out = []
        for sk, _, k, data in merged:
            if out and out[-1][0].sort_key() == sk:
                for t, c in data.items():
                    out[-1][1][t] = out[-1][1].get(t, 0) + c
            else:
                out.append((k, dict(data)))
        return out

    def quick_match(self, search):
        """getNearestKmers quick branch (DB.java:76-90,129-137): group the ascending search
        stream by key; per group two-pointer inner join (jar:StreamUtils.matchTwoStreams --
        both sides advance on a hit) against kmers(key)."""
        res = []
        i = 0
        n = len(search)
        while i < n:
            key = search[i][0].key(self.key_length)
            j = i
            while j < n and search[j][0].key(self.key_length) == key:
                j += 1
            length = search[i][0].length()        # DB.java:132
            ref = restricted_stream(self.kmers(key), self.min_k, self.max_k, length, length, "ref")   # :134
            a, b = i, 0
            while a < j and b < len(ref):
                c = search[a][0].compare_to(ref[b][0])
                if c == 0:
                    res.append((search[a][0], search[a][1], ref[b][1]))
                    a += 1
                    b += 1
                elif c < 0:
                    a += 1
                else:
                    b += 1
            i = j
        return res


def restricted_stream(entries, stream_min, stream_max, min_length, max_length, kind):
    """KmerUtils.restrictedStream (KmerUtils.java:59-74): the stream itself when its lengths are the wanted ones; else
    the k-mers shorter than min_length dropped, the rest cut to max_length (Kmer.limitTo, Kmer.java:277-289), and runs
    of equal neighbours folded with the data type's reducer (MergeableDataType.getKmerReducer: the merger applied to
    the first k-mer's data -- TreeCountMap.addAll for a reference database, Set.addAll of ReadPos objects, which are
    never equal to one another, for a read database)."""
    if stream_min == min_length and stream_max == max_length:
        return entries
    out = []
    for k, data in entries:
        if k.length() < min_length:
            continue
        if max_length < k.length():
            k = Kmer(list(k.chars[:max_length]))
        if out and out[-1][0].chars == k.chars:    # groupAndReduceStream: equal to the previous raw element
            if kind == "ref":
                for t, c in data.items():
                    out[-1][1][t] = out[-1][1].get(t, 0) + c
            else:
                out[-1][1].extend(data)
        else:
            out.append((k, dict(data) if kind == "ref" else list(data)))
    return out


def match_lines(search_blocks, db, min_k=None, max_k=None):
    """Matcher.doMatching (Matcher.java:235-250), read-vs-reference, maxDiff 0, one search file; min_k / max_k =
    Matcher -k / -K (default: the search file's own, Matcher.java:147-148).  Only the quick branch (search k-mers of
    one length, DB.java:83): the search stream is KmerFile.allRestrictedKmers(minK, maxK) (Matcher.java:241) and the
    database side of every key is restricted to that length (DB.java:134).
    Line text: ResultsDataType.java:15-35 -> KMER \\t r:p|r:p \\t -|TxC|TxC
    (KmerWithDataDataType.java:74-82, DataPairDataType.java:43-50, KmerDiff.java:54-67,
    ReadPosDataType.java:46-49, CountDataType.java:109-113 with separators "x" and "|")."""
    f_min, f_max, key_len, ids, rc = parse_meta(search_blocks[-1])
    min_k = f_min if min_k is None else min_k
    max_k = f_max if max_k is None else max_k
    kind = "read" if ids == READ_ID else "ref"
    if min_k < db.min_k or max_k > db.max_k:      # DB.java:78-81
        raise InconsistentDataException("Search kmers contain kmers of a length inconsistent with the database")
    if min_k != max_k:
        raise NotImplementedError("search k-mers of several lengths take the trie branch (near_match_lines)")
    search = []
    for key in range(4 ** key_len):
        search.extend(decode_block(search_blocks.get(key, b""), kind))
    search = restricted_stream(search, f_min, f_max, min_k, max_k, kind)
    lines = []
    for kmer, sdata, rdata in db.quick_match(search):
        if kind == "read":
            s = "|".join("%d:%d" % (r, p) for r, p in sorted(sdata))  # canonical order
        else:
            s = "|".join("%dx%d" % (t, sdata[t]) for t in sorted(sdata))
        r = "|".join("%dx%d" % (t, rdata[t]) for t in sorted(rdata))
        lines.append("%s\t%s\t-|%s" % (kmer, s, r))
    return lines


# ------------------------------------------------------------------- Database/Root.java, Node.java
class _Node:                                     # Node.java:3-15
    __slots__ = ("seq", "children", "data")

    def __init__(self, seq):
        self.seq = list(seq)
        self.children = [None] * 4
        self.data = None


def _merge_counts(a, b):                         # CountDataType.getMerger: TreeCountMap.addAll, in place
    if a is None or b is None:
        # merge.accept on a cleared node (data == null) is a NullPointerException in Java.  Merges only happen at depths
        # >= shortest, where data is never null, so this is unreachable as far as the source reads -- raised rather than
        # skipped, so that a reachable case fails the trie fuzz instead of passing quietly.  (The trie's
        # parity is pinned by this restatement only: the reference holds no fixture of Matcher's default -k / -K output.)
        raise TypeError("NullPointerException: merge into / from a node without data (Root.java:48-51,95-102)")
    for t, c in b.items():
        a[t] = a.get(t, 0) + c


class Root:
    """Database/Root.java, literally: a radix trie over the k-mers of one key.  Data objects are shared between a
    split node and its tail exactly as there (n1.data = n.data, Root.java:79-80)."""

    def __init__(self, kmer_length, shortest):
        self.root = None
        self.kmer_length = kmer_length
        self.shortest = shortest

    def add_kmer(self, chars, data):             # Root.java:29-113
        if self.root is None:
            self.root = _Node(chars)
            self.root.data = data
            return
        s, n, L = 0, self.root, len(chars)
        while s != L:
            d = 0                                # firstDiff (:200-211)
            lim = min(len(n.seq), L - s)
            while d < lim and chars[s + d] == n.seq[d]:
                d += 1
            if d == len(n.seq):
                s += len(n.seq)
                if s >= self.shortest:
                    _merge_counts(n.data, data)
                if s == L:
                    pass                         # exact match: already there
                elif n.children[chars[s]] is not None:
                    n = n.children[chars[s]]
                else:
                    nn = _Node(chars[s:])
                    nn.data = data
                    n.children[chars[s]] = nn
                    s = L
            else:
                b1 = n.seq[d]
                n1 = _Node(n.seq[d:])
                n1.children = n.children
                n1.data = n.data
                n.children = [None] * 4
                n.children[b1] = n1
                if s + d < L:
                    n2 = _Node(chars[s + d:])
                    n2.data = data
                    n.children[chars[s + d]] = n2
                n.seq = n.seq[:d]
                if s + d < self.shortest:
                    n.data = None
                else:
                    _merge_counts(n.data, data)
                s = L

    def closest_kmers(self, target, maxdiff, just):      # Root.java:115-170
        best = ClosestInfoCollector()
        test_dist = maxdiff
        process = []
        if self.root is not None:
            process.append((self.root, 0, []))
        while process:
            n, prevdist, prevseq = process.pop(0)        # pollFirst
            lim = min(len(n.seq), len(target) - len(prevseq))
            curdist = prevdist + sum(1 for i in range(lim) if target[i + len(prevseq)] != n.seq[i])   # numDiff
            if curdist <= test_dist:
                newseq = (prevseq + n.seq)[:len(target)]
                if len(prevseq) + len(n.seq) >= len(target):
                    if curdist < test_dist and just:
                        test_dist = curdist
                        best = ClosestInfoCollector()
                    best.add(newseq, n.data, curdist)
                else:
                    front = [(n.children[i], curdist, list(newseq)) for i in range(4) if n.children[i] is not None]
                    process[0:0] = front                 # addFirst from 3 down to 0: child 0 is polled first
        return best


def _rc_chars(chars):                            # Kmer.rc (Kmer.java:57-88)
    return [c ^ 2 for c in reversed(chars)]


class ClosestInfoCollector:                      # Database/ClosestInfoCollector.java
    def __init__(self):
        self.matched = {}                        # TreeMap<Kmer, CI>: tuple(chars) -> (data, dist); iterated ascending
        self.mindist = 255

    def add(self, chars, data, dist):            # :30-50
        k, rc = tuple(chars), tuple(_rc_chars(chars))
        if rc in self.matched:
            if dist < self.matched[rc][1]:
                del self.matched[rc]
                self.matched[k] = (data, dist)
        else:
            self.matched[k] = (data, dist)
        self.mindist = min(dist, self.mindist)

    def add_all(self, other):                    # :52-59
        for k in sorted(other.matched):
            self.add(list(k), *other.matched[k])
        self.mindist = min(self.mindist, other.mindist)

    def result(self, search):                    # getResult (:82-92) + KmerDiffComparaotr (:119-147)
        out = []
        for k in sorted(self.matched):
            diffs = [(i, k[i]) for i in range(len(search)) if search[i] != k[i]]        # KmerDiff (KmerDiff.java:11-32)
            out.append(((len(diffs), diffs), self.matched[k][0]))
        out.sort(key=lambda x: x[0])             # TreeSet: size first, then (position, base) pairs in turn
        dedup = []
        for key, data in out:                    # equal under the comparator = the same k-mer: cannot happen twice
            if not dedup or dedup[-1][0] != key:
                dedup.append((key, data))
        return [(diffs, data) for (_, diffs), data in dedup]


def get_close_keys(keychars, max_diff):          # KmerUtils.getCloseKeys (KmerUtils.java:18-47)
    def key(chars):
        v = 0
        for c in chars:
            v = v * 4 + c
        return v
    closest, cur = {key(keychars)}, [list(keychars)]
    for _ in range(max_diff):
        nxt = []
        for ck in cur:
            for j in range(len(ck)):
                for b in range(4):
                    nb = list(ck)
                    nb[j] = b
                    nxt.append(nb)
                    closest.add(key(nb))
        cur = nxt
    return sorted(closest)


def kmer_diff_text(diffs):                       # KmerDiff.toString (KmerDiff.java:54-67)
    return "".join("%d%s" % (p, _CODE2CHAR[b]) for p, b in diffs) if diffs else "-"


def near_match_lines(search_blocks, db, max_diff, just, min_k=None, max_k=None):
    """Matcher.doMatching with -n max_diff [-j]; min_k / max_k = Matcher -k / -K (default: the search file's own lengths,
    Matcher.java:147-148): the search stream is KmerFile.allRestrictedKmers(minK, maxK) (Matcher.java:241) and the trie
    is built with those lengths (DB.java:86,154) over ALL the database's k-mers.  DB.getNearestKmers' non-quick branch
    (DB.java:83-90) -> processNearestCommonKey per group of search k-mers with a common key (DB.java:139-184): for
    every close key a Root over that key's k-mers of every file, closestKmers per search k-mer, the collectors
    combined as there; lines as match_lines, the diff text in place of "-"."""
    f_min, f_max, key_len, ids, rc = parse_meta(search_blocks[-1])
    min_k = f_min if min_k is None else min_k
    max_k = f_max if max_k is None else max_k
    if min_k < db.min_k or max_k > db.max_k:      # DB.java:78-81
        raise InconsistentDataException("Search kmers contain kmers of a length inconsistent with the database")
    kind = "read" if ids == READ_ID else "ref"
    lines = []
    roots = {}

    def root_for(key):
        if key not in roots:
            r = Root(max_k, min_k)
            for f in db.files:                                           # DB.java:155-158: file by file, unmerged
                for k, data in decode_block(f.get(key, b""), "ref"):
                    r.add_kmer(list(k.chars), dict(data))
            roots[key] = r
        return roots[key]

    for key in range(4 ** key_len):
        group = restricted_stream(decode_block(search_blocks.get(key, b""), kind), f_min, f_max, min_k, max_k, kind)
        if not group:
            continue
        close = get_close_keys(group[0][0].chars[:key_len], max_diff)
        best = [ClosestInfoCollector() for _ in group]
        for ck in close:
            r = root_for(ck)
            for i, (kmer, _) in enumerate(group):
                newci = r.closest_kmers(list(kmer.chars), max_diff, just)
                old = best[i]
                if newci.mindist == old.mindist or not just:             # DB.java:164-171
                    old.add_all(newci)
                if newci.mindist < old.mindist and just:
                    best[i] = newci
        for (kmer, sdata), ci in zip(group, best):
            res = ci.result(kmer.chars)
            if not res:
                continue
            if kind == "read":
                s = "|".join("%d:%d" % (r_, p) for r_, p in sorted(sdata))
            else:
                s = "|".join("%dx%d" % (t, sdata[t]) for t in sorted(sdata))
            lines.append("%s\t%s\t%s" % (kmer, s, " ".join(
                "%s|%s" % (kmer_diff_text(diffs), "|".join("%dx%d" % (t, data[t]) for t in sorted(data))) for diffs, data in res)))
    return lines


# --------------------------------------------------------------------- Utils/TaxaCounts.java
def taxa_counts(db):
    """TaxaCounts default mode (TaxaCounts.java:119-150, 355-360, 386-395, 427-436)."""
    match, unique = {}, {}
    for key in range(db.max_key):
        for _, data in db.kmers(key):
            for t in data:
                match[t] = match.get(t, 0) + 1
            if len(data) == 1:
                t = next(iter(data))
                unique[t] = unique.get(t, 0) + 1
    return ["%d\t%d\t%d" % (t, match[t], unique.get(t, 0)) for t in sorted(match)]


# --------------------------------------------------------------------- Utils/CollectByRead.java
def _counts_from_string(s):
    """CountDataType.fromString with separators "x" and "|" (CountDataType.java:97-107)."""
    m = {}
    for p in s.split("|"):
        t, c = p.split("x")
        m[int(t)] = int(c)
    return m


def _counts_to_string(m):
    """CountDataType.toString (CountDataType.java:109-113): ascending taxon (TreeCountMap)."""
    return "|".join("%dx%d" % (t, m[t]) for t in sorted(m))


def collect_by_read(match_lines_):
    """CollectByRead.main (CollectByRead.java:62-83) over the lines of a Matcher output (ResultsFile.java:23-26).
    processKmer (:159-184): dist -> summed taxon counts of the line's matches; one record (ReadPos, that map) per
    ReadPos of the line, filed under (read / 1000) * 1000.  The record goes through odt.compress (:177), whose
    CountDataType.compress caps every count at 255 (CountDataType.java:40).  ProcessIndex (:105-124): a TreeSet
    ordered by ReadPos alone (an equal ReadPos met later is dropped), printed `r:p \t dist:TxC|TxC dist:...`
    (DataPairDataType.java:43-50, jar MapCompressor.toString: entries `key:value` joined by " ") and joined
    by "\n" without a trailing newline.  -> {index: text bytes}, the blocks of the output IndexedFile."""
    per = {}
    for line in match_lines_:
        kmer, rest = line.split("\t", 1)                       # KmerWithDataDataType.fromString
        reads_s, matches_s = rest.split("\t", 1)               # DataPairDataType "\t", limit 2
        dist_taxa = {}
        for ent in matches_s.split(" "):                       # SetDataType " "
            diff_s, counts_s = ent.split("|", 1)               # DataPairDataType "|", limit 2
            # KmerDiff.dist() = number of (position, base) pairs of the text `3G17A`; "-" holds none (KmerDiff.java:54-72)
            d = dist_taxa.setdefault(sum(1 for ch in diff_s if ch.isalpha()), {})
            for t, c in _counts_from_string(counts_s).items():
                d[t] = d.get(t, 0) + c                         # TreeCountMap.addAll
        capped = {dist: {t: min(c, 255) for t, c in m.items()} for dist, m in dist_taxa.items()}
        for rp in reads_s.split("|"):                          # SetDataType<ReadPos> "|"
            r, p = rp.split(":")
            r, p = int(r), _sshort(int(p))
            per.setdefault((r // 1000) * 1000, []).append(((r, p), capped))
    out = {}
    for index in sorted(per):
        seen, rows = set(), []
        for rp, m in per[index]:
            if rp not in seen:                                 # TreeSet.add keeps the element already present
                seen.add(rp)
                rows.append((rp, m))
        rows.sort(key=lambda x: x[0])                          # ReadPos.compareTo: read, then (short) pos
        out[index] = "\n".join("%d:%d\t%s" % (rp[0], rp[1], " ".join("%d:%s" % (dist, _counts_to_string(m[dist]))
                                                                     for dist in sorted(m))) for rp, m in rows).encode()
    return out


# --------------------------------------------------------------------- Utils/ReadClassifier.java
def _process_read(taxids):
    """ReadClassifier.ProcessIndex.processRead (ReadClassifier.java:178-220), literally."""
    call = -1
    for tids in taxids:
        if len(tids) == 1:
            cur = next(iter(tids))
            if call == -1:
                call = cur
            elif cur != call:
                call = -2
    if call >= 0:
        for tids in taxids:
            if call not in tids:
                call = -2
    return call if call >= 0 else None


def read_classifier(collect_blocks):
    """ReadClassifier.ProcessIndex.call (ReadClassifier.java:95-176) over the blocks of a CollectByRead file:
    per block the lines in order, a read's taxon sets are the key sets of its dist-0 maps; one line
    `read \t call \n` per called read.  Every input block gives an output block (possibly empty)."""
    out = {}
    for index in sorted(collect_blocks):
        text = collect_blocks[index].decode()
        lines = text.split("\n") if text else []
        called, cur_read, taxids = [], -1, []
        first = True
        for line in lines:
            rp_s, map_s = line.split("\t", 1)
            read = int(rp_s.split(":")[0])
            m = {}
            for ent in map_s.split(" "):                       # jar MapCompressor.fromString
                kv = ent.split(":")
                m[int(kv[0])] = _counts_from_string(kv[1])
            if first:
                cur_read, first = read, False
            elif read != cur_read:
                call = _process_read(taxids)
                if call is not None:
                    called.append((cur_read, call))
                cur_read, taxids = read, []
            if 0 in m:
                taxids.append(set(m[0]))
        call = _process_read(taxids)
        if call is not None:
            called.append((cur_read, call))
        out[index] = "".join("%d\t%d\n" % rc for rc in called).encode()
    return out


# ------------------------------------------------ jar IndexedFiles.* 
def _gz_member(data, level):
    """jar GzipCompressorOutputStream + GzipParameters (A.2): mtime 0, XFL 2/4/0, OS 255, raw deflate."""
    import zlib
    xfl = 2 if level == 9 else (4 if level == 1 else 0)
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = co.compress(data) + co.flush()
    return bytes([0x1F, 0x8B, 8, 0, 0, 0, 0, 0, xfl, 0xFF]) + body + \
        struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


def write_indexed_file(path, blocks, zip_level=5, hr=False):
    """jar IndexedOutputFile (A.1): `blocks` = {key: bytes} as FileCreator.create writes them (key -1 first,
    then 0 .. 4^l-1, one block each).  zip_level None = unzipped (41-byte header), else zipped (64-byte).
    hr = the human-readable flavour (MakeDatabase -h): text index, footer flag 1."""
    z = zip_level is not None
    body = bytearray(b"\0" * (64 if z else 41))
    index = []
    for key in ([-1] if -1 in blocks else []) + [k for k in sorted(blocks) if k >= 0]:   # (downstream tools: no meta block)
        blob = _gz_member(blocks[key], zip_level) if z else blocks[key]
        index.append((key, len(body), len(blob)))
        body += blob
    if hr:                                                   # one text line per key: key \t pos:len
        idx = "".join("%d\t%d:%d\n" % (key, pos, ln) for key, pos, ln in sorted(index)).encode()
    else:
        idx = struct.pack(">i", len(index))
        for key, pos, ln in sorted(index):                   # TreeMap order: -1 first
            idx += struct.pack(">ii", key, 1) + struct.pack(">qq", pos, ln)
    iblob = _gz_member(idx, zip_level) if z else idx
    ipos = len(body)
    body += iblob
    footer = ("%-19d\n%-19d\n%s\n%1d\n" % (ipos, len(iblob), "1", 1 if hr else 0)).encode()
    fblob = _gz_member(footer, 0) if z else footer
    fpos = len(body)
    body += fblob
    header = ("%-19d %-19d%s" % (fpos, len(fblob), " \n")).encode()
    hblob = _gz_member(header, 0) if z else header
    body[0:len(hblob)] = hblob
    with open(path, "wb") as f:
        f.write(bytes(body))


# ---------------------------------------------------- MakeDatabase -p input (OtherFiles/SeqToTaxID.java output)
def sequence_compress(seq):
    """Sequence.compressedBytes (Sequence.java:82-115): [int length] + 2 bits per base, first base in the top bits."""
    code = {"A": 0, "C": 1, "T": 2, "G": 3}
    out = bytearray(struct.pack(">i", len(seq)))
    for i in range(0, len(seq), 4):
        chunk = seq[i:i + 4]
        b = 0
        for ch in chunk:
            b = (b << 2) | code[ch]
        out.append((b << (2 * (4 - len(chunk)))) & 0xFF)
    return bytes(out)


def write_preprocessed_file(path, groups, zip_level=5, hr=False):
    """The file SeqToTaxID writes and MakeDatabase -p reads (SeqToTaxID.java:127-135,190-200; MakeDatabase.java:192,
    320-420): an IndexedFile keyed by String (jar StringCompressor: [short length][bytes], compressor id "2"),
    one block per key; a block holds records [int taxid][Sequence.compressedBytes] or, human-readable, lines
    `taxid ~ SEQUENCE` (DataPairDataType's default separator, DataPairDataType.java:17-20).
    groups = {index string: [(taxid, "ACGT...")]}."""
    z = zip_level is not None
    body = bytearray(b"\0" * (64 if z else 41))
    index = []
    for key in sorted(groups):
        if hr:
            raw = "".join("%d ~ %s\n" % (t, s) for t, s in groups[key]).encode()
        else:
            raw = b"".join(struct.pack(">i", t) + sequence_compress(s) for t, s in groups[key])
        blob = _gz_member(raw, zip_level) if z else raw
        index.append((key, len(body), len(blob)))
        body += blob
    if hr:
        idx = "".join("%s\t%d:%d\n" % (key, pos, ln) for key, pos, ln in index).encode()
    else:
        idx = struct.pack(">i", len(index))
        for key, pos, ln in index:
            kb = key.encode()
            idx += struct.pack(">h", len(kb)) + kb + struct.pack(">i", 1) + struct.pack(">qq", pos, ln)
    iblob = _gz_member(idx, zip_level) if z else idx
    ipos = len(body)
    body += iblob
    footer = ("%-19d\n%-19d\n%s\n%1d\n" % (ipos, len(iblob), "2", 1 if hr else 0)).encode()
    fblob = _gz_member(footer, 0) if z else footer
    fpos = len(body)
    body += fblob
    header = ("%-19d %-19d%s" % (fpos, len(fblob), " \n")).encode()
    hblob = _gz_member(header, 0) if z else header
    body[0:len(hblob)] = hblob
    with open(path, "wb") as f:
        f.write(bytes(body))


def preprocessed_kmers_as_fasta(groups):
    """PreProcessedSpliterator.tryAdvance (MakeDatabase.java:358-407) with minK == maxK emits every window of
    every sequence with its taxid -- what the FASTA path emits for the same sequences under the same ids (the
    stored sequences hold valid bases only).  -> the equivalent FASTA bytes, for build_ref_db."""
    return b"".join(b">%d\n%s\n" % (t, s.encode()) for key in sorted(groups) for t, s in groups[key])


def preprocessed_kmers(groups, min_k, max_k):
    """PreProcessedSpliterator.tryAdvance (MakeDatabase.java:358-407), literally, over every index of the input in
    turn (MakeDatabase.java:192-237): sequences shorter than minK are skipped; from every start the k-mer of
    min(maxK, bases left) bases, until fewer than minK bases are left.  -> [(Kmer, taxid)] in stream order."""
    code = {"A": 0, "C": 1, "T": 2, "G": 3}
    out = []
    for key in sorted(groups):
        records = iter(groups[key])
        curseq, curid, curstart = None, None, 0
        while True:
            done = False
            while curseq is None or len(curseq) < min_k:          # :361-391
                nxt = next(records, None)
                if nxt is None:
                    done = True
                    break
                curid, curseq, curstart = nxt[0], nxt[1], 0
            if done:
                break
            end = min(curstart + max_k, len(curseq))              # :393-396
            piece = curseq[curstart:end]
            curstart += 1
            if len(curseq) - curstart < min_k:                    # :399-402
                curseq = None
            out.append((Kmer([code[c] for c in piece.upper()]), curid))   # Kmer.createUnchecked(raw codes)
    return out


def hr_database_blocks(k, key_len, rc, kind, keys, val_off, val_a, val_b, bucket_off):
    """The blocks FileCreator.create writes with hr = true (MakeDatabase -h), from a database's CSR arrays:
    meta = minK, maxK, keyLength, Compressor.getStringID(id), rc as text lines (FileCreator.java:166-179); per
    key, one line per k-mer ascending: diff(last, next) + TAB + values + NEWLINE, where last starts as k spaces
    in every key block and diff drops the leading characters next shares with last (FileCreator.java:234-242,
    349-370); values = CountDataType.toString `TxC|TxC` with the UNSATURATED count (CountDataType.java:109-113)
    or SetDataType<ReadPos> with its default separator: `r:p r:p` (DataCollector.java:67, ReadPosDataType.java:46-49)."""
    ids = "2048" if kind == 0 else "257 2052"
    blocks = {-1: ("%d\n%d\n%d\n%s\n%s\n" % (k, k, key_len, ids, "1" if rc else "0")).encode()}
    letters = "ACTG"
    for key in range(4 ** key_len):
        last = " " * k
        out = []
        for u in range(int(bucket_off[key]), int(bucket_off[key + 1])):
            nxt = "".join(letters[(int(keys[u]) >> (62 - 2 * i)) & 3] for i in range(k))
            first = 0
            while last[first] == nxt[first]:
                first += 1
            a, b = int(val_off[u]), int(val_off[u + 1])
            if kind == 0:
                vals = "|".join("%dx%d" % (int(val_a[v]), int(val_b[v])) for v in range(a, b))
            else:
                vals = " ".join("%d:%d" % (int(val_a[v]), int(val_b[v])) for v in range(a, b))
            out.append(nxt[first:] + "\t" + vals + "\n")
            last = nxt
        blocks[key] = "".join(out).encode()
    return blocks


def read_indexed_file(path):
    """jar IndexedInputFile.<init> + getInputStream(key) (A.1) -> {key: bytes}."""
    import zlib
    data = open(path, "rb").read()
    z = data[:2] == b"\x1f\x8b"

    def blk(pos, ln):
        raw = data[pos:pos + ln]
        if not z:
            return raw
        d = zlib.decompressobj(31)
        return d.decompress(raw) + d.flush()
    import re
    fpos, flen = [int(x) for x in re.split(" +", blk(0, 100)[:41].decode().strip())[:2]]
    lines = blk(fpos, flen).decode().split("\n")
    raw = blk(int(lines[0].strip()), int(lines[1].strip()))
    if lines[3].strip() == "1":                              # hr index: key \t pos:len pos:len
        out = {}
        for line in raw.decode().splitlines():
            key, rest = line.split("\t")
            out[int(key)] = b"".join(blk(*[int(x) for x in part.split(":")]) for part in rest.split())
        return out
    (n,) = struct.unpack_from(">i", raw, 0)
    off, out = 4, {}
    for _ in range(n):
        key, nb = struct.unpack_from(">ii", raw, off)
        off += 8
        parts = []
        for _ in range(nb):
            pos, ln = struct.unpack_from(">qq", raw, off)
            off += 16
            parts.append(blk(pos, ln))
        out[key] = b"".join(parts)
    return out


# ----------------------------------------------------------------- Kmers/Dust.java, RunOfSame
def dust_test(kmer, threshold):  # Dust.java:12-30
    counts = [0] * 64
    ch = kmer.chars
    for i in range(len(ch) - 2):
        counts[ch[i] * 16 + ch[i + 1] * 4 + ch[i + 2]] += 1
    score = sum(c * (c - 1) for c in counts)
    return score < threshold * len(ch) * 2


def run_of_same_test(kmer, length):  # RunOfSame.java:12-36
    ch = kmer.chars
    c = ch[0]
    l = 1
    for i in range(1, len(ch)):
        if ch[i] == c:
            l += 1
            if l == length:
                return False
        else:
            c = ch[i]
            l = 1
    return True


def filter_stream(kwds, dust=None, runofsame=None):
    """MakeDatabase.filterAndAdd (MakeDatabase.java:279-291)."""
    if dust is not None:
        kwds = [kw for kw in kwds if dust_test(kw[0], dust)]
    if runofsame is not None:
        kwds = [kw for kw in kwds if run_of_same_test(kw[0], runofsame)]
    return kwds


# --------------------------------------------------------------------- convenience pipelines
def read_id_map(text):
    """MakeDatabase -m (MakeDatabase.java:140-145): br.lines().forEach(line -> { parts = line.split("\\t");
    map.put(parts[0], Integer.parseInt(parts[1])); }).  `text`: the file's bytes (already gunzipped)."""
    import re
    lines = re.split("\r\n|\n|\r", text.decode("latin-1"))
    if lines and lines[-1] == "":
        lines.pop()                                # BufferedReader.lines: no empty line after the last terminator
    m = {}
    for line in lines:
        parts = line.split("\t")
        while parts and parts[-1] == "":           # String.split drops trailing empty strings
            parts.pop()
        v = parts[1]                               # IndexError ~ ArrayIndexOutOfBoundsException
        if not re.fullmatch("[+-]?[0-9]+", v) or not -2 ** 31 <= int(v) < 2 ** 31:
            raise ValueError("NumberFormatException: " + v)
        m[parts[0]] = int(v)
    return m


def build_ref_db(fasta_bytes, k, key_length, min_k=None, rc=True, dust=None, runofsame=None, id_map=None):
    """MakeDatabase -a (MakeDatabase.java:127-163): FASTA -> reference DB blocks.  id_map: the dict of -m
    (KmersFromFile.java:44-47: the mapper is map.get(header id) instead of Integer.parseInt)."""
    min_k = k if min_k is None else min_k
    fc = FileCreator(key_length, "ref", rc)
    mapper = fa_to_refdb_mapper if id_map is None else (lambda id_str, pos: id_map[id_str])   # KeyError ~ the null's NPE
    kw = kmers_from_file(fasta_bytes, fa_state_changer(), min_k, k, mapper)
    fc.add_kmers(filter_stream(kw, dust, runofsame), min_k, k)
    return fc.create()


def build_read_db(fastq_bytes, k, key_length, min_k=None, dust=None, runofsame=None):
    """MakeDatabase -q (MakeDatabase.java:165-185): FASTQ -> read DB blocks + read-name map."""
    min_k = k if min_k is None else min_k
    rm = ReadIDMapping()
    fc = FileCreator(key_length, "read", False)
    kw = kmers_from_file(fastq_bytes, fq_state_changer(), min_k, k, make_fq_mapper(rm))
    fc.add_kmers(filter_stream(kw, dust, runofsame), min_k, k)
    return fc.create(), rm.names
