"""Host-side mirror of the reference's interfaces for the hot path, on top of the C ABI.

Names, argument meaning and error behaviour follow the Java classes they stand in for
(paths relative to the reference's src/):
  FileCreator  <- KmerFiles/FileCreator.java        (ctor :33, addKmers :65, create :117, close :219)
  KmerDatabase <- KmerFiles/KmerFile.java           (decoded DB: meta + per-key k-mers)
  DB           <- Database/DB.java                  (ctor checks :30-74, getNearestKmers :76-90, kmers :99)
  taxa_counts  <- Utils/TaxaCounts.java             (default mode :119-150)
  match_lines  <- DataTypes/ResultsDataType.java    (toString chain, :15-35)
All compute happens in libkmers_gpu.so; nothing here falls back to the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (InconsistentDataException, KmersGpuError, PAYLOAD_READPOS, PAYLOAD_TAXON)

REF_ID = [2048]          # Counts/CountDataType.java:115-121
READ_ID = [257, 2052]    # jar SetCompressor [257] ++ Reads/ReadPosDataType.java:57-62


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One per (process, GPU): owns the HIP stream and the device-memory pool."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.kmg_create(device, C.byref(h))
        if rc != 0:
            raise KmersGpuError(rc, "kmg_create failed (is a GPU visible?)")
        self.h = h
        self.device = device

    def check(self, rc):
        if rc == 0:
            return
        msg = self.lib.kmg_last_error(self.h).decode()
        if rc == _lib.KMG_EINVAL:
            raise InconsistentDataException(rc, msg)
        raise KmersGpuError(rc, msg)

    @property
    def stream(self):
        return self.lib.kmg_stream(self.h)

    def close(self):
        if self.h:
            self.lib.kmg_destroy(self.h)
            self.h = None

    # ---- measurement hook
    def prof_enable(self, on=True):
        self.check(self.lib.kmg_prof_enable(self.h, 1 if on else 0))

    def prof_reset(self):
        self.check(self.lib.kmg_prof_reset(self.h))

    def prof(self):
        out = {}
        for i in range(self.lib.kmg_prof_count()):
            ms, n, b = C.c_double(), C.c_uint64(), C.c_uint64()
            self.check(self.lib.kmg_prof_get(self.h, i, C.byref(ms), C.byref(n), C.byref(b)))
            out[self.lib.kmg_prof_name(i).decode()] = {"ms": ms.value, "launches": n.value, "bytes": b.value}
        return out


class SequenceBatch:
    """What the host-side record splitter hands to FileCreator.addKmers: the characters > 32 of each
    sequence back to back, their offsets and, per sequence, the taxon (reference) or a header token
    (reads; equal tokens <=> equal header strings)."""

    def __init__(self, bases, seq_off, seq_id=None, seq_tok=None):
        self.bases = np.ascontiguousarray(bases, dtype=np.uint8)
        self.seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        self.seq_id = None if seq_id is None else np.ascontiguousarray(seq_id, dtype=np.int32)
        self.seq_tok = None if seq_tok is None else np.ascontiguousarray(seq_tok, dtype=np.int64)
        self.nseq = len(self.seq_off) - 1 if len(self.seq_off) else 0


class KmerDatabase:
    """A device-resident database (the decoded content of one KmerFile)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.h = handle
        k, kl, rc, kind = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        u, v = C.c_uint64(), C.c_uint64()
        ctx.check(ctx.lib.kmg_db_info(handle, C.byref(k), C.byref(kl), C.byref(rc), C.byref(kind), C.byref(u), C.byref(v)))
        self.k, self.key_length, self.rc, self.kind = k.value, kl.value, bool(rc.value), kind.value
        self.n_unique, self.n_values = u.value, v.value
        self.max_length = self.k
        self.min_length = ctx.lib.kmg_db_min_k(handle)
        self._host = None

    @classmethod
    def load(cls, ctx, k, key_length, rc, kind, keys, val_off, val_a, val_b, lens=None, min_length=None, keys_lo=None):
        """lens / min_length: a database of k-mers of several lengths (kmg_db_load_var).
        keys_lo: bases 32 .. k-1 of k-mers of more than 32 bases (kmg_db_load_wide)."""
        keys = np.ascontiguousarray(keys, np.uint64)
        val_off = np.ascontiguousarray(val_off, np.uint64)
        val_a = np.ascontiguousarray(val_a, np.int32)
        val_b = np.ascontiguousarray(val_b, np.int32)
        h = C.c_void_p()
        if k > 32:
            keys_lo = np.ascontiguousarray(keys_lo, np.uint64)
            var = min_length is not None and min_length != k
            lens = np.ascontiguousarray(lens, np.uint8) if var else None
            ctx.check(ctx.lib.kmg_db_load_wide(ctx.h, k, min_length if var else k, key_length, 1 if rc else 0, kind, _ptr(keys),
                                               _ptr(keys_lo), _ptr(lens), len(keys), _ptr(val_off), _ptr(val_a), _ptr(val_b), C.byref(h)))
        elif min_length is not None and min_length != k:
            lens = np.ascontiguousarray(lens, np.uint8)
            ctx.check(ctx.lib.kmg_db_load_var(ctx.h, k, min_length, key_length, 1 if rc else 0, kind, _ptr(keys), _ptr(lens),
                                              len(keys), _ptr(val_off), _ptr(val_a), _ptr(val_b), C.byref(h)))
        else:
            ctx.check(ctx.lib.kmg_db_load(ctx.h, k, key_length, 1 if rc else 0, kind, _ptr(keys), len(keys), _ptr(val_off),
                                          _ptr(val_a), _ptr(val_b), C.byref(h)))
        return cls(ctx, h)

    def lens(self):
        """Number of bases of every k-mer (kmg_db_fetch_lens)."""
        out = np.empty(max(self.n_unique, 1), np.uint8)
        self.ctx.check(self.ctx.lib.kmg_db_fetch_lens(self.h, _ptr(out)))
        return out[:self.n_unique]

    def restrict(self, length, max_length=None):
        """KmerUtils.restrictedStream(.., length, length, ..) over the whole database (kmg_db_restrict); with max_length
        > length the stream function with two lengths (kmg_db_restrict_range)."""
        h = C.c_void_p()
        if max_length is None or int(max_length) == int(length):
            self.ctx.check(self.ctx.lib.kmg_db_restrict(self.ctx.h, self.h, int(length), C.byref(h)))
        else:
            self.ctx.check(self.ctx.lib.kmg_db_restrict_range(self.ctx.h, self.h, int(length), int(max_length), C.byref(h)))
        return KmerDatabase(self.ctx, h)

    @classmethod
    def load_tensors(cls, ctx, k, key_length, rc, kind, keys, val_off, val_a, val_b):
        """load() from torch tensors on the context's GPU (int64 key bits / offsets, int32 values): kmg_db_load_dev.
        The caller synchronises the stream that produced them first."""
        import torch
        want = torch.device("cuda", ctx.device)
        for t, dt in ((keys, torch.int64), (val_off, torch.int64), (val_a, torch.int32), (val_b, torch.int32)):
            if t.device != want or t.dtype != dt or t.dim() != 1:
                raise ValueError("load_tensors wants 1-D int64/int64/int32/int32 tensors on %s" % want)
        if val_off.numel() != keys.numel() + 1 or val_a.numel() != val_b.numel():
            raise ValueError("val_off must hold one more element than keys; val_a and val_b the same number")
        keys, val_off, val_a, val_b = keys.contiguous(), val_off.contiguous(), val_a.contiguous(), val_b.contiguous()
        h = C.c_void_p()
        ctx.check(ctx.lib.kmg_db_load_dev(ctx.h, k, key_length, 1 if rc else 0, kind, keys.data_ptr(), keys.numel(),
                                          val_off.data_ptr(), val_a.data_ptr() if val_a.numel() else None,
                                          val_b.data_ptr() if val_b.numel() else None, val_a.numel(), C.byref(h)))
        return cls(ctx, h)

    def to_tensors(self, device=None):
        """-> (keys int64[U], val_off int64[U+1], val_a int32[V], val_b int32[V]) torch tensors on the context's
        GPU, copied device to device (kmg_db_copy_dev)."""
        import torch
        dev = torch.device("cuda", self.ctx.device) if device is None else torch.device(device)
        U, V = self.n_unique, self.n_values
        out = (torch.empty(U, dtype=torch.int64, device=dev), torch.empty(U + 1, dtype=torch.int64, device=dev),
               torch.empty(V, dtype=torch.int32, device=dev), torch.empty(V, dtype=torch.int32, device=dev))
        torch.cuda.synchronize(dev)
        for which, t in enumerate(out):
            self.ctx.check(self.ctx.lib.kmg_db_copy_dev(self.h, which, t.data_ptr() if t.numel() else None, 0, t.numel()))
        return out

    def bucket_offsets(self):
        """Host copy of bucket_off[4^l + 1] alone (index of every key bucket's first k-mer)."""
        if self._host is not None:
            return self._host["bucket_off"]
        boff = np.empty(4 ** self.key_length + 1, np.uint64)
        self.ctx.check(self.ctx.lib.kmg_db_fetch(self.h, None, None, None, None, _ptr(boff)))
        return boff

    def data_id(self):
        return REF_ID if self.kind == PAYLOAD_TAXON else READ_ID

    def meta(self):
        """The key -1 block, FileCreator.java:153-164."""
        ids = self.data_id()
        b = bytes([self.min_length, self.k, self.key_length]) + len(ids).to_bytes(4, "big")
        for i in ids:
            b += i.to_bytes(4, "big")
        return b + bytes([1 if self.rc else 0])

    def fetch(self):
        """-> dict of host CSR arrays (cached)."""
        if self._host is None:
            U, V = self.n_unique, self.n_values
            keys = np.empty(U, np.uint64)
            val_off = np.empty(U + 1, np.uint64)
            val_a = np.empty(V, np.int32)
            val_b = np.empty(V, np.int32)
            boff = np.empty(4 ** self.key_length + 1, np.uint64)
            self.ctx.check(self.ctx.lib.kmg_db_fetch(self.h, _ptr(keys), _ptr(val_off), _ptr(val_a), _ptr(val_b), _ptr(boff)))
            self._host = dict(keys=keys, val_off=val_off, val_a=val_a, val_b=val_b, bucket_off=boff)
            if self.k > 32:
                lo = np.empty(max(U, 1), np.uint64)
                self.ctx.check(self.ctx.lib.kmg_db_fetch_keys_lo(self.h, _ptr(lo)))
                self._host["keys_lo"] = lo[:U]
        return self._host

    def encode_blocks(self):
        """Per-key block bytes as MakeAndWriteKey writes them -> (bytes, block_off[4^l+1])."""
        n = C.c_uint64()
        boff = np.empty(4 ** self.key_length + 1, np.uint64)
        self.ctx.check(self.ctx.lib.kmg_db_encode(self.ctx.h, self.h, None, 0, C.byref(n), None))
        out = np.empty(max(n.value, 1), np.uint8)
        self.ctx.check(self.ctx.lib.kmg_db_encode(self.ctx.h, self.h, _ptr(out), n.value, C.byref(n), _ptr(boff)))
        return out[:n.value].tobytes(), boff

    def close(self):
        if self.h:
            self.ctx.lib.kmg_db_free(self.h)
            self.h = None


class FileCreator:
    """KmerFiles/FileCreator.java.  data_kind: 'ref' = DataCollector.getCountInstance (taxon payload,
    the reference passes rc=True, MakeDatabase.java:130), 'read' = getReadInstance (rc=False, :167)."""

    def __init__(self, ctx, key_length, max_kmer_length, data_kind, rc, dust=None, runofsame=None, key_range=None):
        """dust / runofsame: the -D / -R stream filters of MakeDatabase.filterAndAdd (MakeDatabase.java:279-291),
        fused into the extraction kernel.  key_range = (first, end): only the k-mers of those key buckets are built
        (kmg_build_set_key_range) -- one pass of a database built in key ranges, or one rank's share."""
        self.ctx = ctx
        self.key_length = key_length
        self.max_k = max_kmer_length
        self.kind = PAYLOAD_TAXON if data_kind == "ref" else PAYLOAD_READPOS
        self.rc = rc
        self.min_k = -1
        self._min_set = False
        self.n_windows = 0
        ctx.check(ctx.lib.kmg_build_begin(ctx.h, max_kmer_length, key_length, 1 if rc else 0, self.kind))
        if dust is not None or runofsame is not None:
            ctx.check(ctx.lib.kmg_build_set_filters(ctx.h, -1 if dust is None else dust,
                                                    -1 if runofsame is None else runofsame))
        if key_range is not None:
            ctx.check(ctx.lib.kmg_build_set_key_range(ctx.h, int(key_range[0]), int(key_range[1])))

    def add_kmers(self, batch, min_length=None, max_length=None, flags=False):
        """addKmers (FileCreator.java:65-110).  Returns, for read DBs, the read index of every sequence
        (-1 = no k-mer, so no entry in the read-name map); for reference DBs with flags=True, 0 / -1 per sequence: it
        produced a k-mer / it did not (the reference maps a header to its taxon at the first k-mer only)."""
        min_length = self.max_k if min_length is None else min_length
        max_length = self.max_k if max_length is None else max_length
        if self.min_k == -1:
            self.min_k = min_length
        if self.min_k != min_length or self.max_k != max_length:
            raise InconsistentDataException(
                _lib.KMG_EINVAL, "New stream does not have the same min or max kmer length as a previous stream")
        if min_length != max_length and not self._min_set:
            self.ctx.check(self.ctx.lib.kmg_build_set_min_k(self.ctx.h, int(min_length)))
            self._min_set = True
        ridx = None
        if self.kind == PAYLOAD_READPOS or flags:
            ridx = np.empty(batch.nseq, np.int32)
        self.ctx.check(self.ctx.lib.kmg_build_add(self.ctx.h, _ptr(batch.bases), len(batch.bases), _ptr(batch.seq_off),
                                                  _ptr(batch.seq_id), _ptr(batch.seq_tok), batch.nseq, _ptr(ridx)))
        return ridx

    def add_packed(self, packed, seq_off, byte_off, seq_id, min_length=None):
        """addKmers for 2-bit packed sequences as MakeDatabase -p reads them (kmg_build_add_packed): `packed` holds
        Sequence.cBytes of every sequence back to back; the letters are restored on the device."""
        packed = np.ascontiguousarray(packed, np.uint8)
        seq_off = np.ascontiguousarray(seq_off, np.uint64)
        byte_off = np.ascontiguousarray(byte_off, np.uint64)
        seq_id = np.ascontiguousarray(seq_id, np.int32)
        min_length = self.max_k if min_length is None else min_length
        if self.min_k == -1:
            self.min_k = min_length
        if min_length != self.max_k and not self._min_set:
            self.ctx.check(self.ctx.lib.kmg_build_set_min_k(self.ctx.h, int(min_length)))
            self._min_set = True
        self.ctx.check(self.ctx.lib.kmg_build_add_packed(self.ctx.h, _ptr(packed), len(packed), _ptr(seq_off), _ptr(byte_off),
                                                         _ptr(seq_id), len(seq_id)))

    def create(self):
        """FileCreator.create's compute -> KmerDatabase (the caller writes it through the container)."""
        h = C.c_void_p()
        nw = C.c_uint64()
        self.ctx.check(self.ctx.lib.kmg_build_finish(self.ctx.h, C.byref(h), C.byref(nw)))
        self.n_windows = nw.value
        return KmerDatabase(self.ctx, h)


def merge_databases(dbs):
    """kmg_db_merge: DB.kmers over several files (DB.java:99-109) -- equal k-mers reduced with addAll."""
    ctx = dbs[0].ctx
    arr = (C.c_void_p * len(dbs))(*[d.h for d in dbs])
    h = C.c_void_p()
    ctx.check(ctx.lib.kmg_db_merge(ctx.h, arr, len(dbs), C.byref(h)))
    return KmerDatabase(ctx, h)


class DB:
    """Database/DB.java over device-resident reference databases.  A multi-file database (name.1 .. name.N,
    Matcher.java:80-89) is merged on the GPU once, at construction (DB.kmers, DB.java:99-109)."""

    def __init__(self, files):
        if not files:
            raise ValueError("Empty file list")          # DB.java:44-47
        f0 = files[0]
        for f in files:                                   # DB.java:36-42, 57-64
            if f.data_id() != f0.data_id():
                raise InconsistentDataException(_lib.KMG_EINVAL, "Files contain different datatypes")
            if (f.min_length, f.max_length, f.key_length) != (f0.min_length, f0.max_length, f0.key_length) or not f.rc:
                raise InconsistentDataException(
                    _lib.KMG_EINVAL, "Files contains different kmer parameters (min/max length, key length or "
                    "reverse complement included")
        self.files = list(files)
        self._merged = files[0] if len(files) == 1 else None
        self.ctx = f0.ctx
        self.min_length, self.max_length, self.key_length = f0.min_length, f0.max_length, f0.key_length
        self.max_key = 4 ** self.key_length

    @property
    def merged(self):
        """The files' k-mers as one database (merged on first use)."""
        if self._merged is None:
            self._merged = merge_databases(self.files)
        return self._merged

    def get_max_key(self):
        return self.max_key

    def get_nearest_kmers(self, search, max_diff=0, just=False):
        """Quick branch of getNearestKmers (DB.java:76-90): exact matches of a search database.
        -> hit_index[n_unique(search)] into this DB's k-mers (-1 = no match)."""
        if search.min_length < self.min_length or search.max_length > self.max_length:
            raise InconsistentDataException(
                _lib.KMG_EINVAL, "Search kmers contain kmers of a length inconsistent with the database")
        if max_diff != 0:
            raise KmersGpuError(_lib.KMG_EUNSUPPORTED, "inexact search (-n > 0) is not on the GPU path")
        hit = np.empty(search.n_unique, np.int64)
        nh = C.c_uint64()
        self.ctx.check(self.ctx.lib.kmg_match(self.ctx.h, search.h, self.merged.h, _ptr(hit), C.byref(nh)))
        return hit, nh.value


def match_into(ctx, search, ref, out):
    """kmg_match_dev: hit index of every k-mer of `search` in `ref` (-1 = miss) into the int64 torch tensor `out`
    on the context's GPU; returns once the result is there."""
    import torch
    if out.dtype != torch.int64 or out.numel() != search.n_unique or out.device != torch.device("cuda", ctx.device) \
            or not out.is_contiguous():
        raise ValueError("match_into wants a contiguous int64 tensor of search.n_unique elements on the context's GPU")
    if search.n_unique == 0:
        return out
    torch.cuda.synchronize(out.device)
    ctx.check(ctx.lib.kmg_match_dev(ctx.h, search.h, ref.h, out.data_ptr()))
    torch.cuda.ExternalStream(ctx.stream, device=out.device).synchronize()
    return out


def match_trie(search, ref, min_length=None, stored=False):
    """DB.getNearestKmers' non-quick branch without differences (DB.java:83,139-184; Root.java:29-170) for a search
    database of k-mers of several lengths (kmg_match_trie) -> (rows, hit): hit[i] = row of search k-mer i's answer in
    the taxon database `rows`, -1 = no line.  min_length = the search stream's declared shortest length (default: the
    search database's); stored = the handles were loaded from files (else counts are cut to the 255 a file holds
    before they add up, as in match_lines)."""
    ref = ref.merged if isinstance(ref, DB) else ref      # (a set whose files share keys keeps them: kmg_match_trie walks the files)
    ctx = search.ctx
    hit = np.empty(max(search.n_unique, 1), np.int64)
    nh = C.c_uint64()
    h = C.c_void_p()
    ctx.check(ctx.lib.kmg_match_trie(ctx.h, search.h, ref.h, int(search.min_length if min_length is None else min_length),
                                     0 if stored else 255, C.byref(h), _ptr(hit), C.byref(nh)))
    return KmerDatabase(ctx, h), hit[:search.n_unique]


def match_near(search, ref, max_diff):
    """Inexact search for k-mers of one length (kmg_match_near; DB.processNearestCommonKey, DB.java:139-184): every
    reference k-mer within Hamming distance max_diff of every search k-mer.  -> (pair_off[n_unique + 1], ref_index[],
    dist[]): the matches of search k-mer i are pair_off[i] .. pair_off[i + 1], ascending by reference k-mer."""
    ref = ref.merged if isinstance(ref, DB) else ref
    ctx = search.ctx
    off = np.zeros(search.n_unique + 1, np.uint64)
    n = C.c_uint64()
    ctx.check(ctx.lib.kmg_match_near(ctx.h, search.h, ref.h, int(max_diff), _ptr(off), None, None, 0, C.byref(n)))
    idx, dist = np.empty(max(n.value, 1), np.int64), np.empty(max(n.value, 1), np.uint8)
    if n.value:
        ctx.check(ctx.lib.kmg_match_near(ctx.h, search.h, ref.h, int(max_diff), None, _ptr(idx), _ptr(dist), n.value, C.byref(n)))
    return off, idx[:n.value], dist[:n.value]


def taxa_counts(db):
    """Utils/TaxaCounts.java default mode -> (taxon[], match[], unique[]) ascending by taxon."""
    f = db.merged if isinstance(db, DB) else db
    cap = min(max(f.n_values, 1), 1 << 16)
    while True:
        t = np.empty(cap, np.int32)
        m = np.empty(cap, np.int64)
        q = np.empty(cap, np.int64)
        n = C.c_uint64(cap)
        f.ctx.check(f.ctx.lib.kmg_taxacounts(f.ctx.h, f.h, _ptr(t), _ptr(m), _ptr(q), C.byref(n)))
        if n.value <= cap:
            return t[:n.value], m[:n.value], q[:n.value]
        cap = n.value                      # more taxa than rows offered: once more with room for all of them


def taxa_count_lines(db):
    """TaxaCounts.java:142-150: `taxon \\t match \\t unique` per taxon."""
    t, m, q = taxa_counts(db)
    return ["%d\t%d\t%d" % x for x in zip(t.tolist(), m.tolist(), q.tolist())]


def collect_by_read(search, ref):
    """Utils/CollectByRead.java over the join of a read database and a reference database (kmg_collect_by_read)
    -> (read[], pos[], ref_index[]) in ReadPos order."""
    ref = ref.merged if isinstance(ref, DB) else ref
    ctx = search.ctx
    cap = max(search.n_values, 1)
    read, pos, idx = np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.int64)
    n = C.c_uint64()
    ctx.check(ctx.lib.kmg_collect_by_read(ctx.h, search.h, ref.h, C.byref(n), _ptr(read), _ptr(pos), _ptr(idx)))
    return read[:n.value], pos[:n.value], idx[:n.value]


def collect_by_read_blocks(search, ref):
    """The blocks of CollectByRead's output file: {(read / 1000) * 1000: lines `r:p \\t 0:TxC|TxC` joined by
    "\\n"} (CollectByRead.java:105-124,159-184); counts capped at 255 as on the reference's way through its
    temp file; of two records with one (read, pos) the first is kept (TreeSet.add)."""
    ref = ref.merged if isinstance(ref, DB) else ref
    read, pos, idx = collect_by_read(search, ref)
    r = ref.fetch()
    out, lines, cur, last = {}, [], None, None
    for rd, ps, h in zip(read.tolist(), pos.tolist(), idx.tolist()):
        if (rd, ps) == last:
            continue
        last = (rd, ps)
        key = (rd // 1000) * 1000
        if key != cur:
            if cur is not None:
                out[cur] = "\n".join(lines).encode()
            cur, lines = key, []
        c, d = int(r["val_off"][h]), int(r["val_off"][h + 1])
        lines.append("%d:%d\t0:%s" % (rd, ps, "|".join("%dx%d" % (x, min(y, 255)) for x, y in
                                                         zip(r["val_a"][c:d].tolist(), r["val_b"][c:d].tolist()))))
    if cur is not None:
        out[cur] = "\n".join(lines).encode()
    return out


def classify_reads(search, ref, n_reads, with_hits=False):
    """Utils/ReadClassifier.java processRead for every read (kmg_classify_reads) -> call[n_reads], -1 = no call.
    with_hits: -> (call, has_hit) with has_hit[r] = 1 if read r has a matched k-mer (kmg_classify_reads_hits)."""
    ref = ref.merged if isinstance(ref, DB) else ref
    ctx = search.ctx
    call = np.empty(max(n_reads, 1), np.int32)
    if not with_hits:
        ctx.check(ctx.lib.kmg_classify_reads(ctx.h, search.h, ref.h, n_reads, _ptr(call)))
        return call[:n_reads]
    seen = np.zeros(max(n_reads, 1), np.uint8)
    ctx.check(ctx.lib.kmg_classify_reads_hits(ctx.h, search.h, ref.h, n_reads, _ptr(call), _ptr(seen)))
    return call[:n_reads], seen[:n_reads]


def classify_reads_rows(search, rows, hit, n_reads):
    """The same calls from answers found by another search (kmg_classify_reads_rows): `rows`, `hit` as match_trie
    hands them out -- Matcher's search over several lengths.  -> (call, has_hit)"""
    ctx = search.ctx
    call = np.empty(max(n_reads, 1), np.int32)
    seen = np.zeros(max(n_reads, 1), np.uint8)
    hit = np.ascontiguousarray(hit, np.int64)
    ctx.check(ctx.lib.kmg_classify_reads_rows(ctx.h, search.h, rows.h, _ptr(hit) if hit.size else None, n_reads, _ptr(call), _ptr(seen)))
    return call[:n_reads], seen[:n_reads]


def read_call_blocks(call, has_hit):
    """The blocks of ReadClassifier's output file from per-read calls: one block per (read / 1000) * 1000 that
    holds a read with a matched k-mer (CollectByRead.java:105-124 writes no other), lines `read \t call \n` for
    the reads that have a call (ReadClassifier.java:140-160)."""
    call, has_hit = np.asarray(call), np.asarray(has_hit)
    out = {}
    for key in np.unique((np.nonzero(has_hit)[0] // 1000) * 1000).tolist():
        seg = call[key:key + 1000]
        out[key] = "".join("%d\t%d\n" % (key + i, c) for i, c in enumerate(seg.tolist()) if c >= 0).encode()
    return out


_BASES = np.frombuffer(b"ACTG", dtype=np.uint8)


def kmer_strings(keys, k, keys_lo=None):
    """Sequence.toString (Sequence.java:117-134) for an array of left-aligned 2-bit k-mers (keys_lo: bases 32 .. of
    k-mers of more than 32 bases)."""
    if k > 32:
        return [a + b for a, b in zip(kmer_strings(keys, 32), kmer_strings(keys_lo, k - 32))]
    keys = np.asarray(keys, np.uint64)
    shifts = (np.uint64(62) - np.uint64(2) * np.arange(k, dtype=np.uint64))
    codes = ((keys[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.intp)
    return [row.tobytes().decode() for row in _BASES[codes]]


def match_lines(search, ref, hit, stored=False, ref_stored=None):
    """One line per matched search k-mer, ascending (Matcher.java:247-249): KMER \\t r:p|r:p \\t -|TxC|TxC
    (ResultsDataType.java:15-35).  Matcher reads its databases from files, whose counts are bytes saturated at 255
    (CountDataType.java:40): for handles straight from a build the counts are capped here to what the file would
    hold; stored=True says the handles were loaded from files (a set of several files adds the stored bytes as
    longs, DB.java:99-109, and a line can show more than 255)."""
    s, r = search.fetch(), ref.fetch()
    cap = (1 << 31) if stored else 255
    rcap = cap if ref_stored is None else ((1 << 31) if ref_stored else 255)     # (match_trie's rows hold sums: ref_stored=True)
    idx = np.nonzero(hit >= 0)[0]
    names = kmer_strings(s["keys"][idx], search.k, s["keys_lo"][idx] if search.k > 32 else None)
    if search.min_length != search.k:                    # k-mers of several lengths: zero-padded behind their own
        names = [n[:l] for n, l in zip(names, search.lens()[idx].tolist())]
    lines = []
    for name, i in zip(names, idx.tolist()):
        a, b = int(s["val_off"][i]), int(s["val_off"][i + 1])
        if search.kind == PAYLOAD_READPOS:
            left = "|".join("%d:%d" % (x, y) for x, y in zip(s["val_a"][a:b].tolist(), s["val_b"][a:b].tolist()))
        else:
            left = "|".join("%dx%d" % (x, min(y, cap)) for x, y in zip(s["val_a"][a:b].tolist(), s["val_b"][a:b].tolist()))
        h = int(hit[i])
        c, d = int(r["val_off"][h]), int(r["val_off"][h + 1])
        right = "|".join("%dx%d" % (x, min(y, rcap)) for x, y in zip(r["val_a"][c:d].tolist(), r["val_b"][c:d].tolist()))
        lines.append("%s\t%s\t-|%s" % (name, left, right))
    return lines


def _revcomp_key(key, k, width=64):
    """Kmer.getRC on a left-aligned 2-bit k-mer (in `width` bits): bases reversed, each complemented (code ^ 2)."""
    v, out = int(key) >> (width - 2 * k), 0
    for _ in range(k):
        out = (out << 2) | ((v & 3) ^ 2)
        v >>= 2
    return out << (width - 2 * k)


def near_match_lines(search, ref, max_diff, just=False, stored=False):
    """Matcher -n max_diff [-j]: the lines of _near_match_rows (k-mers of one length), or, for k-mers of several lengths,
    of near_match_lines_var."""
    one = ref.merged if isinstance(ref, DB) else ref
    if search.min_length != search.k or one.min_length != one.k:
        return near_match_lines_var(search, ref, max_diff, just, stored=stored)
    return _near_match_rows(search, ref, max_diff, just, stored)[1]


def near_match_lines_var(search, ref, max_diff, just=False, min_length=None, stored=False):
    """Matcher -n max_diff [-j] over k-mers of several lengths (DB.processNearestCommonKey, DB.java:139-184, over the trie,
    Root.java:29-170).  A search k-mer of T bases walks the trie to depth T, so what it can meet are the DISTINCT T-base
    prefixes p of the stored k-mers of at least T bases, each within max_diff differences of it, and what it is handed
    for p is the data object of the node covering depth T on p's path -- the closed form of kmg_match_trie,
    asked for p itself.  Per length T: the prefixes = the database restricted to T (kmg_db_restrict), their answers =
    kmg_match_trie with the prefixes as the search side, the neighbours = kmg_match_near of the search k-mers of that
    length against the prefixes; the per-k-mer rules (-j, reverse complements, order, diff text) are _near_match_rows'.
    Lines come out in the search stream's order."""
    one = ref.merged if isinstance(ref, DB) else ref
    ctx = search.ctx
    shortest = int(search.min_length if min_length is None else min_length)
    if shortest < one.min_length or search.k > one.k:                                # DB.java:78-81
        raise InconsistentDataException(_lib.KMG_EINVAL, "Search kmers contain kmers of a length inconsistent with the database")
    s = search.fetch()
    lens = search.lens() if search.min_length != search.k else np.full(search.n_unique, search.k, np.uint8)
    out = []
    for T in sorted(set(lens.tolist())):
        sel = np.nonzero(lens == T)[0]
        cnt = (s["val_off"][sel + 1] - s["val_off"][sel]).astype(np.int64)
        voff = np.zeros(len(sel) + 1, np.uint64)
        voff[1:] = np.cumsum(cnt)
        take = np.concatenate([np.arange(int(a), int(b)) for a, b in zip(s["val_off"][sel], s["val_off"][sel + 1])]) if len(sel) else np.zeros(0, np.int64)
        sub = KmerDatabase.load(ctx, T, search.key_length, search.rc, search.kind, s["keys"][sel], voff,
                                np.ascontiguousarray(s["val_a"][take]), np.ascontiguousarray(s["val_b"][take]),
                                keys_lo=s["keys_lo"][sel] if T > 32 else None)
        prefixes = one.restrict(T) if (one.min_length != one.k or one.k != T) else one
        rows, row_of = match_trie(prefixes, one, shortest, stored=stored)
        idx, lines = _near_match_rows(sub, prefixes, max_diff, just, stored, data_rows=rows, row_of=row_of)
        out.extend(zip(sel[idx].tolist(), lines))
        rows.close()
        sub.close()
        if prefixes is not one:
            prefixes.close()
    out.sort(key=lambda e: e[0])
    return [l for _, l in out]


def _near_match_rows(search, ref, max_diff, just=False, stored=False, data_rows=None, row_of=None):
    """-> (indices of the search k-mers with a line, the lines).  data_rows / row_of: the values printed for reference
    k-mer h are row row_of[h] of the database data_rows (sums: printed as they are) instead of h's own.
    Matcher -n max_diff [-j] for k-mers of one length: one line per search k-mer that has a match within max_diff
    differences, `KMER \\t values \\t DIFF|TxC|TxC DIFF|TxC ...` (ResultsDataType.java:15-35).  The GPU hands over
    every reference k-mer within reach in ascending order (kmg_match_near); here, per search k-mer and on a handful
    of matches: -j keeps the best distance only (Root.java:144-148, DB.java:164-171), of a k-mer and its reverse
    complement the one met first stays unless the later one is closer (ClosestInfoCollector.add, :30-50), entries
    are ordered by number of differences, then (position, base) (KmerDiffComparaotr, :119-147), and a difference
    prints as position + new base, no difference as "-" (KmerDiff.java:54-67)."""
    ref = ref.merged if isinstance(ref, DB) else ref
    cap = (1 << 31) if stored else 255            # see match_lines
    off, idx, dist = match_near(search, ref, max_diff)
    s, r = search.fetch(), ref.fetch()
    vals, rcap = (r, cap) if data_rows is None else (data_rows.fetch(), 1 << 31)
    k = search.k
    wide = k > 32                                 # two words a k-mer: one 128-bit integer here
    width = 128 if wide else 64
    whole = (lambda f, j: (int(f["keys"][j]) << 64) | int(f["keys_lo"][j])) if wide else (lambda f, j: int(f["keys"][j]))
    rows = np.nonzero(off[1:] > off[:-1])[0]
    names = kmer_strings(s["keys"][rows], k, s["keys_lo"][rows] if wide else None)
    letters = "ACTG"
    lines = []
    for name, i in zip(names, rows.tolist()):
        lo, hi = int(off[i]), int(off[i + 1])
        cand = list(zip(idx[lo:hi].tolist(), dist[lo:hi].tolist()))
        if just:
            best = min(d for _, d in cand)
            cand = [c for c in cand if c[1] == best]
        kept = {}                                           # reference k-mer (key) -> (hit, dist), in meeting order
        for h, d in cand:
            key = whole(r, h)
            rc = _revcomp_key(key, k, width)
            if rc in kept:
                if d < kept[rc][1]:
                    del kept[rc]
                    kept[key] = (h, d)
            else:
                kept[key] = (h, d)
        skey = whole(s, i)
        entries = []
        for key, (h, d) in kept.items():
            x = (skey ^ key) >> (width - 2 * k)
            diffs = [(p, (key >> (width - 2 - 2 * p)) & 3) for p in range(k) if (x >> (2 * (k - 1 - p))) & 3]
            entries.append(((len(diffs), diffs), h))
        entries.sort(key=lambda e: e[0])
        a0, a1 = int(s["val_off"][i]), int(s["val_off"][i + 1])
        if search.kind == PAYLOAD_READPOS:
            left = "|".join("%d:%d" % (x, y) for x, y in zip(s["val_a"][a0:a1].tolist(), s["val_b"][a0:a1].tolist()))
        else:
            left = "|".join("%dx%d" % (x, min(y, cap)) for x, y in zip(s["val_a"][a0:a1].tolist(), s["val_b"][a0:a1].tolist()))
        parts = []
        for (_, diffs), h in entries:
            h = h if row_of is None else int(row_of[h])
            c0, c1 = int(vals["val_off"][h]), int(vals["val_off"][h + 1])
            text = "".join("%d%s" % (p, letters[b]) for p, b in diffs) if diffs else "-"
            parts.append("%s|%s" % (text, "|".join("%dx%d" % (x, min(y, rcap)) for x, y in
                                                      zip(vals["val_a"][c0:c1].tolist(), vals["val_b"][c0:c1].tolist()))))
        lines.append("%s\t%s\t%s" % (name, left, " ".join(parts)))
    return rows, lines
