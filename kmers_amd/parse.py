"""Host-side FASTA/FASTQ record splitting (libkmers_host.so, csrc/host_parse.cpp): the ID/KMER/OTHER
state machine of OtherFiles/KmersFromFileStateChanger.java:46-68 -> SequenceBatch for FileCreator."""
import ctypes as C
import gzip
import os

import numpy as np

from .api import SequenceBatch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkmers_host.so")
_lib = None

FASTA, FASTQ = 0, 1


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: run __graft_entry__.build()" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        vp, u64 = C.c_void_p, C.c_uint64
        lib.kmh_parse.restype = vp
        lib.kmh_parse.argtypes = [C.c_char_p, u64, C.c_int, C.c_int]
        lib.kmh_free.argtypes = [vp]
        lib.kmh_free.restype = None
        lib.kmh_stream_open.restype = vp
        lib.kmh_stream_open.argtypes = [C.c_int]
        lib.kmh_stream_close.restype = None
        lib.kmh_stream_close.argtypes = [vp]
        lib.kmh_stream_next.restype = vp
        lib.kmh_stream_next.argtypes = [vp, C.c_char_p, u64, C.c_int, C.c_int]
        lib.kmh_stream_pending.restype = u64
        lib.kmh_stream_pending.argtypes = [vp]
        lib.kmh_apply_id_map.restype = u64
        lib.kmh_apply_id_map.argtypes = [vp, C.c_char_p, u64]
        for n in ("nseq", "nbases", "ntok"):
            f = getattr(lib, "kmh_" + n)
            f.restype, f.argtypes = u64, [vp]
        for n in ("bases", "seq_off", "seq_id", "id_ok", "seq_tok", "name_off", "names"):
            f = getattr(lib, "kmh_" + n)
            f.restype, f.argtypes = vp, [vp]
        lib.kmh_eof_in_kmer.restype = C.c_int
        lib.kmh_eof_in_kmer.argtypes = [vp]
        lib.kmh_high_bytes.restype = C.c_int
        lib.kmh_high_bytes.argtypes = [vp]
        lib.kmh_empty_header.restype = C.c_int
        lib.kmh_empty_header.argtypes = [vp]
        lib.kmh_decode_blocks.restype = vp
        lib.kmh_decode_blocks.argtypes = [C.c_char_p, u64, C.c_int]
        lib.kmh_decoded_free.argtypes = [vp]
        lib.kmh_decoded_free.restype = None
        for n in ("nkeys", "nvals"):
            f = getattr(lib, "kmh_decoded_" + n)
            f.restype, f.argtypes = u64, [vp]
        for n in ("keys", "keys_lo", "val_off", "val_a", "val_b", "lens"):
            f = getattr(lib, "kmh_decoded_" + n)
            f.restype, f.argtypes = vp, [vp]
        _lib = lib
    return _lib


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


def maybe_gunzip(data: bytes) -> bytes:
    """jar Zip.ZipOrNot.isGZipped: first two bytes 1f 8b."""
    return gzip.decompress(data) if data[:2] == b"\x1f\x8b" else data


class ParsedFile:
    def __init__(self, batch, names, id_ok, eof_in_kmer, high_bytes=0, empty_header=0, named=False):
        self.batch = batch
        self.named = named            # FASTA: headers were looked up in an id map (-m), not read as ints
        self.names = names            # FASTQ: header string of every token
        self.id_ok = id_ok            # FASTA: header parsed as a Java int
        self.eof_in_kmer = eof_in_kmer
        self.empty_header = empty_header   # FASTQ: an empty header line (ReadIDMapping.java:10-16 starts with "")
        self.high_bytes = high_bytes  # a sequence line held a byte >= 128 (Sequence.java:15-33 reads those differently)


def parse(data: bytes, mode: int, k: int, id_map: bytes = None) -> ParsedFile:
    """id_map (FASTA only): the text of MakeDatabase's -m file, "<sequence id>\\t<taxon id>" per line
    (MakeDatabase.java:138-147); headers are looked up in it instead of being read as ints."""
    lib = _load()
    data = maybe_gunzip(data)
    h = lib.kmh_parse(data, len(data), 2 if (mode == FASTA and id_map is not None) else mode, k)
    try:
        if mode == FASTA and id_map is not None:
            id_map = maybe_gunzip(id_map)
            bad = lib.kmh_apply_id_map(h, id_map, len(id_map))
            if bad:
                raise ValueError("line %d of the id map is not <sequence id><tab><int>" % bad)
        nseq, nb = lib.kmh_nseq(h), lib.kmh_nbases(h)
        bases = _arr(lib.kmh_bases(h), nb, np.uint8)
        off = _arr(lib.kmh_seq_off(h), nseq + 1, np.uint64)
        if mode == FASTA:
            ids = _arr(lib.kmh_seq_id(h), nseq, np.int32)
            ok = _arr(lib.kmh_id_ok(h), nseq, np.uint8)
            return ParsedFile(SequenceBatch(bases, off, seq_id=ids), None, ok, lib.kmh_eof_in_kmer(h), lib.kmh_high_bytes(h),
                              named=id_map is not None)
        tok = _arr(lib.kmh_seq_tok(h), nseq, np.int64)
        nt = lib.kmh_ntok(h)
        noff = _arr(lib.kmh_name_off(h), nt + 1, np.uint64)
        raw = _arr(lib.kmh_names(h), int(noff[-1]) if nt else 0, np.uint8).tobytes()
        names = [raw[int(noff[i]):int(noff[i + 1])].decode("latin-1") for i in range(nt)]
        return ParsedFile(SequenceBatch(bases, off, seq_tok=tok), names, None, lib.kmh_eof_in_kmer(h), lib.kmh_high_bytes(h),
                          lib.kmh_empty_header(h))
    finally:
        lib.kmh_free(h)


def parse_pieces(pieces, mode: int, k: int, id_map: bytes = None):
    """The input handed over in pieces (kmh_stream_*): yields one ParsedFile per piece, whose sequences, one piece after
    the other, are those of parse() on the whole input.  FASTQ / named FASTA: .names holds the header of every SEQUENCE of
    the piece, and tokens are equal across pieces iff the headers are."""
    lib = _load()
    named = mode == FASTA and id_map is not None
    st = lib.kmh_stream_open(2 if named else mode)
    try:
        pieces = list(pieces)
        for i, data in enumerate(pieces):
            h = lib.kmh_stream_next(st, data, len(data), 1 if i == len(pieces) - 1 else 0, k)
            try:
                if named:
                    bad = lib.kmh_apply_id_map(h, id_map, len(id_map))
                    if bad:
                        raise ValueError("line %d of the id map is not <sequence id><tab><int>" % bad)
                nseq, nb = lib.kmh_nseq(h), lib.kmh_nbases(h)
                bases = _arr(lib.kmh_bases(h), nb, np.uint8)
                off = _arr(lib.kmh_seq_off(h), nseq + 1, np.uint64)
                flags = (lib.kmh_eof_in_kmer(h), lib.kmh_high_bytes(h))
                if mode == FASTA:
                    ids = _arr(lib.kmh_seq_id(h), nseq, np.int32)
                    ok = _arr(lib.kmh_id_ok(h), nseq, np.uint8)
                    yield ParsedFile(SequenceBatch(bases, off, seq_id=ids), None, ok, *flags, named=named)
                else:
                    tok = _arr(lib.kmh_seq_tok(h), nseq, np.int64)
                    noff = _arr(lib.kmh_name_off(h), nseq + 1, np.uint64)
                    raw = _arr(lib.kmh_names(h), int(noff[-1]) if nseq else 0, np.uint8).tobytes()
                    names = [raw[int(noff[j]):int(noff[j + 1])].decode("latin-1") for j in range(nseq)]
                    yield ParsedFile(SequenceBatch(bases, off, seq_tok=tok), names, None, *flags, lib.kmh_empty_header(h))
            finally:
                lib.kmh_free(h)
    finally:
        lib.kmh_stream_close(st)


def decode_blocks(data: bytes, kind: int, with_lens=False, with_lo=False):
    """KmerFile's prefix-delta decode (KmerFile.java:200-250) of all key blocks back to back
    -> (keys u64, val_off u64, val_a i32, val_b i32[, lens u8][, keys_lo u64]).  kind 0 = reference (counts), 1 = reads.
    keys_lo: bases 32 .. of k-mers of more than 32 bases."""
    lib = _load()
    h = lib.kmh_decode_blocks(data, len(data), kind)
    if not h:
        raise ValueError("malformed k-mer blocks, or k-mers longer than 64 bases (not supported on the GPU path)")
    try:
        nk, nv = lib.kmh_decoded_nkeys(h), lib.kmh_decoded_nvals(h)
        out = (_arr(lib.kmh_decoded_keys(h), nk, np.uint64), _arr(lib.kmh_decoded_val_off(h), nk + 1, np.uint64),
               _arr(lib.kmh_decoded_val_a(h), nv, np.int32), _arr(lib.kmh_decoded_val_b(h), nv, np.int32))
        if with_lens:
            out = out + (_arr(lib.kmh_decoded_lens(h), nk, np.uint8),)
        if with_lo:
            out = out + (_arr(lib.kmh_decoded_keys_lo(h), nk, np.uint64),)
        return out
    finally:
        lib.kmh_decoded_free(h)
