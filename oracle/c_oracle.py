"""ctypes wrapper around oracle/kmers_oracle.c (TEST INFRASTRUCTURE ONLY -- see that file's header)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libkmers_oracle.so")
_SO_WIDE = os.path.join(_HERE, "_build", "libkmers_oracle_wide.so")      # the same source with -DKO_WIDE: k <= 64


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load(so=_SO):
    if not os.path.exists(so):
        build()
    lib = C.CDLL(so)
    u64, vp = C.c_uint64, C.c_void_p
    lib.ko_build.restype = vp
    lib.ko_build.argtypes = [C.c_char_p, u64, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.ko_free.argtypes = [vp]
    lib.ko_set_filters.argtypes = [C.c_int, C.c_int]
    lib.ko_set_filters.restype = None
    lib.ko_set_key_range.argtypes = [C.c_uint32, C.c_uint32]
    lib.ko_set_key_range.restype = None
    lib.ko_set_min_k.argtypes = [C.c_int]
    lib.ko_set_min_k.restype = None
    lib.ko_min_k.argtypes = [vp]
    lib.ko_lens.restype, lib.ko_lens.argtypes = vp, [vp]
    lib.ko_restrict.restype, lib.ko_restrict.argtypes = vp, [vp, C.c_int]
    for n in ("U", "V", "n_windows", "n_records", "n_reads"):
        f = getattr(lib, "ko_" + n)
        f.restype, f.argtypes = u64, [vp]
    for n in ("keys", "keys_lo", "val_off", "val_a", "val_b", "bucket_off", "name_off", "names"):
        f = getattr(lib, "ko_" + n)
        f.restype, f.argtypes = vp, [vp]
    lib.ko_encode_blocks.restype = u64
    lib.ko_encode_blocks.argtypes = [vp, vp, u64, vp]
    lib.ko_match.restype = u64
    lib.ko_match.argtypes = [vp, vp, vp, C.c_int]
    lib.ko_match_lines.restype = u64
    lib.ko_match_lines.argtypes = [vp, vp, vp, vp, u64]
    lib.ko_taxacounts.restype = u64
    lib.ko_taxacounts.argtypes = [vp, vp, vp, vp, u64]
    return lib


_lib = None
_lib_wide = None


def lib(wide=False):
    """The restatement for k <= 32, or (wide) its two-word build for k <= 64."""
    global _lib, _lib_wide
    if wide:
        if _lib_wide is None:
            _lib_wide = _load(_SO_WIDE)
        return _lib_wide
    if _lib is None:
        _lib = _load()
    return _lib


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class OracleDB:
    """Result of MakeDatabase (-a: mode 0, -q: mode 1) as CSR arrays."""

    def __init__(self, data: bytes, mode: int, k: int, key_len: int, threads: int = 1, dust: int = -1,
                 runofsame: int = -1, key_range=None, min_k=None, _handle=None, wide=None):
        """k = maxK (MakeDatabase -K), min_k = minK (-k; default: the same).
        key_range = (first, end): only the slice of the database in those key buckets (ko_set_key_range).
        wide: the two-word build of the restatement (default: for k > 32)."""
        self.wide = (k > 32) if wide is None else bool(wide)
        L = lib(self.wide)
        if _handle is not None:
            self._h = _handle
        else:
            L.ko_set_filters(dust, runofsame)          # MakeDatabase -D / -R
            L.ko_set_min_k(0 if min_k is None else int(min_k))
            if key_range is not None:
                L.ko_set_key_range(int(key_range[0]), int(key_range[1]))
            try:
                self._h = L.ko_build(data, len(data), mode, k, key_len, threads)
            finally:
                L.ko_set_filters(-1, -1)
                L.ko_set_key_range(0, 0)
                L.ko_set_min_k(0)
        if not self._h:
            raise ValueError("ko_build rejected the parameters")
        h = self._h
        self.k, self.key_len, self.kind = k, key_len, mode
        self.min_k = L.ko_min_k(h)
        self.U, self.V = L.ko_U(h), L.ko_V(h)
        self.lens = _arr(L.ko_lens(h), self.U, np.uint8)
        self.n_windows, self.n_records = L.ko_n_windows(h), L.ko_n_records(h)
        self.keys = _arr(L.ko_keys(h), self.U, np.uint64)
        self.keys_lo = _arr(L.ko_keys_lo(h), self.U, np.uint64) if self.wide else np.zeros(self.U, np.uint64)
        self.val_off = _arr(L.ko_val_off(h), self.U + 1, np.uint64)
        self.val_a = _arr(L.ko_val_a(h), self.V, np.int32)
        self.val_b = _arr(L.ko_val_b(h), self.V, np.int32)
        self.bucket_off = _arr(L.ko_bucket_off(h), 4 ** key_len + 1, np.uint64)
        self.names = []
        if mode == 1 and _handle is None:
            nr = L.ko_n_reads(h)
            off = _arr(L.ko_name_off(h), nr + 1, np.uint64)
            raw = _arr(L.ko_names(h), int(off[-1]) if nr else 0, np.uint8).tobytes()
            self.names = [raw[int(off[i]):int(off[i + 1])].decode("latin-1") for i in range(nr)]

    def close(self):
        if self._h:
            lib(self.wide).ko_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def restrict(self, length):
        """KmerUtils.restrictedStream(.., length, length, ..) over the whole database (ko_restrict)."""
        h = lib(self.wide).ko_restrict(self._h, int(length))
        if not h:
            raise ValueError("length outside the database's k-mer lengths")
        out = OracleDB(b"", self.kind, int(length), self.key_len, _handle=h, wide=self.wide)
        if self.wide and int(length) <= 32:
            out.keys_lo = np.zeros(out.U, np.uint64)
        return out

    def encode_blocks(self):
        """-> (bytes of all 4^l blocks back to back, block_off[4^l+1])."""
        L = lib(self.wide)
        nk = 4 ** self.key_len
        off = np.zeros(nk + 1, dtype=np.uint64)
        need = L.ko_encode_blocks(self._h, None, 0, off.ctypes.data)
        out = np.zeros(max(need, 1), dtype=np.uint8)
        L.ko_encode_blocks(self._h, out.ctypes.data, need, off.ctypes.data)
        return out[:need].tobytes(), off

    def match(self, ref: "OracleDB", threads: int = 1):
        hit = np.empty(self.U, dtype=np.int64)
        assert self.wide == ref.wide
        n = lib(self.wide).ko_match(self._h, ref._h, hit.ctypes.data, threads)
        return hit, n

    def match_lines(self, ref: "OracleDB", hit):
        L = lib(self.wide)
        hit = np.ascontiguousarray(hit, dtype=np.int64)
        need = L.ko_match_lines(self._h, ref._h, hit.ctypes.data, None, 0)
        out = np.zeros(max(need, 1), dtype=np.uint8)
        L.ko_match_lines(self._h, ref._h, hit.ctypes.data, out.ctypes.data, need)
        return out[:need].tobytes()

    def taxacounts(self):
        L = lib(self.wide)
        cap = max(self.V, 1)
        t = np.zeros(cap, np.int32)
        m = np.zeros(cap, np.int64)
        q = np.zeros(cap, np.int64)
        n = L.ko_taxacounts(self._h, t.ctypes.data, m.ctypes.data, q.ctypes.data, cap)
        return t[:n], m[:n], q[:n]
