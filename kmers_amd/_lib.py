"""ctypes binding of libkmers_gpu.so (include/kmers_gpu.h).  No CPU fallback: if the HIP library is
missing or no GPU is visible the product path fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KMERS_GPU_LIB: another build of the same library (A/B timing of kernel variants on one box)
LIB_PATH = os.environ.get("KMERS_GPU_LIB") or os.path.join(_HERE, "libkmers_gpu.so")

KMG_OK = 0
KMG_EINVAL = -1
KMG_ENOMEM = -2
KMG_EDEVICE = -3
KMG_EUNSUPPORTED = -4
KMG_ESTATE = -5
PAYLOAD_TAXON = 0
PAYLOAD_READPOS = 1

# every symbol include/kmers_gpu.h declares
SYMBOLS = [
    "kmg_create", "kmg_destroy", "kmg_last_error", "kmg_stream",
    "kmg_build_begin", "kmg_build_set_filters", "kmg_build_set_min_k", "kmg_build_set_key_range", "kmg_build_add", "kmg_build_add_dev", "kmg_build_finish",
    "kmg_db_info", "kmg_db_fetch", "kmg_db_load", "kmg_db_free", "kmg_db_encode", "kmg_db_merge", "kmg_db_copy_dev", "kmg_db_load_dev", "kmg_db_payload_bits", "kmg_build_add_packed", "kmg_db_min_k", "kmg_db_fetch_lens", "kmg_db_load_var", "kmg_db_restrict", "kmg_db_restrict_range", "kmg_match_trie", "kmg_db_fetch_keys_lo", "kmg_db_load_wide",
    "kmg_match", "kmg_match_dev", "kmg_match_keys_dev", "kmg_match_keys_wide_dev", "kmg_match_dev_noalloc", "kmg_match_near", "kmg_taxacounts", "kmg_collect_by_read", "kmg_classify_reads", "kmg_classify_reads_hits", "kmg_classify_reads_rows",
    "kmg_comm_unique_id", "kmg_comm_create", "kmg_comm_destroy", "kmg_comm_info", "kmg_match_sharded", "kmg_db_exchange_by_key_range",
    "kmg_prof_enable", "kmg_prof_reset", "kmg_prof_get", "kmg_prof_count", "kmg_prof_name", "kmg_calib_copy",
]


class KmersGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("kmers_gpu error %d: %s" % (code, msg))
        self.code = code


class InconsistentDataException(KmersGpuError):
    """Exceptions/InconsistentDataException.java -- parameter mismatches (KMG_EINVAL)."""


_lib = None


def load():
    """Load the HIP library.  Raises if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    P = C.POINTER
    lib.kmg_create.argtypes = [i32, P(vp)]
    lib.kmg_destroy.argtypes = [vp]
    lib.kmg_destroy.restype = None
    lib.kmg_last_error.argtypes = [vp]
    lib.kmg_last_error.restype = C.c_char_p
    lib.kmg_stream.argtypes = [vp]
    lib.kmg_stream.restype = vp
    lib.kmg_build_begin.argtypes = [vp, i32, i32, i32, i32]
    lib.kmg_build_set_filters.argtypes = [vp, i32, i32]
    lib.kmg_build_set_min_k.argtypes = [vp, i32]
    lib.kmg_db_min_k.argtypes = [vp]
    lib.kmg_db_fetch_lens.argtypes = [vp, vp]
    lib.kmg_db_load_var.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, u64, vp, vp, vp, P(vp)]
    lib.kmg_db_fetch_keys_lo.argtypes = [vp, vp]
    lib.kmg_db_load_wide.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp, u64, vp, vp, vp, P(vp)]
    lib.kmg_db_restrict.argtypes = [vp, vp, i32, P(vp)]
    lib.kmg_db_restrict_range.argtypes = [vp, vp, i32, i32, P(vp)]
    lib.kmg_match_trie.argtypes = [vp, vp, vp, i32, i32, P(vp), vp, P(u64)]
    lib.kmg_build_set_key_range.argtypes = [vp, C.c_uint32, C.c_uint32]
    lib.kmg_build_add.argtypes = [vp, vp, u64, vp, vp, vp, u64, vp]
    lib.kmg_build_add_dev.argtypes = [vp, vp, u64, vp, vp, vp, u64, vp]
    lib.kmg_build_add_packed.argtypes = [vp, vp, u64, vp, vp, vp, u64]
    lib.kmg_build_finish.argtypes = [vp, P(vp), P(u64)]
    lib.kmg_db_info.argtypes = [vp, P(i32), P(i32), P(i32), P(i32), P(u64), P(u64)]
    lib.kmg_db_fetch.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.kmg_db_load.argtypes = [vp, i32, i32, i32, i32, vp, u64, vp, vp, vp, P(vp)]
    lib.kmg_db_free.argtypes = [vp]
    lib.kmg_db_free.restype = None
    lib.kmg_db_encode.argtypes = [vp, vp, vp, u64, P(u64), vp]
    lib.kmg_db_merge.argtypes = [vp, P(vp), i32, P(vp)]
    lib.kmg_db_payload_bits.argtypes = [vp]
    lib.kmg_db_copy_dev.argtypes = [vp, i32, vp, u64, u64]
    lib.kmg_db_load_dev.argtypes = [vp, i32, i32, i32, i32, vp, u64, vp, vp, vp, u64, P(vp)]
    lib.kmg_match.argtypes = [vp, vp, vp, vp, P(u64)]
    lib.kmg_match_near.argtypes = [vp, vp, vp, i32, vp, vp, vp, u64, P(u64)]
    lib.kmg_match_dev.argtypes = [vp, vp, vp, vp]
    lib.kmg_match_keys_dev.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.kmg_match_keys_wide_dev.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp]
    lib.kmg_match_dev_noalloc.argtypes = [vp, vp, vp, vp]
    lib.kmg_collect_by_read.argtypes = [vp, vp, vp, P(u64), vp, vp, vp]
    lib.kmg_classify_reads.argtypes = [vp, vp, vp, u64, vp]
    lib.kmg_classify_reads_hits.argtypes = [vp, vp, vp, u64, vp, vp]
    lib.kmg_classify_reads_rows.argtypes = [vp, vp, vp, vp, u64, vp, vp]
    lib.kmg_comm_unique_id.argtypes = [vp]
    lib.kmg_comm_create.argtypes = [vp, i32, i32, vp, P(vp)]
    lib.kmg_comm_destroy.argtypes = [vp]
    lib.kmg_comm_destroy.restype = None
    lib.kmg_comm_info.argtypes = [vp, P(i32), P(i32)]
    lib.kmg_match_sharded.argtypes = [vp, vp, vp, vp, vp]
    lib.kmg_db_exchange_by_key_range.argtypes = [vp, vp, vp, P(vp)]
    lib.kmg_taxacounts.argtypes = [vp, vp, vp, vp, vp, P(u64)]
    lib.kmg_prof_enable.argtypes = [vp, i32]
    lib.kmg_prof_reset.argtypes = [vp]
    lib.kmg_prof_get.argtypes = [vp, i32, P(C.c_double), P(u64), P(u64)]
    lib.kmg_calib_copy.argtypes = [vp, u64]
    lib.kmg_prof_count.argtypes = []
    lib.kmg_prof_name.argtypes = [i32]
    lib.kmg_prof_name.restype = C.c_char_p
    for s in SYMBOLS:
        f = getattr(lib, s)
        if f.restype is C.c_int:
            pass
    _lib = lib
    return lib
