"""ctypes binding of syzgydb_amd/libsyzgy_scan.so (include/syzgy_scan.h).

The library is the product; this module only loads it and declares the
prototypes.  There is no Python or CPU fallback: if the shared object is
missing, loading fails loudly with the build command to run.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SZG_LIB_PATH selects an alternative build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("SZG_LIB_PATH") or os.path.join(_HERE, "libsyzgy_scan.so")

SZG_EUCLIDEAN = 0
SZG_COSINE = 1

SZG_OK = 0
SZG_E_INVALID = -1
SZG_E_NOMEM = -2
SZG_E_DEVICE = -3
SZG_E_TRUNCATED = -4
SZG_E_NODEVICE = -5
SZG_E_RANGE = -6
SZG_E_UNSUPPORTED = -7

# every symbol include/syzgy_scan.h declares (tests check the .so exports them all)
EXPORTS = [
    "szg_index_create", "szg_index_destroy", "szg_row_bytes", "szg_index_load",
    "szg_index_append", "szg_index_overwrite", "szg_index_tombstone", "szg_index_rows",
    "szg_index_live_rows", "szg_index_read_rows", "szg_search_topk", "szg_search_radius",
    "szg_strerror", "szg_last_error", "szg_abi_version", "szg_set_timing", "szg_get_stats",
    "szg_reset_stats", "szg_set_option", "szg_index_synth", "szg_index_set_row_base",
    "szg_merge_topk", "szg_merge_topk_records", "szg_index_append_f64", "szg_distances", "szg_pair_distances", "szg_index_overwrite_f64",
    "szg_search_radius_batch", "szg_comm_unique_id", "szg_comm_create", "szg_comm_create_host", "szg_comm_destroy",
    "szg_comm_reserve", "szg_index_attach_comm", "szg_search_topk_sharded", "szg_search_radius_sharded",
    "szg_comm_merge_topk", "szg_comm_merge_radius", "szg_comm_get_stats", "szg_comm_reset_stats",
    "szg_comm_last_radius", "szg_comm_chain_topk", "szg_comm_debug_inject",
]
SZG_COMM_ID_BYTES = 128
# int (*szg_allgather_fn)(void *user, const void *send, void *recv, uint64_t bytes_per_rank)
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64)
# int (*szg_replay_fn)(void *user, int j, int k, uint64_t *heap_rows, double *heap_dist, int32_t *heap_n)
REPLAY_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64),
                             ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32))


class SzgCommStats(ctypes.Structure):
    _fields_ = [("exchanges", ctypes.c_uint64), ("exchange_us", ctypes.c_double), ("host_us", ctypes.c_double),
                ("rccl_ranks", ctypes.c_int), ("zero_copy", ctypes.c_int), ("chained_replays", ctypes.c_uint64),
                ("status_rounds", ctypes.c_uint64), ("chain_rounds", ctypes.c_uint64)]

# include/syzgy_pager.h
PAGER_EXPORTS = [
    "szg_pager_open", "szg_pager_close", "szg_pager_options", "szg_pager_count", "szg_pager_skipped",
    "szg_pager_ids", "szg_pager_vectors", "szg_pager_metadata", "szg_pager_load",
]
SZG_E_IO = -8
SZG_E_FORMAT = -9


class SzgStats(ctypes.Structure):
    _fields_ = [
        ("queries", ctypes.c_uint64),
        ("scan_launches", ctypes.c_uint64),
        ("escalations", ctypes.c_uint64),
        ("scan_bytes", ctypes.c_uint64),
        ("scan_ms", ctypes.c_double),
        ("total_ms", ctypes.c_double),
        ("timed_launches", ctypes.c_uint64),
        ("full_replays", ctypes.c_uint64),
        ("mq_launches", ctypes.c_uint64),
        ("mq_queries", ctypes.c_uint64),
        ("mq_fallbacks", ctypes.c_uint64),
        ("host_prep_us", ctypes.c_double),
        ("host_finish_us", ctypes.c_double),
        ("host_enqueue_us", ctypes.c_double),
        ("sketch_queries", ctypes.c_uint64),
        ("sketch_fallbacks", ctypes.c_uint64),
        ("mq_bf16_sweeps", ctypes.c_uint64),
    ]


class SzgError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        msg = "%s failed: %d" % (where, code)
        if detail:
            msg += " (%s)" % detail
        super().__init__(msg)


_lib = None


def load():
    """Load the HIP library; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "syzgydb_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C syzgydb_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp = ctypes.c_void_p
    u8p = ctypes.POINTER(ctypes.c_uint8)
    u64p = ctypes.POINTER(ctypes.c_uint64)
    f64p = ctypes.POINTER(ctypes.c_double)
    i32p = ctypes.POINTER(ctypes.c_int32)
    intp = ctypes.POINTER(ctypes.c_int)

    L.szg_abi_version.restype = ctypes.c_int
    L.szg_abi_version.argtypes = []
    L.szg_strerror.restype = ctypes.c_char_p
    L.szg_strerror.argtypes = [ctypes.c_int]
    L.szg_last_error.restype = ctypes.c_char_p
    L.szg_last_error.argtypes = []
    L.szg_row_bytes.restype = ctypes.c_int64
    L.szg_row_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    L.szg_index_create.restype = ctypes.c_int
    L.szg_index_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   intp, ctypes.c_int]
    L.szg_index_destroy.restype = None
    L.szg_index_destroy.argtypes = [vp]
    L.szg_index_load.restype = ctypes.c_int
    L.szg_index_load.argtypes = [vp, u8p, ctypes.c_uint64]
    L.szg_index_append.restype = ctypes.c_int
    L.szg_index_append.argtypes = [vp, u8p, ctypes.c_uint64]
    L.szg_index_overwrite.restype = ctypes.c_int
    L.szg_index_overwrite.argtypes = [vp, ctypes.c_uint64, u8p]
    L.szg_index_tombstone.restype = ctypes.c_int
    L.szg_index_tombstone.argtypes = [vp, ctypes.c_uint64]
    L.szg_index_rows.restype = ctypes.c_uint64
    L.szg_index_rows.argtypes = [vp]
    L.szg_index_live_rows.restype = ctypes.c_uint64
    L.szg_index_live_rows.argtypes = [vp]
    L.szg_index_read_rows.restype = ctypes.c_int
    L.szg_index_read_rows.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64, u8p]
    L.szg_search_topk.restype = ctypes.c_int
    L.szg_search_topk.argtypes = [vp, f64p, ctypes.c_int, ctypes.c_int, u64p, u64p, f64p, i32p]
    L.szg_search_radius.restype = ctypes.c_int
    L.szg_search_radius.argtypes = [vp, f64p, ctypes.c_double, u64p, u64p, f64p, ctypes.c_uint64,
                                    u64p]
    L.szg_set_timing.restype = ctypes.c_int
    L.szg_set_timing.argtypes = [vp, ctypes.c_int]
    L.szg_get_stats.restype = ctypes.c_int
    L.szg_get_stats.argtypes = [vp, ctypes.POINTER(SzgStats)]
    L.szg_reset_stats.restype = ctypes.c_int
    L.szg_reset_stats.argtypes = [vp]
    L.szg_set_option.restype = ctypes.c_int
    L.szg_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_int64]
    L.szg_index_synth.restype = ctypes.c_int
    L.szg_index_synth.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]
    L.szg_index_set_row_base.restype = ctypes.c_int
    L.szg_index_set_row_base.argtypes = [vp, ctypes.c_uint64]
    L.szg_merge_topk.restype = ctypes.c_int
    L.szg_merge_topk.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, u64p, f64p,
                                 i32p, u64p, f64p, i32p, u8p]
    L.szg_merge_topk_records.restype = ctypes.c_int
    L.szg_merge_topk_records.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_int64), u64p, f64p, i32p, u8p]
    L.szg_index_append_f64.restype = ctypes.c_int
    L.szg_index_append_f64.argtypes = [vp, f64p, ctypes.c_uint64]
    L.szg_index_overwrite_f64.restype = ctypes.c_int
    L.szg_index_overwrite_f64.argtypes = [vp, ctypes.c_uint64, f64p]
    L.szg_distances.restype = ctypes.c_int
    L.szg_distances.argtypes = [vp, f64p, u64p, ctypes.c_uint64, f64p]
    L.szg_pair_distances.restype = ctypes.c_int
    L.szg_pair_distances.argtypes = [vp, u64p, u64p, ctypes.c_uint64, f64p]
    L.szg_search_radius_batch.restype = ctypes.c_int
    L.szg_search_radius_batch.argtypes = [vp, f64p, ctypes.c_int, f64p, u64p, u64p, f64p, ctypes.c_uint64, u64p]
    L.szg_comm_unique_id.restype = ctypes.c_int
    L.szg_comm_unique_id.argtypes = [u8p]
    L.szg_comm_create.restype = ctypes.c_int
    L.szg_comm_create.argtypes = [ctypes.POINTER(vp), u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.szg_comm_create_host.restype = ctypes.c_int
    L.szg_comm_create_host.argtypes = [ctypes.POINTER(vp), ALLGATHER_FN, vp, ctypes.c_int, ctypes.c_int]
    L.szg_comm_destroy.restype = None
    L.szg_comm_destroy.argtypes = [vp]
    L.szg_comm_reserve.restype = ctypes.c_int
    L.szg_comm_reserve.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.szg_index_attach_comm.restype = ctypes.c_int
    L.szg_index_attach_comm.argtypes = [vp, vp]
    L.szg_search_topk_sharded.restype = ctypes.c_int
    L.szg_search_topk_sharded.argtypes = [vp, f64p, ctypes.c_int, ctypes.c_int, u64p, u64p, f64p, i32p, u8p]
    L.szg_search_radius_sharded.restype = ctypes.c_int
    L.szg_search_radius_sharded.argtypes = [vp, f64p, ctypes.c_int, f64p, u64p, u64p, f64p, ctypes.c_uint64, u64p]
    L.szg_comm_merge_topk.restype = ctypes.c_int
    L.szg_comm_merge_topk.argtypes = [vp, ctypes.c_int, ctypes.c_int, u64p, f64p, i32p, u64p, f64p, i32p, u8p]
    L.szg_comm_merge_radius.restype = ctypes.c_int
    L.szg_comm_merge_radius.argtypes = [vp, ctypes.c_int, u64p, u64p, f64p, u64p, f64p, ctypes.c_uint64, u64p]
    L.szg_comm_get_stats.restype = ctypes.c_int
    L.szg_comm_get_stats.argtypes = [vp, ctypes.POINTER(SzgCommStats)]
    L.szg_comm_reset_stats.restype = ctypes.c_int
    L.szg_comm_reset_stats.argtypes = [vp]
    L.szg_comm_last_radius.restype = ctypes.c_int
    L.szg_comm_last_radius.argtypes = [vp, ctypes.c_int, u64p, f64p, ctypes.c_uint64, u64p]
    L.szg_comm_chain_topk.restype = ctypes.c_int
    L.szg_comm_chain_topk.argtypes = [vp, ctypes.c_int, ctypes.c_int, REPLAY_FN, vp, u64p, f64p, i32p]
    L.szg_comm_debug_inject.restype = ctypes.c_int
    L.szg_comm_debug_inject.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.szg_pager_open.restype = ctypes.c_int
    L.szg_pager_open.argtypes = [ctypes.POINTER(vp), ctypes.c_char_p, ctypes.c_int]
    L.szg_pager_close.restype = None
    L.szg_pager_close.argtypes = [vp]
    L.szg_pager_options.restype = ctypes.c_int
    L.szg_pager_options.argtypes = [vp, intp, intp, intp]
    L.szg_pager_count.restype = ctypes.c_uint64
    L.szg_pager_count.argtypes = [vp]
    L.szg_pager_skipped.restype = ctypes.c_uint64
    L.szg_pager_skipped.argtypes = [vp]
    L.szg_pager_ids.restype = ctypes.c_int
    L.szg_pager_ids.argtypes = [vp, u64p]
    L.szg_pager_vectors.restype = ctypes.c_int
    L.szg_pager_vectors.argtypes = [vp, u8p, ctypes.c_uint64]
    L.szg_pager_metadata.restype = ctypes.c_int
    L.szg_pager_metadata.argtypes = [vp, ctypes.c_uint64, ctypes.POINTER(u8p), u64p]
    L.szg_pager_load.restype = ctypes.c_int
    L.szg_pager_load.argtypes = [vp, vp]
    L.szg_debug_f64_probe.restype = ctypes.c_int
    L.szg_debug_f64_probe.argtypes = [ctypes.c_int, f64p, f64p, f64p, ctypes.c_uint64]
    _lib = L
    return L


def check(code, where):
    if code != SZG_OK:
        L = load()
        detail = L.szg_last_error().decode("utf-8", "replace")
        if not detail:
            detail = L.szg_strerror(code).decode()
        raise SzgError(code, where, detail)
