"""ctypes declarations for libebwt2indel.so (include/ebwt2indel.h).  No torch, no fallback:
if the HIP library is not built, importing this module raises."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EB_LIB_PATH") or os.path.join(_HERE, "_build", "libebwt2indel.so")  # EB_LIB_PATH: A/B builds
CLI_PATH = os.path.join(_HERE, "_build", "ebwt2InDel")
CSRC = os.path.join(_HERE, "csrc")

EB_OK, EB_ERR_ARG, EB_ERR_IO, EB_ERR_ALPHABET, EB_ERR_DEVICE, EB_ERR_NOMEM, EB_ERR_STATE = 0, -1, -2, -3, -4, -5, -6
EB_MAX_SLOTS = 8


class EbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ebwt2indel error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("k_left", "k_right", "K", "max_gap", "max_snvs", "mcov_out", "complexity",
                                         "max_variants_per_position", "term")]


class IndexInfo(C.Structure):
    _fields_ = [("n", C.c_uint64), ("F", C.c_uint64 * 4), ("n_blocks", C.c_uint64), ("device_bytes", C.c_uint64),
                ("device", C.c_int32), ("term", C.c_int32)]


class NavStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n", "leaves", "nodes", "lcp_values", "da_values", "minima", "rank_queries", "batches")] + \
               [("seconds_leaves", C.c_double), ("seconds_nodes", C.c_double), ("seconds_traversal", C.c_double)] + \
               [("lcp_pos_sum", C.c_uint64), ("lcp_pos_sq_sum", C.c_uint64)]


class Cluster(C.Structure):
    _fields_ = [("begin", C.c_uint64), ("end", C.c_uint64), ("first_chain", C.c_uint32), ("n_slots", C.c_uint16),
                ("right_found", C.c_uint16), ("right_len", C.c_uint32), ("reserved", C.c_uint32)]


class Chain(C.Structure):
    _fields_ = [("side", C.c_uint8), ("sym", C.c_uint8), ("reserved", C.c_uint16), ("len", C.c_uint32), ("support", C.c_uint32)]


class ScanStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("begin", "end", "runs_closed", "clusters", "bases_in_runs", "survivors", "chains")] + \
               [("hist", C.c_uint64 * 201)] + [(n, C.c_float) for n in ("ms_scan", "ms_extend", "ms_right", "ms_total")] + \
               [("scan_launches", C.c_uint32), ("ms_context", C.c_float), ("requests_extend", C.c_uint64), ("requests_right", C.c_uint64), ("scan_reruns", C.c_uint32), ("batch_tiles", C.c_uint32), ("requests_right_table", C.c_uint64)]


class CallsView(C.Structure):
    _fields_ = [("count", C.c_uint64), ("n_chains", C.c_uint64), ("k_left", C.c_int32), ("k_right", C.c_int32),
                ("clusters", C.POINTER(Cluster)), ("chains", C.POINTER(Chain)), ("left", C.POINTER(C.c_uint64)),
                ("right", C.POINTER(C.c_uint64))]


# every symbol include/ebwt2indel.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "eb_params_init", "eb_params_resolve", "eb_last_error", "eb_device_count",
    "eb_index_create", "eb_index_create_from_file", "eb_index_create_from_device", "eb_index_destroy", "eb_index_bad_symbol",
    "eb_index_info", "eb_index_build_psi", "eb_index_has_psi", "eb_rank4_batch", "eb_access_batch", "eb_lf_range_batch", "eb_lf_range_sym_batch", "eb_fl_batch",
    "eb_lf_node_batch", "eb_rank4_batch_device", "eb_flags_build", "eb_nav_covers_every_slot_once", "eb_flags_build_shard", "eb_flags_device_vectors", "eb_flags_popcounts", "eb_flags_finalize", "eb_flags_set_da", "eb_flags_set_da_file", "eb_flags_set_da_device", "eb_da_stage_file", "eb_flags_set_da_staged", "eb_da_staged_free", "eb_flags_words", "eb_flags_has_da",
    "eb_flags_export", "eb_flags_import", "eb_flags_import_shard_sum", "eb_flags_destroy", "eb_scan_clusters", "eb_scan_clusters_begin", "eb_scan_clusters_end",
    "eb_scan_clusters_cancel", "eb_scan_clusters_begin_ex", "eb_scan_job_wait", "eb_scan_job_device_blob", "eb_scan_job_device_wire", "eb_scan_job_wire_to_host", "eb_host_register", "eb_host_unregister", "eb_calls_get", "eb_calls_free",
    "eb_calls_packed_size", "eb_calls_pack", "eb_calls_unpack", "eb_calls_blob", "eb_write_snp", "eb_format_snp", "eb_free",
    "eb_config_reload", "eb_selftest_pack_fields",
]


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CSRC, "-j8"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (hipcc, gfx950). "
                          "There is no CPU fallback for the ebwt2InDel hot path.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).
    # If torch is imported AFTER this library has pulled in the system runtime, the process ends up with two
    # runtimes and the second finds no GPU.  So when torch is present, let it load its runtime first; the
    # library then binds to that one.  EB_NO_TORCH=1 skips this (pure C / CLI use binds to /opt/rocm).
    # (GPU_MAX_HW_QUEUES is left at the runtime's default of 4 since round 5: the library creates the streams of an index's phase-4
    # jobs right behind the index's own, which puts them on different hardware queues; more queues slow every dispatch)
    if os.environ.get("EB_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    vp, u64, u8, i32 = C.c_void_p, C.c_uint64, C.c_uint8, C.c_int32
    P = C.POINTER
    L.eb_params_init.argtypes = [P(Params)]
    L.eb_params_resolve.argtypes = [P(Params)]
    L.eb_last_error.restype = C.c_char_p
    L.eb_device_count.restype = C.c_int
    L.eb_index_create.argtypes = [vp, u64, u8, C.c_int, P(vp)]
    L.eb_index_create_from_file.argtypes = [C.c_char_p, u8, C.c_int, P(vp)]
    L.eb_index_create_from_device.argtypes = [vp, u64, u8, C.c_int, P(vp)]
    L.eb_index_destroy.argtypes = [vp]
    L.eb_index_destroy.restype = None
    L.eb_index_bad_symbol.argtypes = [P(u8), P(u64)]
    L.eb_index_info.argtypes = [vp, P(IndexInfo)]
    L.eb_index_build_psi.argtypes = [vp]
    L.eb_index_has_psi.argtypes = [vp, P(u64), P(u64)]
    for f in ("eb_rank4_batch", "eb_access_batch", "eb_lf_range_batch", "eb_fl_batch", "eb_lf_node_batch"):
        getattr(L, f).argtypes = [vp, vp, u64, vp]
    L.eb_lf_range_sym_batch.argtypes = [vp, vp, vp, u64, vp]
    L.eb_rank4_batch_device.argtypes = [vp, vp, u64, vp, P(C.c_float)]
    L.eb_flags_build.argtypes = [vp, vp, i32, i32, P(vp), P(NavStats)]
    L.eb_nav_covers_every_slot_once.argtypes = [P(NavStats)]
    L.eb_nav_covers_every_slot_once.restype = C.c_int
    L.eb_flags_build_shard.argtypes = [vp, vp, i32, i32, C.c_uint32, C.c_uint32, P(vp), P(NavStats)]
    L.eb_flags_device_vectors.argtypes = [vp, P(vp), P(u64)]
    L.eb_flags_finalize.argtypes = [vp]
    L.eb_flags_popcounts.argtypes = [vp, P(u64)]
    L.eb_flags_set_da.argtypes = [vp, vp, u64]
    L.eb_flags_set_da_file.argtypes = [vp, C.c_char_p]
    L.eb_flags_set_da_device.argtypes = [vp, vp, u64]
    L.eb_da_stage_file.argtypes = [C.c_char_p, u64, i32, P(vp)]
    L.eb_flags_set_da_staged.argtypes = [vp, vp]
    L.eb_da_staged_free.argtypes = [vp]
    L.eb_da_staged_free.restype = None
    L.eb_flags_words.argtypes = [vp]
    L.eb_flags_words.restype = u64
    L.eb_flags_has_da.argtypes = [vp]
    L.eb_flags_export.argtypes = [vp, vp, vp, vp, vp]
    L.eb_flags_import.argtypes = [u64, C.c_int, vp, vp, vp, vp, P(vp)]
    L.eb_flags_import_shard_sum.argtypes = [u64, C.c_int, vp, vp, vp, vp, P(vp)]
    L.eb_flags_destroy.argtypes = [vp]
    L.eb_flags_destroy.restype = None
    L.eb_scan_clusters.argtypes = [vp, vp, vp, u64, u64, P(Params), P(vp), P(ScanStats)]
    L.eb_scan_clusters_begin.argtypes = [vp, vp, vp, u64, u64, P(Params), P(vp)]
    L.eb_scan_clusters_end.argtypes = [vp, P(vp), P(ScanStats)]
    L.eb_scan_clusters_begin_ex.argtypes = [vp, vp, vp, u64, u64, P(Params), C.c_uint32, P(vp)]
    L.eb_scan_job_wait.argtypes = [vp]
    L.eb_scan_job_device_blob.argtypes = [vp, P(vp), P(u64)]
    L.eb_scan_job_device_wire.argtypes = [vp, P(vp), P(u64)]
    L.eb_scan_job_wire_to_host.argtypes = [vp, vp, u64, P(u64)]
    L.eb_host_register.argtypes = [vp, u64]
    L.eb_host_unregister.argtypes = [vp]
    L.eb_config_reload.argtypes = []
    L.eb_config_reload.restype = None
    L.eb_scan_clusters_cancel.argtypes = [vp]
    L.eb_scan_clusters_cancel.restype = None
    L.eb_calls_get.argtypes = [vp, P(CallsView)]
    L.eb_calls_free.argtypes = [vp]
    L.eb_calls_free.restype = None
    L.eb_calls_packed_size.argtypes = [vp]
    L.eb_calls_packed_size.restype = u64
    L.eb_calls_pack.argtypes = [vp, vp, u64]
    L.eb_calls_unpack.argtypes = [vp, u64, P(vp)]
    L.eb_calls_blob.argtypes = [vp, P(vp), P(u64)]
    L.eb_write_snp.argtypes = [vp, P(Params), C.c_int, C.c_char_p, C.c_int, P(u64), P(u64)]
    L.eb_format_snp.argtypes = [vp, P(Params), C.c_int, P(u64), P(u64), P(vp), P(u64)]
    L.eb_free.argtypes = [vp]
    L.eb_free.restype = None
    L.eb_selftest_pack_fields.argtypes = [i32, vp, u64, vp, P(u64), vp, P(C.c_uint32)]
    return L
