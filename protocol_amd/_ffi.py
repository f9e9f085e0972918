"""ctypes binding of protocol_amd/lib/libeigentrust_hip.so (C ABI: include/eigentrust_hip.h).

The library is the product; this module only declares its entry points.  There is no fallback:
if the shared object is missing, importing the compute classes raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libeigentrust_hip.so")

ET_OK = 0
ET_E_DUP_MEMBER = -1
ET_E_SET_FULL = -2
ET_E_NOT_MEMBER = -3
ET_E_INSUFFICIENT_PEERS = -4
ET_E_SUM_MISMATCH = -5
ET_E_NONCANONICAL_FR = -6
ET_E_INVALID_ARG = -7
ET_E_HIP = -8
ET_E_NO_DEVICE = -9
ET_E_BUFFER_TOO_SMALL = -10
ET_E_OUT_OF_RANGE = -11
ET_E_UNSUPPORTED = -12
ET_E_THRESHOLD_RANGE = -13
ET_E_RATIO_DIGITS = -14
ET_E_RATIO_MISMATCH = -15
ET_E_ZERO_LIMB = -16
ET_E_BAD_DOMAIN = -17
ET_E_RATIONAL_TOO_LARGE = -18
ET_E_PARSE = -19
ET_E_BAD_SIGNATURE = -20
ET_E_BAD_ABOUT = -21
ET_E_RCCL = -22

ET_VARIANT_FR = 1
ET_VARIANT_F64 = 2
ET_EXCHANGE_AUTO, ET_EXCHANGE_RCCL, ET_EXCHANGE_COPY = 0, 1, 2


class EtFr(C.Structure):
    _fields_ = [("b", C.c_uint8 * 32)]


class EtBuildOpts(C.Structure):
    _fields_ = [("n_seg", C.c_uint32), ("seg", C.c_uint32), ("long_row", C.c_uint32),
                ("target_work", C.c_uint32), ("variants", C.c_uint32), ("device", C.c_int32),
                ("n_chunks", C.c_uint32), ("flags", C.c_uint32), ("xcd_cost", C.c_uint32 * 8)]


ET_BUILD_FULL_VECTOR = 1
ET_BUILD_REUSED = 2


class EtGraphInfo(C.Structure):
    _fields_ = [("n", C.c_uint32), ("n_valid", C.c_uint32), ("n_dangling", C.c_uint32),
                ("n_seg", C.c_uint32), ("seg", C.c_uint32),
                ("nnz", C.c_uint64), ("nnz_local", C.c_uint64), ("padded_local", C.c_uint64),
                ("n_slices", C.c_uint32), ("n_work", C.c_uint32),
                ("rows_cap", C.c_uint32), ("nd_cap", C.c_uint32),
                ("vec_len", C.c_uint64), ("device_bytes", C.c_uint64),
                ("f64_sweep_blocks", C.c_uint32), ("exact_sweep_blocks", C.c_uint32),
                ("n_chunks", C.c_uint32), ("chunk_rows", C.c_uint32),
                ("gather", C.c_uint32), ("full_len", C.c_uint64), ("xcd_cost", C.c_uint32 * 8)]


class EtExchangeInfo(C.Structure):
    _fields_ = [("gather", C.c_uint32), ("n_seg", C.c_uint32), ("seg", C.c_uint32),
                ("own_off", C.c_uint32), ("own_len", C.c_uint32),
                ("vec_len", C.c_uint64), ("full_len", C.c_uint64),
                ("recv_off", C.c_uint32 * 8), ("recv_count", C.c_uint32 * 8),
                ("send_off", C.c_uint32 * 8), ("send_count", C.c_uint32 * 8),
                ("send_total", C.c_uint64),
                ("self_off", C.c_uint32), ("self_count", C.c_uint32)]


class EtSliceDesc(C.Structure):
    _fields_ = [("off", C.c_uint32), ("len", C.c_uint32), ("row0", C.c_uint32), ("nrows", C.c_uint32)]


class EtWorkDesc(C.Structure):
    _fields_ = [("slice_begin", C.c_uint32), ("slice_end", C.c_uint32), ("flags", C.c_uint32), ("dslot", C.c_uint32)]


class EtPlanArrays(C.Structure):
    _fields_ = [("pos_of_slot", C.POINTER(C.c_uint32)), ("slices", C.POINTER(EtSliceDesc)),
                ("work", C.POINTER(EtWorkDesc)), ("col", C.POINTER(C.c_uint32)),
                ("value_u32", C.POINTER(C.c_uint32)), ("value_fr", C.POINTER(EtFr)),
                ("scale_fr", C.POINTER(EtFr)), ("scale_f64", C.POINTER(C.c_double)),
                ("inv_nm1_w", EtFr), ("n_active", C.c_uint32), ("n_active_pad", C.c_uint32),
                ("n_dangling_pad", C.c_uint32), ("n_dangling_local", C.c_uint32)]


class EtSweepStats(C.Structure):
    _fields_ = [("blocks", C.c_uint32), ("clustered", C.c_uint32), ("edges", C.c_uint64), ("edges_most", C.c_uint64),
                ("line_fills", C.c_uint64), ("line_fills_most", C.c_uint64), ("chunk_span_most", C.c_uint64), ("xcd_line_fills", C.c_uint64)]


class EtScore(C.Structure):
    _fields_ = [("address", C.c_uint8 * 20), ("score_fr", C.c_uint8 * 32), ("numerator", C.c_uint8 * 32),
                ("denominator", C.c_uint8 * 32), ("score", C.c_uint8 * 32)]


class EtAttestation(C.Structure):
    _fields_ = [("about", C.c_uint8 * 20), ("domain", C.c_uint8 * 20), ("value", C.c_uint8),
                ("message", C.c_uint8 * 32), ("sig_r", C.c_uint8 * 32), ("sig_s", C.c_uint8 * 32),
                ("rec_id", C.c_uint8)]


class EtIterStats(C.Structure):
    _fields_ = [("sum", EtFr), ("changed", C.c_uint64), ("residual_l1", C.c_double),
                ("sum_f64", C.c_double)]


# every symbol include/eigentrust_hip.h declares, with its prototype
vp, u32, u64, i32, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_double
P = C.POINTER
ET_STEP_FINAL = 1
ET_STEP_NO_STATS = 2
ET_INIT_PLAIN = 1

PROTOTYPES = {
    "et_strerror": (C.c_char_p, [i32]),
    "et_last_error_message": (C.c_char_p, []),
    "et_abi_version": (u32, []),
    "et_device_count": (i32, []),
    "et_set_new": (i32, [u32, u32, u64, u64, vp, P(vp)]),
    "et_set_free": (None, [vp]),
    "et_set_add_member": (i32, [vp, vp]),
    "et_set_remove_member": (i32, [vp, vp]),
    "et_set_update_op": (i32, [vp, vp, vp]),
    "et_set_update_op_sparse": (i32, [vp, vp, u32, vp, vp]),
    "et_set_update_op_u32": (i32, [vp, u32, u32, vp, vp]),
    "et_set_members": (i32, [vp, vp, vp]),
    "et_set_converge": (i32, [vp, vp]),
    "et_set_converge_f64": (i32, [vp, dbl, vp, vp, P(dbl)]),
    "et_set_converge_rational": (i32, [vp, vp, vp, C.c_size_t, P(C.c_size_t)]),
    "et_graph_build": (i32, [u32, vp, u64, vp, vp, vp, vp, P(EtBuildOpts), P(vp)]),
    "et_plan_build": (i32, [u32, vp, u64, vp, vp, vp, vp, P(EtBuildOpts), P(vp)]),
    "et_plan_free": (None, [vp]),
    "et_plan_info_get": (i32, [vp, P(EtGraphInfo)]),
    "et_plan_arrays_get": (i32, [vp, P(EtPlanArrays)]),
    "et_plan_sweep_stats": (i32, [vp, i32, P(EtSweepStats)]),
    "et_graph_from_plan": (i32, [vp, P(EtBuildOpts), P(vp)]),
    "et_graph_free": (None, [vp]),
    "et_graph_info_get": (i32, [vp, P(EtGraphInfo)]),
    "et_graph_converge_fr": (i32, [vp, u32, u64, u64, vp]),
    "et_graph_converge_f64": (i32, [vp, u32, dbl, dbl, vp, vp, P(dbl)]),
    "et_graph_init_fr_async": (i32, [vp, u64, u64, vp, u32, vp]),
    "et_graph_init_f64_async": (i32, [vp, dbl, vp, u32, vp]),
    "et_graph_step_fr_async": (i32, [vp, vp, vp, u32, u32, vp]),
    "et_graph_step_f64_async": (i32, [vp, vp, vp, u32, dbl, vp, u32, vp]),
    "et_graph_step_fr_phase_async": (i32, [vp, vp, vp, u32, u32, u32, vp]),
    "et_graph_step_f64_phase_async": (i32, [vp, vp, vp, u32, dbl, vp, u32, u32, vp]),
    "et_graph_exchange_get": (i32, [vp, P(EtExchangeInfo)]),
    "et_plan_exchange_get": (i32, [vp, P(EtExchangeInfo), P(vp), P(vp)]),
    "et_graph_pack_fr_async": (i32, [vp, vp, vp, vp]),
    "et_graph_pack_f64_async": (i32, [vp, vp, vp, vp]),
    "et_graph_unpermute_fr_async": (i32, [vp, vp, vp, vp]),
    "et_graph_unpermute_f64_async": (i32, [vp, vp, vp, vp]),
    "et_graph_permute_f64_host": (i32, [vp, vp, vp]),
    "et_graph_stats_fr": (i32, [vp, u32, vp, P(EtIterStats)]),
    "et_graph_stats_f64": (i32, [vp, u32, vp, P(EtIterStats)]),
    "et_graph_run_fr": (i32, [vp, u32, u64, u64, vp, i32]),
    "et_graph_run_f64": (i32, [vp, u32, dbl, dbl, vp, i32]),
    "et_graph_record_begin": (i32, [vp, vp]),
    "et_graph_record_end": (i32, [vp, vp]),
    "et_graph_event_times": (i32, [vp, P(C.c_float), P(u32)]),
    "et_graph_result_fr": (i32, [vp, vp, vp]),
    "et_graph_result_f64": (i32, [vp, vp, vp]),
    "et_sharded_create": (i32, [u32, vp, u64, vp, vp, vp, vp, u32, vp, u32, u32, u32, P(vp)]),
    "et_sharded_free": (None, [vp]),
    "et_sharded_exchange_kind": (i32, [vp]),
    "et_sharded_info_get": (i32, [vp, u32, P(EtGraphInfo)]),
    "et_sharded_converge_fr": (i32, [vp, u32, u64, u64, vp]),
    "et_sharded_converge_f64": (i32, [vp, u32, dbl, dbl, vp, vp, P(dbl)]),
    "et_batch_create": (i32, [u32, u32, vp, vp, i32, P(vp)]),
    "et_batch_free": (None, [vp]),
    "et_batch_run_fr_async": (i32, [vp, u32, u64, u64, vp]),
    "et_batch_result_fr": (i32, [vp, vp, vp, vp]),
    "et_batch_edges": (i32, [vp, P(u64)]),
    "et_batch_converge_fr": (i32, [u32, u32, vp, vp, u32, u64, u64, vp, vp]),
    "et_batch_converge_fr_noalloc": (i32, [u32, u32, vp, vp, u32, u64, u64, vp, vp]),
    "et_synth_graph": (i32, [u32, u32, u64, u64, u32, P(u64), vp, vp, vp]),
    "et_synth_graph_range": (i32, [u32, u32, u64, u64, u32, u32, u32, P(u64), vp, vp, vp]),
    "et_dplan_create": (i32, [u32, vp, u64, vp, vp, vp, P(vp)]),
    "et_dplan_local_stats": (i32, [vp, vp, vp, vp, vp, P(u32)]),
    "et_dplan_set_global_stats": (i32, [vp, vp, vp, vp, vp, u32, P(EtBuildOpts)]),
    "et_dplan_outgoing": (i32, [vp, u32, P(u64), P(vp), P(vp), P(vp)]),
    "et_dplan_add_incoming": (i32, [vp, u64, vp, vp, vp]),
    "et_dplan_needs": (i32, [vp, u32, P(u64), P(vp)]),
    "et_dplan_add_wanted": (i32, [vp, u32, u64, vp]),
    "et_dplan_info_get": (i32, [vp, P(EtGraphInfo)]),
    "et_dplan_finish": (i32, [vp, P(vp)]),
    "et_dplan_finish_host": (i32, [vp, P(vp)]),
    "et_dplan_free": (None, [vp]),
    "et_dev_fr_mul": (i32, [u32, vp, vp, vp]),
    "et_ratio_decompose": (i32, [vp, C.c_size_t, vp, C.c_size_t, u32, u32, vp, vp]),
    "et_threshold_check": (i32, [vp, vp, C.c_size_t, vp, C.c_size_t, vp, u32, u32, u32, u64, u64, P(C.c_int)]),
    "et_scalar_from_address": (i32, [vp, vp]),
    "et_score_pack": (i32, [vp, vp, vp, C.c_size_t, vp, C.c_size_t, P(EtScore)]),
    "et_attestations_csv_parse": (i32, [C.c_char_p, C.c_size_t, P(EtAttestation), C.c_size_t, P(C.c_size_t)]),
    "et_attestations_csv_format": (i32, [P(EtAttestation), C.c_size_t, C.c_char_p, C.c_size_t, P(C.c_size_t)]),
    "et_scores_csv_format": (i32, [P(EtScore), C.c_size_t, C.c_char_p, C.c_size_t, P(C.c_size_t)]),
    "et_public_inputs_pack": (i32, [vp, vp, u32, vp, vp, vp, C.c_size_t]),
    "et_public_inputs_unpack": (i32, [vp, C.c_size_t, u32, vp, vp, vp, vp]),
    "et_poseidon5_permute": (i32, [vp, vp]),
    "et_poseidon5_sponge": (i32, [vp, C.c_size_t, vp]),
    "et_keccak256": (i32, [vp, C.c_size_t, vp]),
    "et_attestation_hash": (i32, [P(EtAttestation), vp]),
    "et_attestations_recover": (i32, [P(EtAttestation), C.c_size_t, vp, vp]),
    "et_attestations_recover_gpu": (i32, [P(EtAttestation), C.c_size_t, vp, vp, i32]),
    "et_attestation_recover_lane_host": (i32, [P(EtAttestation), vp, vp]),
    "et_secp256k1_public_key": (i32, [vp, vp, vp]),
    "et_secp256k1_sign": (i32, [vp, vp, vp, vp, vp, vp]),
    "et_opinion_validate": (i32, [vp, P(EtAttestation), vp, vp, u32, vp, vp, vp]),
    "et_et_setup": (i32, [P(EtAttestation), C.c_size_t, vp, u32, u32, u64, u64, P(EtScore), C.c_size_t, P(C.c_size_t), vp, C.c_size_t]),
    "et_scores_from_attestations": (i32, [P(EtAttestation), C.c_size_t, vp, u32, u64, u64, vp, vp, vp, C.c_size_t, P(C.c_size_t)]),
    "et_local_scores": (i32, [P(EtAttestation), vp, C.c_size_t, vp, u32, u32, u64, u64, P(EtScore), C.c_size_t, P(C.c_size_t)]),
}

_lib = None


class EigenTrustError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = lib().et_strerror(code).decode()
        if detail:
            msg += " (" + detail + ")"
        super().__init__("[%d] %s" % (code, msg))


def lib():
    """Loads the shared library; raises if it was not built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libeigentrust_hip.so not found at %s; run `make -C protocol_amd/csrc` "
                              "(or __graft_entry__.build())" % LIB_PATH)
        # A process must hold ONE HIP runtime.  PyTorch ships its own copy and loads it globally; the library then
        # binds to that copy -- but only when PyTorch's is in the process first.  Loaded the other way round, the
        # system runtime initialises the GPU for the library and PyTorch's copy finds no device afterwards.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != ET_OK:
        raise EigenTrustError(rc, lib().et_last_error_message().decode(errors="replace"))
