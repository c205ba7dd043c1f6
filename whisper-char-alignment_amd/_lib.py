"""ctypes binding of libwca.so (the C ABI declared in include/wca.h).

There is deliberately NO fallback: if the HIP library is missing or fails to load, importing the
product path raises. PyTorch is used by callers only for device memory and streams.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwca.so")


class WcaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("wca error %d: %s" % (code, msg))
        self.code = code


class TooLongError(WcaError):
    """n_tok > 448 or max_frames > 1500 (the skip condition of infer_ali.py:79)."""


class ModelDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
        "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")]


class AlignOpts(C.Structure):
    _fields_ = [("aggregation", C.c_int32), ("topk", C.c_int32), ("w_colnorm", C.c_float),
                ("w_rownorm", C.c_float), ("w_coverage", C.c_float), ("sot_len", C.c_int32),
                ("medfilt_width", C.c_int32), ("qk_scale", C.c_float)]


class DecodeOpts(C.Structure):
    _fields_ = [("sample_len", C.c_int32), ("eot", C.c_int32), ("timestamp_begin", C.c_int32),
                ("apply_timestamp_rules", C.c_int32), ("max_initial_timestamp_index", C.c_int32), ("no_speech", C.c_int32)]


ERR_TOO_LONG = -2   # WCA_ERR_TOO_LONG
AGGR_MEAN, AGGR_TOPK = 0, 1
PRECISION_SITES = {"logmel": 1, "conv": 2, "enc_gemm": 4, "enc_attn": 8, "cross_kv": 16, "dec": 32, "capture": 64}  # WCA_PSITE_*
SITES = {"qkv": 0, "attention": 1, "out_proj": 2, "fc1": 3, "fc2": 4, "ln1": 5, "ln2": 6}  # WCA_SITE_* of include/wca.h
DTYPE_F32, DTYPE_F16 = 0, 1

_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
_pi32 = C.POINTER(C.c_int32)
_pf = C.POINTER(C.c_float)

# name -> (restype, argtypes); must list every symbol declared in include/wca.h
SIGNATURES = {
    "wca_last_error": (C.c_char_p, []),
    "wca_version": (_i, []),
    "wca_engine_create": (_i, [C.POINTER(ModelDims), _i, _i, C.POINTER(_vp)]),
    "wca_engine_create_ex": (_i, [C.POINTER(ModelDims), _i, _i, _i, C.POINTER(_vp)]),
    "wca_engine_destroy": (None, [_vp]),
    "wca_engine_set_stream": (_i, [_vp, _vp]),
    "wca_engine_synchronize": (_i, [_vp]),
    "wca_load_weight": (_i, [_vp, C.c_char_p, _vp, _i, C.POINTER(_i64), _i]),
    "wca_finalize_weights": (_i, [_vp]),
    "wca_weights_inexact": (_i, [_vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_char_p, _i]),
    "wca_set_allow_rounded_weights": (_i, [_vp, _i]),
    "wca_log_mel": (_i, [_vp, _vp, _i64, _pi32, _i, _vp]),
    "wca_get_attentions": (_i, [_vp, _vp, _vp, _i, _i, _pi32, _pi32, _i, _f, _vp, _vp]),
    "wca_median_filter": (_i, [_vp, _vp, _vp, _i64, _i, _i]),
    "wca_filter_attention": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _pf, _pi32, _pf]),
    "wca_force_align": (_i, [_vp, _vp, _i, _i, _i, _i, C.POINTER(AlignOpts), _pf, _pi32, _pi32, _pi32, _pi32, _pf]),
    "wca_default_find_alignment": (_i, [_vp, _vp, _i, _i, _i, _i, _pi32, _i, _i, _vp, _pf, _pi32, _pi32, _pi32]),
    "wca_attention_weights": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "wca_flac_info": (_i, [_vp, _i64, _pi32, _pi32, _pi32, C.POINTER(_i64)]),
    "wca_flac_decode": (_i, [_vp, _i64, _vp, _i64, C.POINTER(_i64)]),
    "wca_dtw": (_i, [_vp, _pf, _i, _i, _pi32, _pi32, _pi32]),
    "wca_dtw_batch_dev": (_i, [_vp, _vp, _i, _i, _i, _pi32]),
    "wca_probe_heads": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _pf, _pi32]),
    "wca_align_batch": (_i, [_vp, _vp, _i64, _pi32, _vp, _i, _pi32, _pi32, _i, C.POINTER(AlignOpts), _pi32, _pi32]),
    "wca_align_batch_enqueue": (_i, [_vp, _vp, _i64, _pi32, _vp, _i, _pi32, _pi32, _i, C.POINTER(AlignOpts)]),
    "wca_align_batch_fetch": (_i, [_vp, _i, _i, _i, _pi32, _pi32]),
    "wca_encode_batch": (_i, [_vp, _vp, _vp, _i64, _pi32, _i]),
    "wca_greedy_decode": (_i, [_vp, _vp, _vp, _i64, _pi32, _i, _pi32, _i, _vp, _vp, C.POINTER(DecodeOpts), _pi32, _pi32, _pf, _pf]),
    "wca_test_decode_select": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, C.POINTER(DecodeOpts), _vp, _vp]),
    "wca_test_gemm": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "wca_test_gemm_ln": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "wca_test_gemm_stamped": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "wca_test_set_attn_split_drop": (_i, [_i]),
    "wca_test_set_switch": (_i, [C.c_char_p, _i]),
    "wca_test_last_scores": (_i, [_vp, _i, _pf]),
    "wca_test_gemm_pairs": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "wca_test_attention": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i]),
    "wca_test_attention_split": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i]),
    "wca_test_attention_stamped": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "wca_test_layernorm": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i]),
    "wca_test_layernorm_split": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i]),
    "wca_test_encoder": (_i, [_vp, _vp, _i, _vp]),
    "wca_last_stage_ms": (_i, [_vp, _pf]),
    "wca_set_profiling": (_i, [_vp, _i]),
    "wca_last_kernel_ms": (_i, [_vp, _i, C.POINTER(C.c_int), _pf, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "wca_set_overlap": (_i, [_vp, _i]),
    "wca_set_cu_partition": (_i, [_vp, _i]),
    "wca_set_fuse_ln": (_i, [_vp, _i]),
    "wca_set_precision": (_i, [_vp, _i]),
    "wca_get_precision": (_i, [_vp]),
    "wca_set_precision_sites": (_i, [_vp, C.c_uint, _i]),
    "wca_get_precision_sites": (_i, [_vp, C.POINTER(C.c_uint), C.POINTER(C.c_int)]),
    "wca_set_decode_mode": (_i, [_vp, _i, _i]),
    "wca_comm_unique_id": (_i, [_vp]),
    "wca_comm_init": (_i, [_vp, _vp, _i, _i]),
    "wca_comm_destroy": (_i, [_vp]),
    "wca_allgather_results": (_i, [_vp, _vp, _i64, _vp, _i64, C.POINTER(_i64)]),
    "wca_allreduce_counters": (_i, [_vp, C.POINTER(_i64), _i]),
    "wca_collate_plan": (_i, [C.POINTER(_i64), C.POINTER(_i64), _i, C.POINTER(_i64)]),
    "wca_probe_strict_tp": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, C.c_double, _vp]),
    "wca_test_gemm_rows": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i]),
}

_lib = None


def load():
    """Loads libwca.so (once). Raises if it has not been built -- there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libwca.so not found at %s: build it with `python whisper-char-alignment_amd/build.py` "
            "(or __graft_entry__.build()). The alignment engine has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code):
    if code == 0:
        return
    msg = load().wca_last_error()
    msg = msg.decode("utf-8", "replace") if msg else ""
    if code == -2:
        raise TooLongError(code, msg)
    raise WcaError(code, msg)


def i32_array(values):
    arr = (C.c_int32 * len(values))(*[int(v) for v in values])
    return arr
