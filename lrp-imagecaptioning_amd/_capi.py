"""ctypes binding of liblrp_hip.so (include/lrp_hip.h).  The HIP library IS the
product path: if it is missing or fails to load this module raises — there is no
CPU fallback."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liblrp_hip.so")

LRP_ABI_VERSION = 6
LRP_OK, LRP_ERR_INVALID, LRP_ERR_STATE, LRP_ERR_HIP, LRP_ERR_NOMEM, LRP_ERR_RANGE, LRP_ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6
LRP_DEC_ADAPTIVE, LRP_DEC_GRIDTD = 0, 1
LRP_ENC_VGG, LRP_ENC_RESNET = 0, 1
LRP_EXPLAIN_SEQUENCE, LRP_EXPLAIN_SINGLE_STEP = 0, 1
LRP_PREC_FP32, LRP_PREC_BF16X3, LRP_PREC_BF16X3_FAST, LRP_PREC_F16X2 = 0, 1, 2, 3
LRP_MAX_CONV = 32
LRP_TRAIN_FP32, LRP_TRAIN_BF16 = 0, 1


class LrpConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("decoder", C.c_int32),
        ("img_h", C.c_int32), ("img_w", C.c_int32), ("n_conv", C.c_int32),
        ("conv_cin", C.c_int32 * LRP_MAX_CONV), ("conv_cout", C.c_int32 * LRP_MAX_CONV),
        ("conv_pool_after", C.c_int32 * LRP_MAX_CONV), ("conv_name", (C.c_char * 32) * LRP_MAX_CONV),
        ("L", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("E", C.c_int32), ("V", C.c_int32),
        ("max_images", C.c_int32), ("max_tokens", C.c_int32), ("max_caption_len", C.c_int32),
        ("sos_id", C.c_int32), ("eos_id", C.c_int32),
        ("encoder", C.c_int32), ("resnet_stem", C.c_int32), ("resnet_n_stacks", C.c_int32),
        ("resnet_filters", C.c_int32 * 8), ("resnet_blocks", C.c_int32 * 8),
    ]


# every symbol include/lrp_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "lrp_create": (C.c_int, [C.POINTER(LrpConfig), C.POINTER(_P)]),
    "lrp_destroy": (C.c_int, [_P]),
    "lrp_set_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_int32, C.POINTER(C.c_int64)]),
    "lrp_set_weight_dev": (C.c_int, [_P, C.c_char_p, _P, C.c_int32, C.POINTER(C.c_int64), _P]),
    "lrp_encode_images": (C.c_int, [_P, _P, C.c_int32, _P]),
    "lrp_set_features": (C.c_int, [_P, _P, C.c_int32, _P]),
    "lrp_get_features": (C.c_int, [_P, _P, C.c_int32, _P]),
    "lrp_decoder_forward": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, _P]),
    "lrp_read_state": (C.c_int, [_P, C.c_char_p, _P, C.c_size_t, _P]),
    "lrp_decoder_explain": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, _P, _P, _P, _P]),
    "lrp_cnn_explain": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), _P, _P, _P]),
    "lrp_explain_tokens": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, _P, _P, _P, _P, _P]),
    "lrp_set_precision": (C.c_int, [_P, C.c_int32]),
    "lrp_set_fast_layers": (C.c_int, [_P, C.c_int64]),
    "lrp_profile_enable": (C.c_int, [_P, C.c_int32]),
    "lrp_profile_query": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "lrp_profile_records": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "lrp_workspace_bytes": (C.c_int64, [_P]),
    "lrp_op_conv": (C.c_int, [_P, _P, _P, _P, _P] + [C.c_int32] * 7 + [_P]),
    "lrp_op_epsilon_dense": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P]),
    "lrp_op_batchnorm_lrp": (C.c_int, [_P, _P, _P, _P, _P, C.c_float, _P, _P, C.c_int64, C.c_int32, _P]),
    "lrp_op_add_lrp": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, _P]),
    "lrp_decoder_gen_begin": (C.c_int, [_P, C.c_int32, _P]),
    "lrp_decoder_gen_step": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _P, _P]),
    "lrp_op_log_softmax_topk": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "lrp_decoder_gradient": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P]),
    "lrp_cnn_walk": (C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_int32, _P]),
    "lrp_op_avgpool_lrp": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "lrp_op_sgemm": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                               C.c_int32, _P, C.c_int64, _P]),
    "lrp_op_conv_wgrad": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int64, _P]),
    "lrp_op_conv_wgrad_bf16": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int64, _P]),
    "lrp_train_set_precision": (C.c_int, [_P, C.c_int32]),
    "lrp_train_begin": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]),
    "lrp_train_flat_size": (C.c_int64, [_P]),
    "lrp_train_num_params": (C.c_int32, [_P]),
    "lrp_train_param_info": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "lrp_train_step": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "lrp_train_forward": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "lrp_train_drop_forward": (C.c_int, [_P, _P]),
    "lrp_reload_switches": (C.c_int, []),
    "lrp_op_conv_pool_sparse": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "lrp_train_apply": (C.c_int, [_P, _P, _P]),
    "lrp_train_get_master": (C.c_int, [_P, _P, _P]),
    "lrp_heatmap_render": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P]),
    "lrp_preprocess_images": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "lrp_heatmap_scores": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "lrp_last_error": (C.c_char_p, []),
    "lrp_abi_version": (C.c_int, []),
    "lrp_launch_count": (C.c_int64, []),
}

_lib = None


class LrpLibraryMissing(RuntimeError):
    pass


def load():
    """dlopen the HIP library (needs libamdhip64; loading works without a GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64.  Streams and device pointers
    # cross this ABI, so the library MUST bind to the HIP runtime torch uses: import torch first
    # (same SONAME -> the dynamic linker reuses the already loaded copy).  Loading in the other
    # order leaves two runtimes in the process ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise LrpLibraryMissing(
            "%s not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the LRP hot path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.lrp_abi_version() != LRP_ABI_VERSION:
        raise LrpLibraryMissing("liblrp_hip.so ABI %d != binding ABI %d" % (lib.lrp_abi_version(), LRP_ABI_VERSION))
    _lib = lib
    return lib


_EXC = {LRP_ERR_INVALID: ValueError, LRP_ERR_STATE: RuntimeError, LRP_ERR_HIP: RuntimeError,
        LRP_ERR_NOMEM: MemoryError, LRP_ERR_RANGE: NotImplementedError, LRP_ERR_UNSUPPORTED: NotImplementedError}


def check(rc):
    """Map status codes to the exception types the reference raises:
    NotImplementedError('index out of range of captions') E:538-539, ValueError AB:332-333."""
    if rc != LRP_OK:
        msg = load().lrp_last_error().decode("utf-8", "replace")
        raise _EXC.get(rc, RuntimeError)(msg)
