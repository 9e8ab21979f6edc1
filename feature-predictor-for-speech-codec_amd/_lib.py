"""ctypes binding of libfpcodec.so (include/fpcodec.h).  Loading fails loudly: there is
no CPU fallback anywhere in this package."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FPC_LIB_PATH") or os.path.join(_HERE, "libfpcodec.so")  # override: kernel-variant experiments
_lib = None
ABI_VERSION = 4  # the FPC_ABI_VERSION of include/fpcodec.h this binding is written against

SYMBOLS = [
    "fpc_last_error", "fpc_abi_version", "fpc_build_info", "fpc_device_count", "fpc_selftest",
    "fpc_predictor_create", "fpc_predictor_destroy", "fpc_predictor_forward", "fpc_predictor_status",
    "fpc_predictor_set_split", "fpc_predictor_fallback_groups",
    "fpc_codebooks_create", "fpc_codebooks_destroy", "fpc_codebooks_hist_size",
    "fpc_encode", "fpc_decode_features", "fpc_vq_quantize", "fpc_scl_quantize", "fpc_ceps2lpc",
    "fpc_lpcnet_create", "fpc_lpcnet_destroy", "fpc_lpcnet_workspace_bytes",
    "fpc_lpcnet_synthesize", "fpc_lpcnet_condition", "fpc_lpcnet_last_decode_ms",
    "fpc_lpcnet_kernel_variant", "fpc_lpcnet_set_chunk_frames", "fpc_lpcnet_set_pairing",
    "fpc_lpcnet_last_streams_per_workgroup", "fpc_lpcnet_paired_utterances",
    "fpc_trainer_create", "fpc_trainer_destroy", "fpc_trainer_step", "fpc_trainer_export",
    "fpc_cb_workspace_bytes", "fpc_cb_find_nearest", "fpc_cb_update", "fpc_cb_mean0", "fpc_kmeans1d",
]


class FpcError(RuntimeError):
    pass


def build():
    """compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])


class PredictorWeights(C.Structure):
    _fields_ = [("in_features", C.c_int), ("gru_units1", C.c_int), ("gru_units2", C.c_int),
                ("fc_units", C.c_int)] + [(n, C.c_void_p) for n in (
                    "rnn1_weight_ih", "rnn1_weight_hh", "rnn1_bias_ih", "rnn1_bias_hh",
                    "rnn2_weight_ih", "rnn2_weight_hh", "rnn2_bias_ih", "rnn2_bias_hh",
                    "fc_weight", "fc_bias")]


LPCNET_KEYS = ["embed_pitch", "conv1_kernel", "conv1_bias", "conv2_kernel", "conv2_bias",
               "dense1_kernel", "dense1_bias", "dense2_kernel", "dense2_bias", "embed_sig",
               "gru_a_kernel", "gru_a_recurrent", "gru_a_bias", "gru_b_kernel", "gru_b_recurrent",
               "gru_b_bias", "md_kernel", "md_bias", "md_factor"]
LPCNET_SHAPES = {
    "embed_pitch": (256, 64), "conv1_kernel": (3, 84, 128), "conv1_bias": (128,),
    "conv2_kernel": (3, 128, 128), "conv2_bias": (128,), "dense1_kernel": (128, 128),
    "dense1_bias": (128,), "dense2_kernel": (128, 128), "dense2_bias": (128,),
    "embed_sig": (256, 128), "gru_a_kernel": (512, 1152), "gru_a_recurrent": (384, 1152),
    "gru_a_bias": (2, 1152), "gru_b_kernel": (512, 48), "gru_b_recurrent": (16, 48),
    "gru_b_bias": (2, 48), "md_kernel": (256, 16, 2), "md_bias": (256, 2), "md_factor": (256, 2),
}


class LpcnetWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in LPCNET_KEYS]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FpcError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        # the version first: symbol sets and signatures differ between versions (an older library reached through
        # FPC_LIB_PATH must be named as such, not fail at some symbol lookup)
        try:
            got = int(L.fpc_abi_version())
        except AttributeError:
            got = -1
        if got != ABI_VERSION:
            raise FpcError(f"{LIB_PATH} has ABI version {got}, this binding needs {ABI_VERSION} "
                           "(include/fpcodec.h FPC_ABI_VERSION): rebuild the library from this tree")
        L.fpc_last_error.restype = C.c_char_p
        L.fpc_build_info.restype = C.c_char_p
        L.fpc_lpcnet_workspace_bytes.restype = C.c_longlong
        L.fpc_lpcnet_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.fpc_lpcnet_last_decode_ms.restype = C.c_float
        L.fpc_lpcnet_last_decode_ms.argtypes = [C.c_void_p]
        L.fpc_lpcnet_kernel_variant.argtypes = [C.c_void_p]
        L.fpc_lpcnet_set_chunk_frames.argtypes = [C.c_void_p, C.c_int]
        L.fpc_lpcnet_set_pairing.argtypes = [C.c_void_p, C.c_int]
        L.fpc_lpcnet_last_streams_per_workgroup.argtypes = [C.c_void_p]
        L.fpc_lpcnet_paired_utterances.argtypes = [C.c_int, C.c_int]
        L.fpc_trainer_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.fpc_trainer_destroy.argtypes = [C.c_void_p]
        L.fpc_trainer_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_float),
                                       C.c_void_p]
        L.fpc_trainer_export.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.fpc_cb_workspace_bytes.restype = C.c_longlong
        L.fpc_cb_workspace_bytes.argtypes = [C.c_int, C.c_int]
        L.fpc_cb_find_nearest.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_void_p]
        L.fpc_cb_update.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]
        L.fpc_cb_mean0.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.fpc_kmeans1d.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fpc_predictor_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.fpc_predictor_destroy.argtypes = [C.c_void_p]
        L.fpc_predictor_status.argtypes = [C.c_void_p]
        L.fpc_predictor_set_split.argtypes = [C.c_void_p, C.c_int]
        L.fpc_predictor_fallback_groups.argtypes = [C.c_void_p]
        L.fpc_predictor_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
        L.fpc_codebooks_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                           C.POINTER(C.c_void_p)]
        L.fpc_codebooks_destroy.argtypes = [C.c_void_p]
        L.fpc_codebooks_hist_size.argtypes = [C.c_void_p]
        L.fpc_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                 C.c_float, C.c_int] + [C.c_void_p] * 10
        L.fpc_decode_features.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p]
        L.fpc_vq_quantize.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
        L.fpc_scl_quantize.argtypes = L.fpc_vq_quantize.argtypes
        L.fpc_ceps2lpc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]
        L.fpc_lpcnet_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.fpc_lpcnet_destroy.argtypes = [C.c_void_p]
        L.fpc_lpcnet_synthesize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
        L.fpc_lpcnet_condition.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise FpcError(f"{what} failed ({rc}): {lib().fpc_last_error().decode()}")


def require_gpu():
    import torch
    if not torch.cuda.is_available() or lib().fpc_device_count() < 1:
        raise FpcError("fpcodec_amd needs a HIP device (MI355X); there is no CPU fallback")


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
