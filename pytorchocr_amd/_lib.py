"""ctypes binding of libptocr_hip.so (include/ptocr_hip.h).  There is NO fallback: if the library is missing
or a call fails, a RuntimeError is raised (the product path must never silently run on the CPU)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libptocr_hip.so")


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "N", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "pad_h", "pad_w", "Ho", "Wo",
        "relu", "res_mode", "out_up", "out_ldc", "out_coff", "convt2x2", "cout_store", "res_ldc")]


RES_NONE, RES_ADD_PRE_RELU, RES_ADD_UP2_POST_RELU = 0, 1, 2

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "pytorchocr_amd: %s is missing -- build it with `python -m pytorchocr_amd.build` "
                "(there is no CPU fallback)" % LIB_PATH)
        # torch ships its own HIP runtime; import it FIRST so libptocr_hip.so binds to that same runtime
        # (streams and device pointers are shared with torch -- two runtimes in one process do not see each other)
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        L.ptocr_last_error.restype = C.c_char_p
        L.ptocr_build_tag.restype = C.c_char_p
        from . import build as _b
        tag, want = L.ptocr_build_tag().decode(), _b._flags_tag()
        if tag != want:
            raise RuntimeError(
                "pytorchocr_amd: %s was built with other compile flags (tag %s, expected %s for the flags of this environment) -- a timing "
                "knock-out of tools/dbg left behind?  Rebuild with `python -m pytorchocr_amd.build`" % (LIB_PATH, tag, want))
        L.ptocr_conv3x3_wino4_patches.restype = C.c_long
        L.ptocr_live_allocations.restype = C.c_long
        _lib = L
    return _lib


# every symbol include/ptocr_hip.h declares
EXPORTS = [
    "ptocr_last_error", "ptocr_version", "ptocr_build_tag", "ptocr_device_arch", "ptocr_set_allocator", "ptocr_live_allocations",
    "ptocr_conv2d_f32", "ptocr_conv3x3_wino_f32", "ptocr_wino_set_timing_buffer", "ptocr_conv3x3_wino4_f32", "ptocr_conv3x3_wino4_split_f32", "ptocr_conv3x3_wino4_pool2_f32", "ptocr_conv3x3_wino4_patches", "ptocr_wino4_set_timing_buffer", "ptocr_conv3x3_wino4r_f32", "ptocr_conv3x3_wino4r_pool2_f32", "ptocr_conv3x3_wino4r_pyramid_f32", "ptocr_wino4r_set_timing_buffer", "ptocr_conv7x7s2_stem_f32", "ptocr_conv7x7s2_stem_nchw_f32", "ptocr_conv7x7s2_stem_relu_pool_f32", "ptocr_conv7x7s2_stem_relu_pool_nchw_f32", "ptocr_conv1x1_k64_f32", "ptocr_conv1x1_small_k_f32", "ptocr_conv3x3_small_relu_pool_f32", "ptocr_nchw_to_nhwc_f32", "ptocr_nhwc_to_nchw_f32", "ptocr_maxpool2d_f32",
    "ptocr_convt2x2_sigmoid_f32", "ptocr_db_head_tail_f32", "ptocr_asf_scale_channel_spatial_f32", "ptocr_asf_work_floats", "ptocr_asf_scale_spatial_f32", "ptocr_asf_scale_channel_f32", "ptocr_asf_pyramid_f32", "ptocr_dwconv_f32", "ptocr_dwconv2_f32", "ptocr_cls_head_f32", "ptocr_se_scale_f32", "ptocr_preprocess_u8_f32", "ptocr_warp_crops_u8",
    "ptocr_pwconv_bf16", "ptocr_conv3x3_bf16", "ptocr_conv3x3_lat_bf16", "ptocr_conv3x3_planes_bf16", "ptocr_expand_dw3x3s2_bf16", "ptocr_dwconv_bf16", "ptocr_dwconv_bf16_nblk", "ptocr_se_fc_f32", "ptocr_se_fc_t_f32", "ptocr_se_fc_split_f32", "ptocr_stem3x3s2_bf16", "ptocr_db_head_tail_bf16",
    "ptocr_dbpost_create", "ptocr_dbpost_destroy", "ptocr_db_postprocess", "ptocr_db_postprocess_ex", "ptocr_dbpost_debug_results", "ptocr_dbpost_debug_states", "ptocr_dbpost_debug_labels", "ptocr_dbpost_debug_stamps", "ptocr_dbpost_debug_plant", "ptocr_dbpost_set_route", "ptocr_dbpost_last_device_ms",
    "ptocr_linear_f32", "ptocr_lstm_bidir_f32", "ptocr_lstm_check", "ptocr_lstm_stats", "ptocr_lstm_same_xcd_calls", "ptocr_lstm_fast_workgroups", "ptocr_lstm_set_colocate", "ptocr_lstm_set_spin_limit", "ptocr_ctc_greedy_f32", "ptocr_linear_ctc_greedy_f32", "ptocr_softmax_rows_f32",
]


def missing_exports():
    """Symbols declared in include/ptocr_hip.h that the built library does not export (should be empty)."""
    L = lib()
    return [n for n in EXPORTS if not hasattr(L, n)]


CALLS = 0           # entry points that returned through check(): one kernel launch each on the model paths (bench.py reports launches per forward)


def check(rc, what=""):
    global CALLS
    CALLS += 1
    if rc != 0:
        raise RuntimeError("libptocr_hip %s failed: %s" % (what, lib().ptocr_last_error().decode()))


def ptr(t):
    """torch tensor (device or host) -> void*"""
    return C.c_void_p(t.data_ptr())


def cur_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
