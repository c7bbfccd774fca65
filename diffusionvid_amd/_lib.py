"""ctypes binding of libdvid_hip.so (the C ABI declared in include/dvid_hip.h).

The product path has no CPU fallback: if the shared library is missing, cannot be loaded, or no
HIP device is visible, the first use raises.  Build with `python -c "import __graft_entry__ as g;
g.build()"` or `make -C diffusionvid_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DVID_LIB") or os.path.join(_HERE, "libdvid_hip.so")          # DVID_LIB: another build of the library (A/B timing)

c_void_p, c_int, c_float, c_int64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
c_float_p = C.POINTER(C.c_float)


class DvidConfig(C.Structure):
    _fields_ = [
        ("hidden_dim", c_int), ("nheads", c_int), ("dim_feedforward", c_int), ("dim_dynamic", c_int),
        ("num_classes", c_int), ("num_cls", c_int), ("num_reg", c_int), ("num_heads", c_int),
        ("num_heads_cond", c_int), ("pooler_resolution", c_int), ("sampling_ratio", c_int),
        ("res_blocks", c_int * 4), ("pixel_mean", c_float * 3), ("pixel_std", c_float * 3),
        ("backbone_type", c_int), ("swin_embed_dim", c_int), ("swin_depths", c_int * 4), ("swin_heads", c_int * 4),
        ("swin_window", c_int),
    ]


# name -> (restype, argtypes); every symbol of include/dvid_hip.h
SIGNATURES = {
    "dvid_last_error": (C.c_char_p, []),
    "dvid_version": (c_int, []),
    "dvid_model_create": (c_int, [C.POINTER(DvidConfig), C.POINTER(c_void_p)]),
    "dvid_model_destroy": (c_int, [c_void_p]),
    "dvid_model_set_tensor": (c_int, [c_void_p, C.c_char_p, c_void_p, C.POINTER(c_int64), c_int]),
    "dvid_model_finalize": (c_int, [c_void_p]),
    "dvid_model_set_precision": (c_int, [c_void_p, c_int]),
    "dvid_model_take_range_flag": (c_int, [c_void_p, C.POINTER(c_int), c_void_p]),
    "dvid_set_chains": (c_int, [c_void_p, c_int]),
    "dvid_set_stem_layout": (c_int, [c_void_p, c_int]),
    "dvid_workspace_reserve": (c_int, [c_void_p, c_int, c_int, c_int, c_int]),
    "dvid_workspace_generation": (C.c_ulonglong, [c_void_p]),
    "dvid_backbone_resnet_fpn": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dvid_backbone_swin_fpn": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dvid_backbone_resnet_fpn_frames": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dvid_backbone_swin_fpn_frames": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dvid_rcnn_head": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                               c_void_p, C.POINTER(c_int64), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dvid_global_xattn": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "dvid_global_memory_project": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "dvid_roialign_v2_multilevel": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                            c_void_p, c_void_p, c_void_p]),
    "dvid_roialign_v2_multilevel_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                                c_void_p, c_void_p, c_void_p]),
    "dvid_conv2d_nhwc_f32": (c_int, [c_void_p] * 8 + [c_int] * 12 + [c_void_p]),
    "dvid_mha_f32": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_int64] * 3 + [c_void_p]),
    "dvid_dynconv_f32": (c_int, [c_void_p] * 7 + [c_int, c_void_p]),
    "dvid_select_topk_features": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                          c_void_p]),
    "dvid_counter_normal": (c_int, [c_void_p, C.c_int64, c_int, C.c_uint64, c_void_p]),
    "dvid_noise_to_boxes": (c_int, [c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_void_p]),
    "dvid_ddim_renew_step": (c_int, [c_void_p] * 6 + [c_int] * 3 + [c_float] * 9 + [c_void_p]),
    "dvid_postproc_topk_nms": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_int,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dvid_cdist": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "dvid_fps_greedy": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dvid_gather_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dvid_conv2d_nhwc_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 13 + [c_void_p]),
    "dvid_bottleneck64_tail_f16": (c_int, [c_void_p] * 10 + [c_int] + [c_void_p] * 2 + [c_int] * 3 + [c_void_p]),
    "dvid_bottleneck128_tail_f16": (c_int, [c_void_p] * 10 + [c_int] * 3 + [c_void_p]),
    "dvid_mha_f16": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_int64] * 3 + [c_void_p]),
    "dvid_dynconv": (c_int, [c_void_p] * 7 + [c_int, c_void_p]),
    "dvid_add_layernorm": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_void_p]),
    "dvid_nhwc_from_nchw": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dvid_nchw_from_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dvid_f32_to_f16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "dvid_resize_u8_to_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                      c_void_p, c_void_p, c_int, c_void_p]),
    "dvid_igemm_num_configs": (c_int, []),
    "dvid_igemm_set_config": (c_int, [c_int]),
    "dvid_igemm_set_tuning": (c_int, [c_int]),
    "dvid_igemm_tuning_passes": (C.c_longlong, []),
    "dvid_igemm_set_conv3x3": (c_int, [c_int]),
    "dvid_igemm_set_wstat": (c_int, [c_int]),
    "dvid_igemm_set_bottleneck_fusion": (c_int, [c_int]),
    "dvid_set_stem_pool": (c_int, [c_int]),
    "dvid_set_option": (c_int, [C.c_char_p, c_int]),
    "dvid_get_option": (c_int, [C.c_char_p, C.POINTER(c_int)]),
    "dvid_reset_options": (c_int, []),
    "dvid_effective_config": (c_int, [C.c_char_p, c_int]),
    "dvid_profile_enable": (c_int, [c_int]),
    "dvid_profile_reset": (c_int, []),
    "dvid_profile_dump": (c_int, [C.c_char_p]),
    "dvid_profile_read": (c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(c_int64)]),
    "dvid_profile_read_bytes": (c_int, [C.POINTER(C.c_double)]),
}

_lib = None


class DvidError(RuntimeError):
    pass


def load():
    """Load the shared library and bind every declared symbol (no GPU needed for this step)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DvidError(f"{LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build()); "
                        "there is no CPU fallback")
    # torch first: it carries its own copy of the HIP runtime, and whichever copy a process loads first is the one the device belongs to.
    # Loaded after libdvid_hip's (the system's libamdhip64), torch's copy takes the GPU and the library's hipGetDeviceCount answers 0 --
    # `__graft_entry__.build(); smoke()` in ONE process failed that way ("no HIP device available") while either call alone worked.
    try:
        import torch  # noqa: F401
    except ImportError:          # symbol / build checks without torch still work
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().dvid_last_error().decode("utf-8", "replace")
        raise DvidError(f"libdvid_hip {what} failed with code {rc}: {msg}")


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)


def ptr(t):
    """device/host pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "libdvid_hip takes contiguous tensors"
    return t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
