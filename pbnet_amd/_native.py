"""ctypes loader for libpbnet_hip.so (the C ABI declared in include/pbnet_hip.h).

The library is the product: there is NO CPU fallback.  If it is missing or fails to load this module raises,
loudly, instead of routing anywhere else.  torch is used only to own device memory and streams.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PBNET_HIP_LIB") or os.path.join(_HERE, "libpbnet_hip.so")   # override: A/B of two builds

_lib = None

c_f32p = ctypes.c_void_p
c_i32p = ctypes.c_void_p
c_vp = ctypes.c_void_p
c_int = ctypes.c_int
c_float = ctypes.c_float
c_size = ctypes.c_size_t

c_i64 = ctypes.c_int64
c_i32 = ctypes.c_int32


class CoordsLayout(ctypes.Structure):
    """pbn_coords_layout (include/pbnet_hip.h)."""
    _fields_ = [("counts", c_i64), ("unique_index", c_i64), ("inverse", c_i64),
                ("keys", c_i64 * 5), ("vals", c_i64 * 5), ("coords", c_i64 * 5), ("k3", c_i64 * 5),
                ("parent_row", c_i64 * 4), ("child_k", c_i64 * 4), ("nbr_down", c_i64 * 4), ("up", c_i64 * 4),
                ("k5", c_i64), ("workspace", c_i64), ("workspace_bytes", c_i64),
                ("capacity", c_i32 * 5), ("_pad", c_i32)]


class PrepareLayout(ctypes.Structure):
    """pbn_prepare_layout (include/pbnet_hip.h)."""
    _fields_ = [("pyramid", CoordsLayout), ("n_unique", c_i64), ("unique_index", c_i64), ("inverse", c_i64),
                ("perm", c_i64), ("inv_perm", c_i64), ("ucoords", c_i64), ("tmp_keys", c_i64), ("tmp_vals", c_i64),
                ("uidx32", c_i64), ("inv32", c_i64), ("sort_keys", c_i64), ("sort_vals", c_i64), ("sort_temp", c_i64),
                ("sort_temp_bytes", c_i64)]


class UnetOp(ctypes.Structure):
    """pbn_unet_op (include/pbnet_hip.h)."""
    _fields_ = [("map_kind", c_i32), ("level_in", c_i32), ("level_out", c_i32),
                ("in_buf", c_i32), ("in_col", c_i32), ("res_buf", c_i32), ("res_col", c_i32), ("out_buf", c_i32),
                ("out_col", c_i32), ("vpo", c_i32), ("n_steps", c_i32), ("cout_p", c_i32), ("relu", c_i32),
                ("in2_buf", c_i32), ("in2_col", c_i32), ("vpo2", c_i32),
                ("w", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p)]


class UnetBuf(ctypes.Structure):
    """pbn_unet_buf (include/pbnet_hip.h)."""
    _fields_ = [("level", c_i32), ("width", c_i32)]


class TrainOp(ctypes.Structure):
    """pbn_train_op (include/pbnet_hip.h)."""
    _fields_ = [("map_kind", c_i32), ("level_in", c_i32), ("level_out", c_i32),
                ("in_buf", c_i32), ("in_col", c_i32), ("pre_buf", c_i32), ("res_buf", c_i32), ("res_col", c_i32),
                ("out_buf", c_i32), ("out_col", c_i32), ("relu", c_i32), ("cin", c_i32), ("cout", c_i32),
                ("vpo", c_i32), ("n_steps", c_i32), ("cout_p", c_i32), ("vpo_d", c_i32), ("n_steps_d", c_i32),
                ("cout_p_d", c_i32), ("want_dx", c_i32), ("dx_accumulate", c_i32), ("_pad", c_i32),
                ("w", ctypes.c_void_p), ("w_d", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p),
                ("running_mean", ctypes.c_void_p), ("running_var", ctypes.c_void_p), ("eps", ctypes.c_float),
                ("momentum", ctypes.c_float), ("stat_off", ctypes.c_int64), ("dw_off", ctypes.c_int64),
                ("dgamma_off", ctypes.c_int64), ("dbeta_off", ctypes.c_int64)]


class PairJob(ctypes.Structure):
    """pbn_pair_job (include/pbnet_hip.h)."""
    _fields_ = [("nbr", ctypes.c_void_p), ("n", c_i32), ("n_offsets", c_i32), ("table", ctypes.c_void_p),
                ("totals", ctypes.c_void_p), ("seg_begin", ctypes.c_void_p), ("in_idx", ctypes.c_void_p),
                ("out_idx", ctypes.c_void_p), ("seg_offset", ctypes.c_void_p)]


class PairLists(ctypes.Structure):
    """pbn_pair_lists (include/pbnet_hip.h)."""
    _fields_ = [("in_idx", ctypes.c_void_p), ("out_idx", ctypes.c_void_p), ("seg_begin", ctypes.c_void_p),
                ("counts", ctypes.c_void_p), ("segment", c_i32), ("n_pairs_estimate", c_i32)]


# name -> (restype, argtypes); must list every symbol of include/pbnet_hip.h (tests/test_abi.py checks this)
SIGNATURES = {
    "pbn_version": (ctypes.c_char_p, []),
    "pbn_last_hip_error": (c_int, []),
    "pbn_cluster_workspace_bytes": (c_size, [c_int, c_int, c_int]),
    "pbn_binary_cluster": (c_int, [c_f32p, c_f32p, c_i32p, c_i32p, c_int, c_int, c_float, c_int, c_float, c_int, c_int,
                                   c_i32p, c_i32p, c_i32p, c_f32p, c_i32p, c_i32p, c_i32p, c_i32p, c_vp, c_size, c_vp]),
    "pbn_get_iou": (c_int, [c_i32p, c_i32p, c_vp, c_i32p, c_f32p, c_int, c_int, c_vp]),
    "pbn_cal_iou_and_masklabel": (c_int, [c_i32p, c_i32p, c_vp, c_i32p, c_f32p, c_int, c_int, c_f32p, c_f32p, c_int,
                                          c_vp]),
    "pbn_hash_capacity": (c_int, [c_int]),
    "pbn_coords_workspace_bytes": (c_size, [c_int]),
    "pbn_coords_unique": (c_int, [c_i32p, c_i32p, c_int, c_vp, c_i32p, c_int, c_i32p, c_i32p, c_i32p, c_i32p, c_vp,
                                  c_size, c_vp]),
    "pbn_coords_stride": (c_int, [c_i32p, c_i32p, c_int, c_int, c_vp, c_i32p, c_int, c_i32p, c_i32p, c_i32p, c_i32p,
                                  c_i32p, c_vp, c_size, c_vp]),
    "pbn_kernel_map": (c_int, [c_i32p, c_i32p, c_int, c_i32p, c_int, c_vp, c_i32p, c_int, c_i32p, c_vp]),
    "pbn_up_table": (c_int, [c_i32p, c_i32p, c_i32p, c_int, c_i32p, c_vp]),
    "pbn_spconv_forward": (c_int, [c_vp, c_int, c_int, c_i32p, c_int, c_i32p, c_i32p, c_int, c_vp, c_int, c_int, c_int,
                                   c_f32p, c_f32p, c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_vp, c_size, c_vp]),
    "pbn_spconv_forward_dual": (c_int, [c_vp, c_int, c_int, c_i32p, c_int, c_i32p, c_int, c_vp, c_int, c_int, c_int, c_f32p, c_f32p,
                                        c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_vp, c_size, c_vp, c_int, c_int, c_int, c_vp]),
    "pbn_unet_set_rows_hint": (None, [c_i32p]),
    "pbn_spconv_family": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "pbn_spconv_wgrad_workspace_bytes": (c_size, [c_int, c_int, c_int]),
    "pbn_spconv_wgrad": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_vp, c_vp, c_i32p, c_i32p, c_int, c_int, c_int, c_int, c_int,
                                 c_f32p, c_vp, c_size, c_vp]),
    "pbn_spconv_wgrad_checked": (c_int, [c_vp, c_int, ctypes.c_longlong, c_vp, c_int, ctypes.c_longlong, c_int, c_vp, c_vp, c_i32p,
                                         c_i32p, c_int, c_int, c_int, c_int, c_int, c_int, c_f32p, c_vp, c_size, c_vp]),
    "pbn_gather_rows": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_vp, c_int, c_vp]),
    "pbn_segment_pool_workspace_bytes": (c_size, [c_int, c_int]),
    "pbn_segment_pool": (c_int, [c_vp, c_int, c_int, c_int, c_i32p, c_int, c_f32p, c_f32p, c_vp, c_size, c_vp]),
    "pbn_local_scene_rows": (c_int, [c_i32p, c_i32p, c_i32p, c_f32p, c_int, c_int, c_i32p, c_vp, c_f32p, c_float, c_vp,
                                     c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_i32p, c_vp, c_int, c_vp]),
    "pbn_gather_pad_rows": (c_int, [c_vp, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp]),
    "pbn_mlp_rows": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_f32p,
                             c_int, c_int, c_vp, c_int, c_int, c_vp]),
    "pbn_select_blocks": (c_int, [c_int]),
    "pbn_sem_argmax_table": (c_int, [c_vp, c_int, c_int, c_i32p, c_int, c_int, c_int, c_vp, c_vp, c_i32p, c_i32p, c_vp]),
    "pbn_select_points": (c_int, [c_vp, c_int, c_int, c_i32p, c_i32p, c_f32p, c_vp, c_int, c_int, c_vp, c_f32p, c_f32p,
                                  c_i32p, c_vp]),
    "pbn_mask_count": (c_int, [c_vp, c_int, c_float, c_vp, c_int, c_int, c_int, c_i32p, c_i32p, c_vp]),
    "pbn_proposal_rows": (c_int, [c_vp, c_int, c_float, c_vp, c_vp, c_int, c_i32p, c_i32p, c_f32p, c_float, c_float, c_vp,
                                  c_int, c_int, c_int, c_vp, c_vp, c_i32p, c_vp, c_vp]),
    "pbn_post_words": (c_int, [c_int]),
    "pbn_proposal_bitmask": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_i32p, c_vp]),
    "pbn_mask_iou": (c_int, [c_vp, c_i32p, c_int, c_int, c_i32p, c_f32p, c_vp]),
    "pbn_superpoint_refine": (c_int, [c_vp, c_i32p, c_int, c_int, c_vp, c_int, c_vp, c_i32p, c_vp, c_vp, c_vp, c_i32p,
                                      c_vp]),
    "pbn_bitmask_to_dense": (c_int, [c_vp, c_i32p, c_int, c_int, c_i32p, c_vp]),
    "pbn_instance_overlap": (c_int, [c_i32p, c_int, c_int, c_i32p, c_int, c_i32p, c_vp]),
    "pbn_rulebook_pair_blocks": (c_int, [c_int]),
    "pbn_bn_workspace_bytes": (c_size, [c_int]),
    "pbn_bn_train_forward": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_f32p, c_f32p, ctypes.c_float, ctypes.c_float, c_f32p,
                                     c_f32p, c_vp, c_int, c_f32p, c_f32p, c_vp, c_size, c_vp]),
    "pbn_bn_train_backward": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_f32p, c_f32p, c_f32p, c_vp, c_int,
                                      c_f32p, c_f32p, c_vp, c_size, c_vp]),
    "pbn_bn_act_train_forward": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_f32p, c_f32p, ctypes.c_float, ctypes.c_float,
                                         c_f32p, c_f32p, c_vp, c_int, c_int, c_vp, c_int, c_f32p, c_f32p, c_vp, c_size, c_vp]),
    "pbn_bn_act_train_backward": (c_int, [c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_f32p, c_f32p, c_f32p,
                                          c_vp, c_int, c_vp, c_int, c_f32p, c_f32p, c_vp, c_size, c_vp]),
    "pbn_rulebook_pair_counts": (c_int, [c_i32p, c_int, c_int, c_i32p, c_i32p, c_vp]),
    "pbn_rulebook_pair_fill_dev": (c_int, [c_i32p, c_int, c_int, c_i32p, c_i32p, c_int, c_i32p, c_vp, c_vp, c_vp, c_vp]),
    "pbn_rulebook_pair_fill": (c_int, [c_i32p, c_int, c_int, c_i32p, c_i32p, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "pbn_gather_rulebook_rows": (c_int, [c_vp, c_int, c_int, c_i32p, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "pbn_pack_weights_batch": (c_int, [c_vp, c_int, c_int, c_int, c_vp]),
    "pbn_pack_weight": (c_int, [c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "pbn_kernel_map_cube": (c_int, [c_i32p, c_i32p, c_int, c_int, c_int, c_int, c_vp, c_i32p, c_int, c_i32p, c_vp]),
    "pbn_coords_arena_bytes": (c_size, [c_int, c_int, ctypes.POINTER(CoordsLayout)]),
    "pbn_coords_build": (c_int, [c_i32p, c_i32p, c_int, c_int, c_int, c_vp, c_size, ctypes.POINTER(CoordsLayout), c_vp]),
    "pbn_coords_prepare_bytes": (c_size, [c_int, c_int, ctypes.POINTER(PrepareLayout)]),
    "pbn_coords_prepare": (c_int, [c_i32p, c_int, c_int, c_int, c_vp, c_size, ctypes.POINTER(PrepareLayout), c_vp]),
    "pbn_coords_prepare_dev": (c_int, [c_i32p, c_i32p, c_int, c_int, c_int, c_vp, c_size, ctypes.POINTER(PrepareLayout), c_vp]),
    "pbn_coords_prepare_hash": (c_int, [c_i32p, c_i32p, c_int, c_int, c_int, c_vp, c_size, ctypes.POINTER(PrepareLayout), c_vp]),
    "pbn_unet_forward_dev": (c_int, [ctypes.POINTER(UnetOp), c_int, ctypes.POINTER(UnetBuf), c_int, ctypes.POINTER(c_i32),
                                     c_i32p, c_vp, c_int, ctypes.POINTER(ctypes.c_void_p), c_vp,
                                     ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), c_vp, c_size, c_int,
                                     c_vp, c_size, c_vp]),
    "pbn_class_gate": (c_int, [c_i32p, c_f32p, c_int, c_int, c_int, c_int, c_i32p, c_i32p, c_i32p, c_vp]),
    "pbn_local_plan_workspace_bytes": (c_size, [c_int]),
    "pbn_local_plan": (c_int, [c_i32p, c_int, c_int, c_i32p, c_f32p, c_i32p, c_f32p, c_i32p, c_int, c_int, c_int, c_i32p,
                               c_i32p, c_i32p, c_f32p, c_i32p, c_vp, c_size, c_vp]),
    "pbn_proposal_offsets": (c_int, [c_i32p, c_int, c_vp, c_vp, c_i32p, c_i32p, c_vp]),
    "pbn_batch_starts": (c_int, [c_i32p, c_i32p, c_int, c_int, c_i32p, c_vp]),
    "pbn_local_scene_rows_dev": (c_int, [c_i32p, c_i32p, c_i32p, c_f32p, c_int, c_int, c_i32p, c_i32p, c_i32p, c_vp, c_f32p,
                                         c_float, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_i32p, c_vp,
                                         c_int, c_vp]),
    "pbn_gather_pad_rows_dev": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_i32p, c_vp, c_int, c_vp]),
    "pbn_mlp_rows_dev": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_int, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int,
                                 c_f32p, c_f32p, c_int, c_int, c_vp, c_int, c_int, c_vp]),
    "pbn_mask_count_dev": (c_int, [c_vp, c_int, c_float, c_vp, c_int, c_i32p, c_int, c_int, c_i32p, c_i32p, c_vp]),
    "pbn_proposal_rows_dev": (c_int, [c_vp, c_int, c_float, c_vp, c_vp, c_int, c_i32p, c_i32p, c_i32p, c_f32p, c_float,
                                      c_float, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_i32p, c_vp, c_vp]),
    "pbn_morton_keys": (c_int, [c_i32p, c_i32p, c_int, c_vp, c_vp]),
    "pbn_unet_arena_bytes": (c_size, [ctypes.POINTER(UnetBuf), c_int, ctypes.POINTER(c_i32), c_int,
                                      ctypes.POINTER(c_i64)]),
    "pbn_unet_forward": (c_int, [ctypes.POINTER(UnetOp), c_int, ctypes.POINTER(UnetBuf), c_int, ctypes.POINTER(c_i32),
                                 c_vp, c_int, ctypes.POINTER(ctypes.c_void_p), c_vp, ctypes.POINTER(ctypes.c_void_p),
                                 ctypes.POINTER(ctypes.c_void_p), c_vp, c_size, c_int, c_vp, c_size, c_vp]),
    "pbn_rulebook_pairs_multi": (c_int, [ctypes.POINTER(PairJob), c_int, c_int, c_vp]),
    "pbn_unet_train_forward": (c_int, [ctypes.POINTER(TrainOp), c_int, ctypes.POINTER(UnetBuf), c_int, ctypes.POINTER(c_i32),
                                       c_vp, c_int, ctypes.POINTER(ctypes.c_void_p), c_vp, ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.POINTER(ctypes.c_void_p), c_vp, c_size, c_vp, c_int, c_vp, c_size, c_vp, c_size,
                                       c_vp]),
    "pbn_unet_train_backward": (c_int, [ctypes.POINTER(TrainOp), c_int, ctypes.POINTER(UnetBuf), c_int, ctypes.POINTER(c_i32),
                                        c_vp, c_int, ctypes.POINTER(ctypes.c_void_p), c_vp, ctypes.POINTER(ctypes.c_void_p),
                                        ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(PairLists), c_vp, c_vp, c_size, c_vp,
                                        c_vp, c_vp, c_int, c_int, c_vp, c_size, c_vp, c_size, c_vp, c_size, c_vp]),
    "pbn_unet_forward_timed": (c_int, [ctypes.POINTER(UnetOp), c_int, ctypes.POINTER(UnetBuf), c_int,
                                       ctypes.POINTER(c_i32), c_vp, c_int, ctypes.POINTER(ctypes.c_void_p), c_vp,
                                       ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), c_vp, c_size,
                                       c_int, c_vp, c_size, c_vp, ctypes.POINTER(ctypes.c_float)]),
}

PBN_OK, PBN_ERR_ARG, PBN_ERR_WORKSPACE, PBN_ERR_HIP, PBN_ERR_RANGE, PBN_ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5
ERRORS = {-1: "PBN_ERR_ARG", -2: "PBN_ERR_WORKSPACE", -3: "PBN_ERR_HIP", -4: "PBN_ERR_RANGE", -5: "PBN_ERR_UNSUPPORTED"}


class NativeLibraryError(ImportError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises NativeLibraryError when the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                "libpbnet_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C pbnet_amd/csrc`; there is no CPU fallback." % LIB_PATH)
        try:
            handle = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # missing ROCm runtime etc.
            raise NativeLibraryError("cannot load %s: %s" % (LIB_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        detail = ""
        if rc == -3:
            detail = " (hipError %d)" % lib().pbn_last_hip_error()
        raise RuntimeError("%s failed: %s%s" % (what, ERRORS.get(rc, str(rc)), detail))


def ptr(t):
    """Device (or host) address of a torch tensor, None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor must be contiguous"
    return ctypes.c_void_p(t.data_ptr())


_torch_C = None


def current_stream():
    global _torch_C
    if _torch_C is None:
        import torch
        _torch_C = torch._C
    return ctypes.c_void_p(_torch_C._cuda_getCurrentRawStream(_torch_C._cuda_getDevice()))


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("pbnet_amd runs on the MI355X only: got a %s tensor (no CPU path)" % t.device)
