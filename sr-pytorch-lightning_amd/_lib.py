"""ctypes binding of libsrk_gfx950.so (the C ABI declared in include/srk.h).

The structures below mirror include/srk.h field for field.  There is NO fallback:
if the shared library is missing or a launcher returns non-zero, a RuntimeError
is raised (the product path never routes through the CPU oracle).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SRK_LIB_PATH: A/B timing of two builds of the same library (tools/); never a different implementation
# SRK_EXACT_RELU=1: the build of the same sources whose ReLU keeps a NaN like torch.relu (csrc/srk_common.h, `make exact`); default: one
# instruction per value, a NaN pre-activation becomes 0 (fp32) / keeps its payload only with a clear sign bit (16-bit)
EXACT_RELU = os.environ.get("SRK_EXACT_RELU") == "1"
LIB_PATH = os.environ.get("SRK_LIB_PATH") or os.path.join(_HERE, "libsrk_gfx950_exact.so" if EXACT_RELU else "libsrk_gfx950.so")

SRK_BF16, SRK_F16, SRK_F32 = 0, 1, 2
OUT_NHWC, OUT_NHWC_PS, OUT_PLANAR = 0, 1, 2

_p, _i, _f = C.c_void_p, C.c_int, C.c_float


class PackArgs(C.Structure):
    _fields_ = [("w", _p), ("bias", _p), ("wpk", _p), ("bias_pk", _p),
                ("Cout", _i), ("Cin", _i), ("KH", _i), ("KW", _i),
                ("KinP", _i), ("CoutP", _i), ("dgrad", _i), ("ps_r", _i), ("dtype", _i), ("rows_layout", _i)]


class ConvArgs(C.Structure):
    _fields_ = [("x", _p), ("x_pitch", _i), ("x_coff", _i), ("x_ps", _i),
                ("N", _i), ("H", _i), ("W", _i), ("Cin", _i),
                ("wpk", _p), ("bias", _p), ("CoutP", _i), ("Cout", _i), ("KH", _i), ("KW", _i),
                ("relu", _i), ("scale", _f),
                ("res", _p), ("res_pitch", _i), ("res_coff", _i),
                ("mask", _p), ("mask_pitch", _i), ("mask_coff", _i), ("mask_from", _i),
                ("out", _p), ("out_pitch", _i), ("out_coff", _i), ("out_mode", _i), ("ps_r", _i),
                ("post_add", _p), ("dtype", _i), ("cout_real", _i), ("relu_bits", _p), ("mask_bits", _p)]


class ConvPairArgs(C.Structure):
    _fields_ = [("x", _p), ("x_pitch", _i), ("x_coff", _i), ("N", _i), ("H", _i), ("W", _i),
                ("w1", _p), ("b1", _p), ("w2", _p), ("b2", _p),
                ("relu_mid", _i), ("scale_mid", _f),
                ("mask", _p), ("mask_pitch", _i), ("mask_coff", _i),
                ("mid", _p), ("mid_pitch", _i), ("mid_coff", _i),
                ("scale_out", _f),
                ("res", _p), ("res_pitch", _i), ("res_coff", _i), ("res_from_x", _i),
                ("out", _p), ("out_pitch", _i), ("out_coff", _i), ("dtype", _i),
                ("pool", _p), ("pool_aux", _p), ("pool_aux_pitch", _i), ("pool_aux_coff", _i),
                ("ca_mode", _i), ("ca_cr", _i), ("ca_gsum", _p), ("ca_gsum_rows", _i), ("ca_sums", _p), ("ca_sums_rows", _i),
                ("ca_s", _p), ("ca_z", _p), ("ca_w1", _p), ("ca_w2", _p), ("ca_slots", _p),
                ("xo", _p), ("xo_pitch", _i), ("xo_coff", _i),
                ("ca_x2", _p), ("ca_x2_pitch", _i), ("ca_x2_coff", _i), ("ca_b1", _p), ("ca_b2", _p), ("ca_s_out", _p), ("ca_z_out", _p)]


class PwPackArgs(C.Structure):
    _fields_ = [("w1", _p), ("b1", _p), ("w2", _p), ("b2", _p), ("Cin", _i), ("Chid", _i), ("Cmid", _i), ("CoutP", _i),
                ("fwd", _p), ("bwd", _p), ("dtype", _i)]


class WnJob(C.Structure):
    _fields_ = [("v", _p), ("g", _p), ("w", _p), ("inv", _p), ("dw", _p), ("dv", _p), ("dg", _p),
                ("rows", _i), ("cols", _i), ("row0", _i), ("pad_", _i)]


class PwArgs(C.Structure):
    _fields_ = [("x", _p), ("x_pitch", _i), ("x_coff", _i), ("P", C.c_longlong), ("Cin", _i), ("Chid", _i), ("CoutP", _i), ("Cout", _i),
                ("wpk", _p), ("out", _p), ("out_pitch", _i), ("out_coff", _i), ("dtype", _i)]


class PwBwdArgs(C.Structure):
    _fields_ = [("x", _p), ("x_pitch", _i), ("x_coff", _i), ("gz", _p), ("gz_pitch", _i), ("gz_coff", _i), ("Cz", _i),
                ("P", C.c_longlong), ("Cin", _i), ("Chid", _i), ("CoutP", _i), ("wpk", _p),
                ("res", _p), ("res_pitch", _i), ("res_coff", _i), ("gx", _p), ("gx_pitch", _i), ("gx_coff", _i),
                ("h_out", _p), ("gh_out", _p), ("dtype", _i)]


class PwWgradArgs(C.Structure):
    _fields_ = [("x", _p), ("x_pitch", _i), ("x_coff", _i), ("gz", _p), ("gz_pitch", _i), ("gz_coff", _i), ("Cz", _i),
                ("P", C.c_longlong), ("Cin", _i), ("Chid", _i), ("Cmid", _i), ("CoutP", _i), ("wpk", _p),
                ("dw1p", _p), ("dw2p", _p), ("db1p", _p), ("db2p", _p), ("nranges", _i), ("dw1", _p), ("db1", _p), ("dw2", _p), ("db2", _p),
                ("dtype", _i)]


class ChanFinalizeArgs(C.Structure):
    _fields_ = [("partial", _p), ("nblocks", _i), ("C", _i), ("Creal", _i), ("mode", _i), ("total", _i),
                ("M", _f), ("eps", _f), ("momentum", _f), ("mean", _p), ("invstd", _p), ("gamma", _p),
                ("weight", _p), ("bias", _p), ("running_mean", _p), ("running_var", _p), ("out", _p),
                ("nbt", _p), ("dgamma_acc", _p), ("dbeta_acc", _p), ("partial2", _p), ("total2", _i), ("dslope_acc", _p)]


class RowsumJob(C.Structure):
    _fields_ = [("src", _p), ("dst", _p), ("n", _i), ("k", _i)]


class AdamSlot(C.Structure):
    _fields_ = [("p", _p), ("g", _p), ("state_off", C.c_longlong), ("n", C.c_longlong), ("step_idx", C.c_longlong)]


class AdamBlock(C.Structure):
    _fields_ = [("slot", _i), ("count", _i), ("start", C.c_longlong)]


class AdamArgs(C.Structure):
    _fields_ = [("slots", _p), ("blocks", _p), ("nslots", _i), ("nblocks", _i), ("m", _p), ("v", _p), ("steps", _p), ("ticket", _p),
                ("lr", _f), ("beta1", _f), ("beta2", _f), ("eps", _f), ("weight_decay", _f), ("maximize", _i),
                ("one_minus_beta1", _f), ("one_minus_beta2", _f)]


class WgradArgs(C.Structure):
    _fields_ = [("x", _p), ("x_pitch", _i), ("x_coff", _i), ("x_ps", _i),
                ("dy", _p), ("dy_pitch", _i), ("dy_coff", _i), ("dy_ps", _i),
                ("N", _i), ("H", _i), ("W", _i), ("Cin", _i), ("Cout", _i), ("KH", _i), ("KW", _i),
                ("dwp", _p), ("dbp", _p), ("nslabs", _i), ("dtype", _i), ("cout_real", _i)]


class WgradFinArgs(C.Structure):
    _fields_ = [("dwp", _p), ("dbp", _p), ("nslabs", _i), ("dw", _p), ("db", _p),
                ("Cout", _i), ("Cin", _i), ("KH", _i), ("KW", _i), ("CinP", _i), ("CoutP", _i),
                ("ps_r", _i), ("scale", _f), ("accumulate", _i)]


class ProjArgs(C.Structure):
    _fields_ = [("x", _p), ("x_pitch", _i), ("out", _p), ("out_pitch", _i), ("wpk", _p), ("bias", _p),
                ("N", _i), ("H", _i), ("W", _i), ("dtype", _i), ("slope", _p), ("slope_stride", _i), ("pre", _p), ("pre_pitch", _i)]


class ProjPackJob(C.Structure):
    _fields_ = [("w4", _p), ("wpk", _p)]


class ProjWgradArgs(C.Structure):
    _fields_ = [("xh", _p), ("xh_pitch", _i), ("g", _p), ("g_pitch", _i), ("scratch", _p), ("dw", _p), ("accumulate", _i),
                ("N", _i), ("H", _i), ("W", _i), ("dtype", _i), ("db", _p), ("bias_side", _i), ("db_accumulate", _i)]


class UnfoldArgs(C.Structure):
    _fields_ = [("x", _p), ("sub", _p), ("dst", _p), ("dst_pitch", _i), ("dst_coff", _i),
                ("N", _i), ("Cin", _i), ("H", _i), ("W", _i), ("KH", _i), ("KW", _i), ("Kstore", _i), ("dtype", _i)]


class ToNhwcArgs(C.Structure):
    _fields_ = [("src", _p), ("dst", _p), ("dst_pitch", _i), ("dst_coff", _i),
                ("N", _i), ("C", _i), ("H", _i), ("W", _i), ("Cstore", _i), ("ps_r", _i), ("scale", _f), ("dtype", _i)]


class ToNchwArgs(C.Structure):
    _fields_ = [("src", _p), ("src_pitch", _i), ("src_coff", _i), ("dst", _p),
                ("N", _i), ("C", _i), ("H", _i), ("W", _i), ("dtype", _i)]


class CaPoolArgs(C.Structure):
    _fields_ = [("t", _p), ("t_pitch", _i), ("t_coff", _i), ("u", _p), ("u_pitch", _i), ("u_coff", _i),
                ("sums", _p), ("N", _i), ("HW", _i), ("C", _i), ("dtype", _i)]


class CaApplyArgs(C.Structure):
    _fields_ = [("t", _p), ("t_pitch", _i), ("t_coff", _i), ("res", _p), ("res_pitch", _i), ("res_coff", _i),
                ("sums", _p), ("w1", _p), ("b1", _p), ("w2", _p), ("b2", _p), ("s_out", _p), ("z_out", _p),
                ("out", _p), ("out_pitch", _i), ("out_coff", _i),
                ("N", _i), ("HW", _i), ("C", _i), ("Cr", _i), ("dtype", _i), ("sums_rows", _i)]


class CaBwdArgs(C.Structure):
    _fields_ = [("g", _p), ("g_pitch", _i), ("g_coff", _i), ("gsum", _p), ("sums", _p), ("s", _p), ("z", _p),
                ("w1", _p), ("w2", _p), ("dw1", _p), ("db1", _p), ("dw2", _p), ("db2", _p),
                ("gt", _p), ("gt_pitch", _i), ("gt_coff", _i),
                ("N", _i), ("HW", _i), ("C", _i), ("Cr", _i), ("dtype", _i), ("sums_rows", _i), ("gsum_rows", _i)]


class PatchDesc(C.Structure):
    _fields_ = [("lr", _p), ("hr", _p), ("lr_h", _i), ("lr_w", _i), ("hr_h", _i), ("hr_w", _i),
                ("top", _i), ("left", _i), ("rot", _i), ("hflip", _i), ("vflip", _i)]


class PatchArgs(C.Structure):
    _fields_ = [("table", _p), ("N", _i), ("C", _i), ("patch_lr", _i), ("scale", _i), ("lr_out", _p), ("hr_out", _p)]


class SseArgs(C.Structure):
    _fields_ = [("sr", _p), ("hr", _p), ("N", _i), ("C", _i), ("H", _i), ("W", _i), ("luma", _i), ("shave", _i), ("sse", _p)]


class SsimArgs(C.Structure):
    _fields_ = [("x", _p), ("y", _p), ("N", _i), ("C", _i), ("H", _i), ("W", _i), ("pool", _i),
                ("sigma", _f), ("k1", _f), ("k2", _f), ("sums", _p)]


class L1Args(C.Structure):
    _fields_ = [("sr", _p), ("hr", _p), ("n", C.c_longlong), ("sign", _p), ("partial", _p), ("gout", _p),
                ("scale", _f), ("grad", _p)]


class UnfoldNhwcArgs(C.Structure):
    _fields_ = [("x", _p), ("x_pitch", _i), ("x_coff", _i), ("cols", _p), ("cols_pitch", _i),
                ("N", _i), ("H", _i), ("W", _i), ("C", _i), ("K", _i), ("stride", _i), ("pad", _i), ("Ho", _i), ("Wo", _i), ("dtype", _i)]


class FoldNhwcArgs(C.Structure):
    _fields_ = [("cols", _p), ("cols_pitch", _i), ("bias", _p), ("out", _p), ("out_pitch", _i), ("out_coff", _i),
                ("N", _i), ("Hi", _i), ("Wi", _i), ("C", _i), ("K", _i), ("stride", _i), ("pad", _i), ("Ho", _i), ("Wo", _i), ("dtype", _i)]


class ChanStatsArgs(C.Structure):
    _fields_ = [("x", _p), ("x_pitch", _i), ("x_coff", _i), ("y", _p), ("y_pitch", _i), ("y_coff", _i),
                ("P", C.c_longlong), ("C", _i), ("mode", _i), ("partial", _p), ("dtype", _i), ("shift", _p), ("shift_out", _p),
                ("gate_out", _p), ("gate_pitch", _i), ("slope", _p), ("slope_stride", _i),
                ("gate_a", _p), ("gate_d", _p), ("partial2", _p)]


class ChanApplyArgs(C.Structure):
    _fields_ = [("x", _p), ("x_pitch", _i), ("x_coff", _i), ("y", _p), ("y_pitch", _i), ("y_coff", _i),
                ("z", _p), ("z_pitch", _i), ("z_coff", _i), ("a", _p), ("b", _p), ("d", _p),
                ("slope", _p), ("slope_stride", _i), ("post_prelu", _i), ("out", _p), ("out_pitch", _i), ("out_coff", _i),
                ("P", C.c_longlong), ("C", _i), ("dtype", _i), ("gate_a", _p), ("gate_d", _p)]


class HrTailArgs(C.Structure):
    _fields_ = [("wt", _p), ("bt", _p), ("wu", _p), ("bu", _p), ("O", _i), ("C", _i), ("Ci", _i),
                ("weff", _p), ("beff", _p), ("wedge", _p), ("bedge", _p), ("wcor", _p), ("bcor", _p),
                ("x", _p), ("x_pitch", _i), ("N", _i), ("H", _i), ("W", _i), ("dtype", _i),
                ("out", _p), ("g", _p), ("dx", _p), ("dx_pitch", _i),
                ("eedge", _p), ("e0", _p), ("ecor", _p), ("k0", _p), ("scratch", _p),
                ("r", _p), ("r0", _p), ("dwt", _p), ("dbt", _p), ("dwu", _p), ("dbu", _p)]


# every launcher declared in include/srk.h: name -> argument struct
LAUNCHERS = {
    "srk_pack_conv_weights": PackArgs,
    "srk_conv2d": ConvArgs,
    "srk_conv_pair": ConvPairArgs,
    "srk_pw_pack": PwPackArgs,
    "srk_pw_forward": PwArgs,
    "srk_pw_backward": PwBwdArgs,
    "srk_pw_wgrad": PwWgradArgs,
    "srk_pw_wgrad_partial": PwWgradArgs,
    "srk_adam_step": AdamArgs,
    "srk_chan_finalize": ChanFinalizeArgs,
    "srk_conv2d_wgrad": WgradArgs,
    "srk_wgrad_finalize": WgradFinArgs,
    "srk_unfold_nchw": UnfoldArgs,
    "srk_nchw_to_nhwc": ToNhwcArgs,
    "srk_nhwc_to_nchw": ToNchwArgs,
    "srk_ca_pool": CaPoolArgs,
    "srk_ca_apply": CaApplyArgs,
    "srk_ca_bwd_apply": CaBwdArgs,
    "srk_sample_patches": PatchArgs,
    "srk_image_sse": SseArgs,
    "srk_image_ssim": SsimArgs,
    "srk_l1_loss_fwd": L1Args,
    "srk_l1_loss_bwd": L1Args,
    "srk_unfold_nhwc": UnfoldNhwcArgs,
    "srk_fold_nhwc": FoldNhwcArgs,
    "srk_chan_stats": ChanStatsArgs,
    "srk_chan_apply": ChanApplyArgs,
    "srk_proj_down": ProjArgs,
    "srk_proj_up": ProjArgs,
    "srk_proj_wgrad": ProjWgradArgs,
    "srk_hrtail_collapse": HrTailArgs,
    "srk_hrtail_edge_fwd": HrTailArgs,
    "srk_hrtail_edge_bwd_x": HrTailArgs,
    "srk_hrtail_edge_bwd_w": HrTailArgs,
    "srk_hrtail_expand": HrTailArgs,
}
OTHER_SYMBOLS = ("srk_conv_tile", "srk_last_error", "srk_version", "srk_device_cus", "srk_wgrad_slabs",
                 "srk_pack_conv_weights_group", "srk_l1_blocks", "srk_wgrad_group_ok", "srk_wgrad_group_job_bytes",
                 "srk_wgrad_group_plan", "srk_conv2d_wgrad_group", "srk_wgrad_finalize_group", "srk_upload_small", "srk_upload_prepare", "srk_upload_eager", "srk_upload_fence", "srk_ca_splits", "srk_chan_stats_blocks",
                 "srk_conv_pair_tiles", "srk_rowsum_group", "srk_pw_shape_ok", "srk_pw_pack_bytes", "srk_pw_pack_group", "srk_weight_norm_group", "srk_pw_wgrad_ranges", "srk_l1_loss_mean", "srk_chan_stats_finalize", "srk_pack_group_tiles", "srk_pack_conv_weights_group_tiled",
                 "srk_proj_pack", "srk_proj_pack_bytes", "srk_proj_wgrad_scratch_floats", "srk_proj_pack_group", "srk_wgrad_slab_cout",
                 "srk_hrtail_scratch_floats", "srk_pw_wgrad_finalize_group", "srk_adam_step_scaled", "srk_adam_check_scaled", "srk_adam_update_scaled", "srk_loss_scale_update", "srk_conv_bits_ok",
                 "srk_conv_trunk", "srk_conv_trunk_ok")

_lib = None


def load():
    """Load the library once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the HIP path.")
    lib = C.CDLL(LIB_PATH)
    # A/B timing against an OLDER build of this library (tools/ab_lib.sh: SRK_LIB_PATH=tools/ubench/libsrk_prev.so): entry points that
    # build does not have yet are left unbound (calling one raises).  The product library (no SRK_LIB_PATH) must export everything.
    older = bool(os.environ.get("SRK_LIB_PATH"))

    class _Absent:
        def __init__(self, name):
            self.name = name
            self.argtypes = self.restype = None

        def __call__(self, *a, **k):
            raise RuntimeError(f"{self.name} is not exported by {LIB_PATH} (an older build loaded through SRK_LIB_PATH)")

    if older:
        for name in list(LAUNCHERS) + list(OTHER_SYMBOLS):
            if not hasattr(lib, name):
                setattr(lib, name, _Absent(name))
    for name, st in LAUNCHERS.items():
        fn = getattr(lib, name)
        fn.argtypes = [C.POINTER(st), C.c_void_p]
        fn.restype = C.c_int
    lib.srk_conv_tile.argtypes = [C.c_int]
    lib.srk_conv_tile.restype = C.c_int
    lib.srk_conv_pair_tiles.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.srk_conv_pair_tiles.restype = C.c_int
    lib.srk_wgrad_slabs.argtypes = [C.POINTER(WgradArgs)]
    lib.srk_wgrad_slabs.restype = C.c_int
    lib.srk_wgrad_slab_cout.argtypes = [C.POINTER(WgradArgs)]
    lib.srk_wgrad_slab_cout.restype = C.c_int
    lib.srk_pack_conv_weights_group.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.srk_pack_conv_weights_group.restype = C.c_int
    lib.srk_pack_group_tiles.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.srk_pack_group_tiles.restype = C.c_int
    lib.srk_pack_conv_weights_group_tiled.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.srk_pack_conv_weights_group_tiled.restype = C.c_int
    lib.srk_wgrad_group_ok.argtypes = [C.POINTER(WgradArgs)]
    lib.srk_wgrad_group_ok.restype = C.c_int
    lib.srk_wgrad_group_job_bytes.restype = C.c_int
    lib.srk_wgrad_group_plan.argtypes = [C.POINTER(WgradArgs), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(C.c_int), C.POINTER(C.c_longlong)]
    lib.srk_wgrad_group_plan.restype = C.c_int
    lib.srk_conv2d_wgrad_group.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.srk_conv2d_wgrad_group.restype = C.c_int
    lib.srk_wgrad_finalize_group.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.srk_wgrad_finalize_group.restype = C.c_int
    lib.srk_rowsum_group.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.srk_rowsum_group.restype = C.c_int
    lib.srk_upload_small.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
    lib.srk_upload_small.restype = C.c_int
    if not isinstance(getattr(lib, "srk_upload_eager", None), _Absent):
        lib.srk_upload_prepare.argtypes = []
        lib.srk_upload_prepare.restype = C.c_int
        lib.srk_upload_eager.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
        lib.srk_upload_eager.restype = C.c_int
        lib.srk_upload_fence.argtypes = []
        lib.srk_upload_fence.restype = C.c_int
    lib.srk_chan_stats_finalize.argtypes = [C.POINTER(ChanStatsArgs), C.POINTER(ChanFinalizeArgs), C.c_void_p, C.c_void_p]
    lib.srk_chan_stats_finalize.restype = C.c_int
    lib.srk_chan_stats_blocks.argtypes = [C.c_longlong]
    lib.srk_chan_stats_blocks.restype = C.c_int
    lib.srk_ca_splits.argtypes = [C.c_int, C.c_int]
    lib.srk_ca_splits.restype = C.c_int
    lib.srk_l1_loss_mean.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]
    lib.srk_l1_loss_mean.restype = C.c_int
    lib.srk_l1_blocks.argtypes = [C.c_longlong]
    lib.srk_l1_blocks.restype = C.c_int
    lib.srk_pw_shape_ok.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.srk_pw_shape_ok.restype = C.c_int
    lib.srk_pw_pack_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    lib.srk_pw_pack_bytes.restype = C.c_longlong
    lib.srk_pw_wgrad_ranges.argtypes = [C.c_longlong, C.c_int]
    lib.srk_pw_wgrad_ranges.restype = C.c_int
    lib.srk_pw_pack_group.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.srk_pw_pack_group.restype = C.c_int
    lib.srk_weight_norm_group.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.srk_weight_norm_group.restype = C.c_int
    lib.srk_proj_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.srk_proj_pack.restype = C.c_int
    lib.srk_proj_pack_group.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.srk_proj_pack_group.restype = C.c_int
    lib.srk_proj_pack_bytes.restype = C.c_longlong
    lib.srk_proj_wgrad_scratch_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.srk_proj_wgrad_scratch_floats.restype = C.c_longlong
    lib.srk_hrtail_scratch_floats.argtypes = [C.c_int, C.c_int]
    lib.srk_hrtail_scratch_floats.restype = C.c_longlong
    lib.srk_conv_bits_ok.argtypes = [C.POINTER(ConvArgs)]
    lib.srk_conv_bits_ok.restype = C.c_int
    lib.srk_conv_trunk_ok.argtypes = [C.POINTER(ConvArgs), C.c_int]
    lib.srk_conv_trunk_ok.restype = C.c_int
    lib.srk_conv_trunk.argtypes = [C.POINTER(ConvArgs), C.c_void_p, C.c_int, C.c_void_p]
    lib.srk_conv_trunk.restype = C.c_int
    lib.srk_adam_step_scaled.argtypes = [C.POINTER(AdamArgs), C.c_void_p, C.c_void_p]
    lib.srk_adam_step_scaled.restype = C.c_int
    for fn in (lib.srk_adam_check_scaled, lib.srk_adam_update_scaled):
        fn.argtypes = [C.POINTER(AdamArgs), C.c_void_p, C.c_void_p]
        fn.restype = C.c_int
    lib.srk_pw_wgrad_finalize_group.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.srk_pw_wgrad_finalize_group.restype = C.c_int
    lib.srk_loss_scale_update.argtypes = [C.c_void_p, C.c_void_p]
    lib.srk_loss_scale_update.restype = C.c_int
    lib.srk_last_error.restype = C.c_char_p
    lib.srk_version.restype = C.c_int
    lib.srk_device_cus.restype = C.c_int
    _lib = lib
    return lib


def call(name, args, stream):
    """Invoke launcher `name`; raise RuntimeError with the library's message on failure."""
    lib = load()
    if getattr(args, "N", 1) == 0:
        return          # empty batch: nothing to launch (torch's convs accept it; outputs are empty, sums stay zero)
    rc = getattr(lib, name)(C.byref(args), C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {lib.srk_last_error().decode(errors='replace')}")


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {load().srk_last_error().decode(errors='replace')}")


def conv_tile(cout):
    return load().srk_conv_tile(int(cout))
