"""ctypes binding of libgpnerf_hip.so (C-ABI declared in include/gpnerf_hip.h).

There is no fallback: if the library is missing or a tensor is not a contiguous fp32 HIP tensor the
call raises.  Tensors stay owned by PyTorch; kernels are enqueued on torch's current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPN_LIB: developer override used to A/B kernel build variants; the default is the in-tree library
LIB_PATH = os.environ.get("GPN_LIB") or os.path.join(_HERE, "csrc", "libgpnerf_hip.so")

ABI_VERSION = 3
MAX_DEPTH = 8
MAX_VIEWS = 32
MAX_SAMPLES = 128
EMB = 128
SEM = 512
NETWIDTH = 64
VIEW_FOLD_FLOATS = 64 * 64 + 16 * 96 + 16
STATIC_WS_BYTES = 64 << 20  # GPN_STATIC_WS_BYTES

_f = C.POINTER(C.c_float)


class Linear(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p)]


class ViewLayer(C.Structure):
    _fields_ = [("attn_norm_w", C.c_void_p), ("attn_norm_b", C.c_void_p),
                ("ff_norm_w", C.c_void_p), ("ff_norm_b", C.c_void_p),
                ("ff1", Linear), ("ff2", Linear),
                ("q", Linear), ("k", Linear), ("v", Linear),
                ("pos0", Linear), ("pos2", Linear),
                ("attn0", Linear), ("attn2", Linear),
                ("out", Linear), ("folded", C.c_void_p)]


class RayLayer(C.Structure):
    _fields_ = [("attn_norm_w", C.c_void_p), ("attn_norm_b", C.c_void_p),
                ("ff_norm_w", C.c_void_p), ("ff_norm_b", C.c_void_p),
                ("ff1", Linear), ("ff2", Linear),
                ("q", Linear), ("k", Linear), ("v", Linear),
                ("out", Linear)]


class GntWeights(C.Structure):
    _fields_ = [("trans_depth", C.c_int), ("in_feat_ch", C.c_int),
                ("precision", C.c_int), ("reserved_", C.c_int),
                ("rgbfeat0", Linear), ("rgbfeat2", Linear),
                ("view", ViewLayer * MAX_DEPTH), ("ray", RayLayer * MAX_DEPTH),
                ("qfc0", Linear * MAX_DEPTH), ("qfc2", Linear * MAX_DEPTH),
                ("norm_w", C.c_void_p), ("norm_b", C.c_void_p),
                ("rgb_fc", Linear), ("sem_adaptor", Linear)]


class ProjectArgs(C.Structure):
    _fields_ = [("xyz", C.c_void_p), ("proj", C.c_void_p), ("src_pos", C.c_void_p),
                ("qry_pos", C.c_float * 3), ("img_h", C.c_float), ("img_w", C.c_float),
                ("N", C.c_int), ("V", C.c_int),
                ("rgb", C.c_void_p), ("rgb_h", C.c_int), ("rgb_w", C.c_int),
                ("feat", C.c_void_p), ("feat_h", C.c_int), ("feat_w", C.c_int), ("feat_c", C.c_int),
                ("deep", C.c_void_p), ("deep_h", C.c_int), ("deep_w", C.c_int), ("deep_c", C.c_int),
                ("rgb_feat", C.c_void_p), ("deep_out", C.c_void_p), ("ray_diff", C.c_void_p),
                ("mask", C.c_void_p), ("pix", C.c_void_p), ("taps", C.c_void_p),
                ("front", C.c_void_p), ("rgb_feat_ld", C.c_int), ("samples_per_ray", C.c_int)]


class SemArgs(C.Structure):
    _fields_ = [("R", C.c_int), ("S", C.c_int), ("V", C.c_int), ("depth", C.c_int),
                ("hidden", C.c_void_p), ("stats", C.c_void_p), ("abar", C.c_void_p),
                ("mask", C.c_void_p),
                ("deep_dense", C.c_void_p),
                ("deep_map", C.c_void_p), ("deep_h", C.c_int), ("deep_w", C.c_int),
                ("xyz", C.c_void_p), ("proj", C.c_void_p),
                ("img_h", C.c_float), ("img_w", C.c_float),
                ("sem3d", C.c_void_p), ("pre", C.c_void_p),
                ("attn2_w", C.c_void_p * MAX_DEPTH), ("attn2_b", C.c_void_p * MAX_DEPTH),
                ("ray_row_width", C.c_int), ("ray_row_offset", C.c_int)]


_P = C.c_void_p
_SIGS = {
    "gpn_abi_version": (C.c_int, []),
    "gpn_last_error": (C.c_char_p, []),
    "gpn_sample_along_ray": (C.c_int, [_P, _P, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, _P,
                                       _P, _P, _P]),
    "gpn_sample_fine": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  _P, _P, _P]),
    "gpn_sample_pdf": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "gpn_raw2outputs": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]),
    "gpn_rays_single_image": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_longlong, _P, _P, _P]),
    "gpn_argmax_labels": (C.c_int, [_P, C.c_int, C.c_longlong, _P, _P, _P]),
    "gpn_repack_nhwc": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "gpn_project_gather": (C.c_int, [C.POINTER(ProjectArgs), _P]),
    "gpn_project_rgbfeat": (C.c_int, [C.POINTER(ProjectArgs), C.POINTER(Linear), C.POINTER(Linear), _P, _P,
                                      C.c_int, _P]),
    "gpn_posenc": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "gpn_rgbfeat": (C.c_int, [C.POINTER(Linear), C.POINTER(Linear), _P, C.c_int, C.c_int, C.c_int,
                              C.c_int, _P, _P, _P, C.c_int, _P]),
    "gpn_view_attn": (C.c_int, [C.POINTER(ViewLayer), _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P,
                                C.c_int, _P, C.c_int, C.c_int, _P]),
    "gpn_view_fold": (C.c_int, [C.POINTER(ViewLayer), _P, _P]),
    "gpn_vis_order_scratch_bytes": (C.c_size_t, []),
    "gpn_vis_order": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P]),
    "gpn_ff": (C.c_int, [_P, _P, C.POINTER(Linear), C.POINTER(Linear), _P, C.c_int, _P, C.c_int, _P]),
    "gpn_qpos": (C.c_int, [C.POINTER(Linear), C.POINTER(Linear), _P, _P, _P, C.c_int, C.c_int, _P,
                           C.c_int, _P]),
    "gpn_ray_attn": (C.c_int, [C.POINTER(RayLayer), _P, C.c_int, C.c_int, _P, _P, C.c_int, _P,
                               C.c_int, _P]),
    "gpn_sem_aggregate": (C.c_int, [C.POINTER(SemArgs), _P]),
    "gpn_heads": (C.c_int, [C.POINTER(GntWeights), _P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P,
                            _P]),
    "gpn_gnt_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "gpn_gnt_forward": (C.c_int, [C.POINTER(GntWeights), _P, C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int,
                                  C.c_int, C.POINTER(SemArgs), _P, _P, _P, _P, C.c_size_t, _P]),
    "gpn_render_pass": (C.c_int, [C.POINTER(GntWeights), C.POINTER(ProjectArgs), _P, _P, _P, C.c_int, C.c_int,
                                  C.POINTER(SemArgs), _P, _P, _P, _P, C.c_size_t, _P]),
    "gpn_gnt_forward_static": (C.c_int, [C.POINTER(GntWeights), _P, C.c_int, _P, _P, _P, _P, _P, C.c_int,
                                         C.c_int, C.c_int, C.POINTER(SemArgs), _P, _P, _P, _P]),
    "gpn_timing_begin": (C.c_int, [C.c_uint, C.c_int]),
    "gpn_timing_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
}

STAGES = ("sample", "project", "project_deep", "posenc", "rgbfeat", "view_attn", "ff", "qpos",
          "ray_attn", "sem", "heads", "vis_order", "proj_rgbfeat")

EXPORTS = tuple(_SIGS)
_lib = None


def lib():
    """Load the shared library once; raises if it is not built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "gp-nerf_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g;"
                " g.build()'` (there is no CPU fallback)" % LIB_PATH)
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        if h.gpn_abi_version() != ABI_VERSION:
            raise RuntimeError("gp-nerf_amd: ABI version mismatch in " + LIB_PATH)
        _lib = h
    return _lib


def check(code, what):
    if code != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, code, lib().gpn_last_error().decode()))


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev_f32(t, name, shape=None):
    """Validate a device tensor handed to a kernel: fp32, contiguous, on a HIP device."""
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s: expected a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s: tensor is on %s; gp-nerf_amd runs on HIP devices only (no CPU "
                           "fallback)" % (name, t.device))
    if t.dtype != torch.float32:
        raise ValueError("%s: expected float32, got %s" % (name, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s: expected a contiguous tensor" % name)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s: expected shape %s, got %s" % (name, tuple(shape), tuple(t.shape)))
    return t


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def as_f32c(t):
    """float32 + contiguous view/copy of a device tensor (API inputs may be permuted views)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


class stage_timing:
    """Context manager around gpn_timing_begin / gpn_timing_end: per-stage kernel time measured
    with HIP events on the launch stream.  `.ms` / `.launches` map stage name -> totals."""

    def __init__(self, stages=STAGES, max_records=8192):
        self.mask = sum(1 << STAGES.index(s) for s in stages)
        self.max_records = max_records
        self.ms, self.launches = {}, {}

    def __enter__(self):
        check(lib().gpn_timing_begin(self.mask, self.max_records), "gpn_timing_begin")
        return self

    def __exit__(self, *exc):
        ms = (C.c_double * len(STAGES))()
        cnt = (C.c_longlong * len(STAGES))()
        check(lib().gpn_timing_end(ms, cnt), "gpn_timing_end")
        self.ms = {s: ms[i] for i, s in enumerate(STAGES)}
        self.launches = {s: cnt[i] for i, s in enumerate(STAGES)}
        return False


PRECISIONS = {"fp32": 0, "bf16x3": 1, "bf16": 2}  # GPN_PREC_* of include/gpnerf_hip.h


def precision_code(mode):
    """'fp32' (exact fp32 MFMA), 'bf16x3' (split-bf16 MFMA, fp32 accumulate) or 'bf16' (plain bf16 operands,
    one MFMA per product, fp32 accumulate: looser tolerance) -> GPN_PREC_*.  The mode
    is an argument of every call (ABI 2): there is no process-wide switch."""
    try:
        return PRECISIONS[mode]
    except KeyError:
        raise ValueError("precision must be one of %s (got %r)" % (sorted(PRECISIONS), mode)) from None
