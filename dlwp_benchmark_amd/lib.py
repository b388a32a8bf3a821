"""ctypes binding of libdlwp_hip.so (the C ABI declared in include/dlwp_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C dlwp_benchmark_amd/csrc`.
There is NO fallback: if the shared object is missing or a symbol is absent, importing the
compute path raises -- the product never routes around the HIP kernels.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DLWP_HIP_LIB") or os.path.join(_HERE, "libdlwp_hip.so")   # override: A/B builds of the same ABI

c_float_p = POINTER(c_float)
c_int32_p = POINTER(c_int32)


class FNO2dDesc(ctypes.Structure):
    """mirror of struct dlwp_fno2d_desc (include/dlwp_hip.h)"""
    _fields_ = [
        ("in_channels", c_int32), ("hidden_channels", c_int32), ("lifting_channels", c_int32),
        ("projection_channels", c_int32), ("out_channels", c_int32), ("n_layers", c_int32),
        ("height", c_int32), ("width", c_int32), ("n_rows", c_int32), ("n_cols", c_int32),
        ("rows_in", c_int32_p), ("rows_out", c_int32_p),
        ("fwd_scale", c_float), ("inv_scale", c_float),
        ("lift_w1", c_void_p), ("lift_b1", c_void_p), ("lift_w2", c_void_p), ("lift_b2", c_void_p),
        ("spec_w", POINTER(c_void_p)), ("spec_b", c_void_p), ("skip_w", POINTER(c_void_p)),
        ("proj_w1", c_void_p), ("proj_b1", c_void_p), ("proj_w2", c_void_p), ("proj_b2", c_void_p),
        ("precision_form", c_int32), ("launch_form", c_int32), ("on_timeout", c_int32), ("unchecked", c_int32),
        ("debug_spin_limit", c_int32),
    ]


class WAttnDesc(ctypes.Structure):
    """mirror of struct dlwp_wattn_desc (include/dlwp_hip.h)"""
    _fields_ = [
        ("grid", c_int32 * 3), ("padded", c_int32 * 3), ("pad_lead", c_int32 * 3), ("window", c_int32 * 3),
        ("shift_fwd", c_int32 * 3), ("shift_back", c_int32 * 3), ("use_mask", c_int32),
        ("mask_b1", c_int32 * 3), ("mask_b2", c_int32 * 3), ("bias_mode", c_int32),
        ("heads", c_int32), ("head_dim", c_int32), ("scale", c_float), ("form", c_int32),
    ]


# name -> (restype, argtypes); every symbol include/dlwp_hip.h declares
SIGNATURES = {
    "dlwp_version": (c_int32, []),
    "dlwp_last_error": (c_char_p, []),
    "dlwp_device_count": (c_int32, []),
    "dlwp_fno2d_plan_create": (c_int32, [POINTER(c_void_p), POINTER(FNO2dDesc), c_void_p]),
    "dlwp_fno2d_plan_destroy": (c_int32, [c_void_p]),
    "dlwp_fno2d_workspace_bytes": (c_size_t, [c_void_p, c_int32]),
    "dlwp_fno2d_status": (c_int32, [c_void_p, c_void_p]),
    "dlwp_fno2d_timeouts": (ctypes.c_uint32, [c_void_p]),
    "dlwp_fno2d_range_reruns": (ctypes.c_uint32, [c_void_p]),
    "dlwp_fno2d_forward_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_size_t, c_void_p]),
    "dlwp_fno2d_rollout_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32,
                                         c_int32, c_int32, c_int32, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dlwp_fno2d_rollout_range_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32,
                                               c_int32, c_int32, c_int32, c_void_p, c_void_p, c_size_t, c_void_p,
                                               c_int32, c_int32]),
    "dlwp_fno2d_rollout_profiled_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32,
                                                  c_int32, c_int32, c_int32, c_void_p, c_void_p, c_size_t, c_void_p,
                                                  POINTER(ctypes.c_double), c_int32_p]),
    "dlwp_spectral_conv2d_plan_create": (c_int32, [POINTER(c_void_p), c_int32, c_int32, c_int32, c_int32,
                                                   c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "dlwp_spectral_conv2d_plan_create_ex": (c_int32, [POINTER(c_void_p), c_int32, c_int32, c_int32, c_int32, c_int32,
                                                      c_int32, c_void_p, c_void_p, c_float, c_float, c_void_p]),
    "dlwp_spectral_conv2d_set_weights_dev": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p]),
    "dlwp_spectral_conv2d_plan_destroy": (c_int32, [c_void_p]),
    "dlwp_spectral_conv2d_workspace_bytes": (c_size_t, [c_void_p, c_int32]),
    "dlwp_window_attn_workspace_bytes": (c_size_t, [POINTER(WAttnDesc), c_int32, c_int32]),
    "dlwp_window_attn_fallbacks": (c_int32, [POINTER(WAttnDesc), c_int32, c_int32, c_void_p, c_void_p, c_int32_p]),
    "dlwp_window_attn_f32": (c_int32, [POINTER(WAttnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                       c_size_t, c_void_p]),
    "dlwp_window_attn_bf16": (c_int32, [POINTER(WAttnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                        c_size_t, c_void_p]),
    "dlwp_window_attn_bf16_io": (c_int32, [POINTER(WAttnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                           c_size_t, c_void_p]),
    "dlwp_window_attn_bwd_workspace_bytes": (c_size_t, [POINTER(WAttnDesc), c_int32]),
    "dlwp_window_attn_bwd_f32": (c_int32, [POINTER(WAttnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_int32, c_void_p, c_size_t, c_void_p]),
    "dlwp_afno2d_mix_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                      c_int32, c_int32, c_int32, c_float, c_float, c_void_p]),
    "dlwp_afno2d_mix_scaled_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                             c_int32, c_int32, c_int32, c_float, c_float, c_float, c_float, c_void_p]),
    "dlwp_afno2d_mix_bwd_f32": (c_int32, [c_void_p] * 11 + [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_float,
                                          c_float, c_float, c_void_p]),
    "dlwp_fft2_plan_create": (c_int32, [ctypes.POINTER(c_void_p), c_int32, c_int32, c_int32]),
    "dlwp_fft2_plan_destroy": (c_int32, [c_void_p]),
    "dlwp_rfft2_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "dlwp_irfft2_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "dlwp_afno_fft_supported": (c_int32, [c_int32, c_int32, c_int32]),
    "dlwp_afno_fft_plan_create": (c_int32, [ctypes.POINTER(c_void_p), c_int32, c_int32, c_int32, c_void_p]),
    "dlwp_afno_fft_plan_destroy": (c_int32, [c_void_p]),
    "dlwp_afno_rfft2_kept_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "dlwp_afno_irfft2_kept_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "dlwp_conv3x3_cyl_f32": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32,
                                       c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "dlwp_conv3x3_ex_f32": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                      c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "dlwp_linear_packed_bytes": (c_size_t, [c_int32, c_int32]),
    "dlwp_linear_pack_f32": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "dlwp_linear_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int32, c_int32, c_int32,
                                  c_void_p]),
    "dlwp_linear_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int32, c_int32, c_int32,
                                   c_void_p]),
    "dlwp_linear_bf16_io": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int32, c_int32, c_int32,
                                      c_int32, c_int32, c_void_p]),
    "dlwp_linear_pack_f16x3": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "dlwp_linear_f16x3": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int32, c_int32, c_int32,
                                    c_void_p]),
    "dlwp_groupnorm_act_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float,
                                         c_int32, c_void_p]),
    "dlwp_conv2d_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                  c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "dlwp_conv_transpose2d_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                            c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "dlwp_avgpool2x2_f32": (c_int32, [c_void_p, c_void_p, ctypes.c_int64, c_int32, c_int32, c_void_p]),
    "dlwp_conv3x3_hpx_f32": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32,
                                       c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "dlwp_healpix_pad_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "dlwp_convlstm_gates_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                          c_void_p]),
    "dlwp_layernorm_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int32, c_float, c_void_p]),
    "dlwp_layernorm_prebias_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int32, c_float,
                                             c_void_p]),
    "dlwp_layernorm_prebias_bf16out": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int32, c_float,
                                                 c_void_p]),
    "dlwp_layernorm_nhwc_to_nchw_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, ctypes.c_int64, c_int32,
                                                  c_float, c_void_p]),
    "dlwp_afno_merge_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                      ctypes.c_int64, c_int32, c_float, c_void_p]),
    "dlwp_patch_embed_1x1_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, ctypes.c_int64,
                                           c_int32, c_void_p]),
    "dlwp_concat_channels_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, ctypes.c_int64, c_void_p]),
    "dlwp_patch_recover_1x1_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, ctypes.c_int64, c_int32, c_int32, c_void_p]),
    "dlwp_token_mlp_packed_bytes": (c_size_t, [c_int32, c_int32]),
    "dlwp_token_mlp_pack_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p,
                                          c_void_p]),
    "dlwp_afno_block_tail_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, ctypes.c_int64,
                                           c_int32, c_int32, c_float, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "dlwp_token_mlp_pack_f16x3": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p,
                                            c_void_p]),
    "dlwp_afno_block_tail_f16x3": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, ctypes.c_int64,
                                             c_int32, c_int32, c_float, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "dlwp_token_mlp_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int32,
                                     c_int32, c_float, c_void_p]),
    "dlwp_token_mlp_emit_norm_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64,
                                               c_int32, c_int32, c_float, c_void_p, c_void_p, c_float, c_void_p,
                                               ctypes.c_int64, c_void_p]),
    "dlwp_weighted_error_sums_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                               c_int32, c_int32, c_int32, c_void_p]),
    "dlwp_weighted_error_sums_acc_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                                   c_int32, c_int32, c_int32, c_void_p]),
    "dlwp_spectral_conv2d_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_size_t, c_void_p]),
}

_lib = None


class DlwpError(RuntimeError):
    pass


def load():
    """Loads libdlwp_hip.so and types every entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DlwpError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C dlwp_benchmark_amd/csrc`.  There is no CPU fallback for the HIP path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise DlwpError(f"libdlwp_hip.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().dlwp_last_error().decode(errors="replace")
        raise DlwpError(f"{what or 'libdlwp_hip'} failed with status {rc}: {msg}")


def require_cuda_tensor(t, name: str):
    import torch

    if t is None:
        return
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise DlwpError(
            f"{name} must be a tensor on an MI355X device (got {'cpu tensor' if isinstance(t, torch.Tensor) else type(t)}); "
            "the HIP path has no CPU fallback -- the CPU restatement lives in oracle/ and is test-only")
    if t.dtype != torch.float32:
        raise DlwpError(f"{name} must be float32 (got {t.dtype})")


def stream_ptr():
    import torch

    return c_void_p(torch.cuda.current_stream().cuda_stream)


class KernelTimer:
    """Measurement aid (bench.py): brackets every call of the chosen C-ABI entry points with HIP events on the stream
    the call launches on (torch's current stream = the `stream` argument every wrapper passes) and sums the elapsed
    time per (entry point, tag).  `tagger(name, args) -> hashable` separates calls of one entry point by shape.
    An EMPTY bracket is recorded beside every call; half of its mean (one marker latency) is subtracted per call,
    which reproduces rocprofv3 --kernel-trace averages within a few per cent (bench.py).
    The product path never uses this class."""

    def __init__(self, names=None, tagger=None):
        self.names = names
        self.tagger = tagger
        self.records = []
        self._empty = []
        self._saved = {}

    def __enter__(self):
        import torch

        lib = load()
        names = self.names or [n for n in SIGNATURES if n.endswith(("_f32", "_bf16", "_f16x3", "_bf16_io", "_bf16out"))]
        for name in names:
            fn = getattr(lib, name)
            self._saved[name] = fn

            def wrapped(*a, _fn=fn, _name=name):
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
                rc = _fn(*a)
                e1.record()
                e2.record()
                self.records.append((_name, self.tagger(_name, a) if self.tagger else None, e0, e1))
                self._empty.append((e1, e2))
                return rc

            setattr(lib, name, wrapped)
        return self

    def __exit__(self, *exc):
        lib = load()
        for name, fn in self._saved.items():
            setattr(lib, name, fn)
        self._saved = {}
        return False

    def summary(self):
        """{(name, tag): {"calls": n, "total_ms": t, "avg_ms": t / n}} after a device synchronisation."""
        import torch

        torch.cuda.synchronize()
        marker = 0.0
        if self._empty:
            marker = 0.5 * sum(a.elapsed_time(b) for a, b in self._empty) / len(self._empty)
        out = {}
        for name, tag, e0, e1 in self.records:
            d = out.setdefault((name, tag), {"calls": 0, "total_ms": 0.0})
            d["calls"] += 1
            d["total_ms"] += max(e0.elapsed_time(e1) - marker, 0.0)
        for d in out.values():
            d["avg_ms"] = d["total_ms"] / d["calls"]
        return out, marker
