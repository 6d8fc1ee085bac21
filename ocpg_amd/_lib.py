"""ctypes loader for libocpg_hip.so -- the only way the product reaches its kernels.

There is deliberately NO fallback: if the library is missing or a tensor is not on the GPU the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# OCPG_HIP_LIB: load another build of the SAME library (kernel experiments / A-B timing); never a fallback path
LIB_PATH = os.environ.get("OCPG_HIP_LIB") or os.path.join(_HERE, "lib", "libocpg_hip.so")
_lib = None

_i64p = ctypes.c_void_p
_vp = ctypes.c_void_p
_int = ctypes.c_int

# symbol -> argtypes; must list every entry point declared in include/ocpg_hip.h
SIGNATURES = {
    "ocpg_msda_fwd_f32": [_vp, _i64p, _i64p, _vp, _vp] + [_int] * 7 + [_vp, _vp, _vp],
    "ocpg_msda_fwd_f64": [_vp, _i64p, _i64p, _vp, _vp] + [_int] * 7 + [_vp, _vp],
    "ocpg_msda_bwd_f32": [_vp, _i64p, _i64p, _vp, _vp, _vp] + [_int] * 7 + [_vp, _vp, _vp, _vp, _vp],
    "ocpg_msda_bwd_value_f32": [_vp, _vp, _vp] + [_int] * 7 + [_vp, _vp, _vp],
    "ocpg_msda_fused_fwd_f32": [_vp, _i64p, _i64p, _vp, _vp] + [_int] * 7 + [_vp, _vp, _vp, _vp],
    "ocpg_msda_fused_bwd_qproj_f32": [_vp, _i64p, _i64p, _vp, _vp, _vp] + [_int] * 7 + [_vp, _vp],
    "ocpg_msda_bwd_value_sel_f32": [_vp, _vp, _vp] + [_int] * 7 + [_vp, _vp, _vp, _vp],
    "ocpg_msda_bwd_locattn_f32": [_vp, _i64p, _i64p, _vp, _vp, _vp] + [_int] * 7 + [_vp, _vp, _vp],
    "ocpg_msda_bwd_f64": [_vp, _i64p, _i64p, _vp, _vp, _vp] + [_int] * 7 + [_vp, _vp, _vp, _vp],
    "ocpg_bn_act_fwd": [_vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _int, ctypes.c_longlong, _int, _int, _vp],
    "ocpg_bn_act_bwd": [_vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _int, ctypes.c_longlong, _int, _int, _vp],
    "ocpg_dynmask_fwd_f32": [_vp, _vp, _vp] + [_int] * 6 + [_vp, _vp, _vp],
    "ocpg_dynmask_bwd_pre_f32": [_vp, _vp, _vp] + [_int] * 6 + [_vp, _vp, _vp, _vp],
    "ocpg_dynmask_bwd_fin_f32": [_vp, _vp, _vp, _vp] + [_int] * 5 + [_vp, _vp, _vp],
    "ocpg_conv3x3_mfma_fwd": [_vp, _vp, _vp, _vp] + [_int] * 7 + [_vp, _vp],
    "ocpg_conv3x3_mfma_fwd_cols": [_vp, _vp, _vp, _vp] + [_int] * 7 + [_vp, _vp, _vp],
    "ocpg_conv3x3_mfma_dgrad": [_vp, _vp] + [_int] * 6 + [_vp, _vp],
    "ocpg_conv3x3_mfma_dgrad_masked": [_vp, _vp, _vp, _vp] + [_int] * 6 + [_vp, _vp],
    "ocpg_conv3x3_mfma_dgrad_w": [_vp, _vp, _vp, _vp] + [_int] * 6 + [_vp, _vp],
    "ocpg_conv3x3_mfma_splits": [_int] * 6,
    "ocpg_conv3x3_mfma_wgrad_splits": [_int] * 6,
    "ocpg_conv3x3_mfma_body_splits": [ctypes.c_longlong, _int, _int],
    "ocpg_conv3x3_mfma_fwd_bn_splitk": [_vp, _vp, _vp, _vp] + [_int] * 8 + [_vp, _vp, _vp],
    "ocpg_conv3x3_mfma_dgrad_w_splitk": [_vp, _vp, _vp, _vp] + [_int] * 7 + [_vp, _vp, _vp],
    "ocpg_conv3x3_mfma_wgrad": [_vp, _vp] + [_int] * 6 + [_vp, _vp],
    "ocpg_conv3x3_mfma_fwd_splitk": [_vp, _vp, _vp] + [_int] * 7 + [_vp, _vp, _int, _vp, _vp],
    "ocpg_gemm": [_vp, _vp, _vp, _vp] + [_int] * 4 + [ctypes.c_longlong] * 10 + [ctypes.c_float, ctypes.c_float, _vp],
    "ocpg_gemm_plans": [],
    "ocpg_gemm_tuned": [_vp],
    "ocpg_gemm_tune_rejected": [],
    "ocpg_gemm_set_tuning": [_int],
    "ocpg_gemm_export_picks": [_vp, ctypes.c_longlong],
    "ocpg_gemm_import_picks": [_vp, ctypes.c_longlong],
    "ocpg_window_means3x3_fwd": [_vp, ctypes.c_longlong, _int, _int, _int, _vp, _vp],
    "ocpg_window_means3x3_bwd": [_vp, ctypes.c_longlong, _int, _int, _vp, _vp],
    "ocpg_window_sums3x3_cl_bands": [_int],
    "ocpg_window_sums3x3_cl": [_vp, _int, _int, _int, _int, _int, _vp, _vp],
    "ocpg_window_means3x3_bwd_cl": [_vp, _int, _int, _int, _int, _vp, _vp, _vp],
    "ocpg_mso_conv3x3": [_vp, _int, _int, _vp, _int, _vp, _vp, _int, _vp, _int, _vp, _vp, _int] + [_int] * 6 + [_vp],
    "ocpg_mso_wgrad_rows": [_int] * 4,
    "ocpg_mso_wgrad": [_vp, _int, _int, _vp, _vp, _vp] + [_int] * 8 + [_vp],
    "ocpg_bilinear_nhwc_fwd": [_vp] + [_int] * 6 + [_vp, _vp],
    "ocpg_bilinear_nhwc_bwd": [_vp] + [_int] * 6 + [_vp, _vp],
    "ocpg_small_linear_fwd": [_vp, _int, _vp, _vp, _int, _int, _int, _int, _vp, _vp],
    "ocpg_small_linear_bwd": [_vp, _int, _vp, _int, _vp, _vp, _int, _int, _int, _vp, _vp, _vp, _vp],
    "ocpg_small_linear_f32_fwd": [_vp, _vp, _vp, _int, _int, _int, _vp, _vp],
    "ocpg_small_linear_f32_bwd": [_vp, _vp, _vp, _int, _int, _int, _vp, _vp, _vp, _vp],
    "ocpg_layernorm_blocks": [ctypes.c_longlong],
    "ocpg_groupnorm_cl_work": [ctypes.c_longlong, _int, _int, _int],
    "ocpg_groupnorm_cl_fwd": [_vp, _int, _vp, _vp, ctypes.c_longlong, _int, _int, _int, ctypes.c_float, _vp, _vp, _vp, _vp, _vp],
    "ocpg_groupnorm_cl_bwd": [_vp, _vp, _int, _vp, _vp, _vp, ctypes.c_longlong, _int, _int, _int, _vp, _vp, _vp, _vp],
    "ocpg_groupnorm_cl2cl_fwd": [_vp, _int, _vp, _vp, ctypes.c_longlong, _int, _int, _int, ctypes.c_float, _vp, _vp, _vp, _vp, _vp],
    "ocpg_groupnorm_cl2cl_bwd": [_vp, _vp, _int, _vp, _vp, _vp, ctypes.c_longlong, _int, _int, _int, _vp, _vp, _vp, _vp],
    "ocpg_layernorm_fwd": [_vp, _int, _vp, _vp, ctypes.c_longlong, _int, ctypes.c_float, _vp, _int, _vp, _vp, _vp],
    "ocpg_layernorm_bwd": [_vp, _int, _vp, _int, _vp, _vp, _vp, ctypes.c_longlong, _int, _vp, _int, _vp, _vp, _vp],
    "ocpg_gather_rows_bwd": [_vp, _vp, _vp, _int, _int, _vp, _vp],
    "ocpg_relpos_bias_fwd": [_vp, _vp, _int, ctypes.c_longlong, _int, _vp, _vp, _vp],
    "ocpg_relpos_bias_bwd": [_vp, _vp, _vp, _int, _int, _int, _int, _vp, _vp],
    "ocpg_gather_rows_pad": [_vp, _vp] + [ctypes.c_longlong] * 4 + [_vp, _vp],
    "ocpg_graph_replace_memsets": [_vp, _vp],
    "ocpg_graph_stats": [_vp, _vp],
    "ocpg_graph_memcpy_nodes": [_vp, _vp, _int],
    "ocpg_gemm_bn_act": [_vp] * 6 + [_int, _int] + [ctypes.c_longlong] * 3 + [_vp],
    "ocpg_levelset_fwd_f32": [_vp] * 3 + [_int] * 6 + [_vp] * 4,
    "ocpg_levelset_bwd_f32": [_vp] * 5 + [_int] * 6 + [_vp] * 3,
    "ocpg_proj_fwd_f32": [_vp] * 5 + [_int] * 5 + [_vp] * 5,
    "ocpg_proj_bwd_f32": [_vp] * 9 + [_int] * 5 + [_vp] * 4,
    "ocpg_matcher_cost_f32": [_vp] * 3 + [ctypes.c_longlong] * 4 + [_vp] * 4 + [_int] * 7 + [ctypes.c_float] * 5 + [_vp] * 4,
    "ocpg_dropout_add_ln_fwd": [_vp] * 4 + [ctypes.c_longlong, _int, ctypes.c_float, ctypes.c_float, ctypes.c_ulonglong, ctypes.c_ulonglong, _vp, _int]
                               + [_vp] * 4,
    "ocpg_dropout_add_ln_bwd": [_vp] * 6 + [ctypes.c_longlong, _int, ctypes.c_float, ctypes.c_ulonglong, ctypes.c_ulonglong, _vp, _int] + [_vp] * 4,
    "ocpg_dropout_add_ln_bwd_slots": [ctypes.c_longlong],
    "ocpg_bias_relu_dropout_fwd": [_vp, _vp, ctypes.c_longlong, _int, ctypes.c_float, ctypes.c_ulonglong, ctypes.c_ulonglong, _vp, _int, _vp, _vp],
    "ocpg_bias_relu_dropout_bwd": [_vp, _vp, ctypes.c_longlong, _int, ctypes.c_float, _int, _vp, _vp, _vp],
    "ocpg_bias_relu_dropout_bwd_slots": [ctypes.c_longlong, _int, _int],
    "ocpg_multi_cast": [_vp] * 4 + [_int, ctypes.c_longlong, _int, _int, _vp],
    "ocpg_multi_cast_sum": [_vp] * 6 + [_int, ctypes.c_longlong, _int, _int, _vp],
    "ocpg_colsum_blocks": [ctypes.c_longlong],
    "ocpg_grad_norm_clip": [_vp] * 3 + [_int, ctypes.c_longlong, ctypes.c_float, _vp, _vp, _vp],
    "ocpg_adamw_step": [_vp] * 8 + [_int, ctypes.c_longlong, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_longlong, _vp],
    "ocpg_grad_norm_clip_amp": [_vp] * 3 + [_int, ctypes.c_longlong, ctypes.c_float, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, _vp, _vp],
    "ocpg_adamw_step_amp": [_vp] * 8 + [_int, ctypes.c_longlong, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp],
    "ocpg_lfm_dft_supported": [_int, _int],
    "ocpg_lfm_dft_split": [_int],
    "ocpg_lfm_spectrum_fwd": [_vp, _vp, _vp] + [_int] * 4 + [_vp, _vp, ctypes.c_float, _vp, _vp, _int, _vp],
    "ocpg_lfm_spectrum_inv": [_vp, _int, _vp, _vp, _vp, _vp] + [_int] * 4 + [_vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp],
    "ocpg_colsum_partials": [_vp, ctypes.c_longlong, _int, _int, _vp, _vp],
    "ocpg_det_loss_fwd_f32": [_vp] * 7 + [ctypes.c_float] + [_int] * 5 + [_vp] * 3,
    "ocpg_det_loss_bwd_f32": [_vp] * 8 + [ctypes.c_float] + [_int] * 5 + [_vp] * 3,
    "ocpg_spectral_gate_fwd": [_vp] * 3 + [_int] * 3 + [_vp, _vp],
    "ocpg_spectral_gate_bwd": [_vp] * 4 + [_int] * 3 + [_vp, _vp, _vp],
    "ocpg_spectral_c2p": [_vp] * 3 + [_int] * 3 + [_vp, _int, _vp],
    "ocpg_spectral_p2c": [_vp] * 4 + [_int] * 3 + [_vp, _vp, _int, _vp],
    "ocpg_masked_ce_fwd_f32": [_vp] * 3 + [_int, ctypes.c_longlong, _vp, _vp],
    "ocpg_masked_ce_bwd_f32": [_vp] * 4 + [_int, ctypes.c_longlong, _vp, _vp],
    "ocpg_im2col3x3_nhwc": [_vp] + [_int] * 6 + [_vp, _int, _vp],
    "ocpg_col2im3x3_nhwc": [_vp] + [_int] * 6 + [_vp, _int, _vp],
    "ocpg_attn_smallk_fwd": [_vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, ctypes.c_float] + [_int] * 5
                            + [ctypes.c_float, ctypes.c_ulonglong, ctypes.c_ulonglong, _vp, _vp, ctypes.c_longlong, _vp, _int, _vp],
    "ocpg_attn_smallk_bwd": [_vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, _vp, ctypes.c_longlong, _vp,
                             ctypes.c_float] + [_int] * 5 + [ctypes.c_float, ctypes.c_ulonglong, ctypes.c_ulonglong, _vp, _vp, ctypes.c_longlong,
                                                             _vp, _vp, _int, _vp],
    "ocpg_win_attn_fwd": [_vp, _vp, _vp, ctypes.c_float] + [_int] * 5 + [_vp, _vp, _int, _vp],
    "ocpg_win_attn_bwd": [_vp, _vp, _vp, _vp, ctypes.c_float] + [_int] * 5 + [_vp] * 6 + [_int, _vp],
    "ocpg_win_attn_bwd_mfma": [_vp, _vp, _vp, _vp, ctypes.c_float] + [_int] * 5 + [_vp] * 6 + [_int, _vp],
}


# ---- optional live kernel timing (bench.py): HIP events on the launch stream around every library call ----------
_TIMING = {"on": False, "events": []}
_UNTIMED = ("ocpg_conv3x3_mfma_body_splits", "ocpg_conv3x3_mfma_wgrad_splits", "ocpg_window_sums3x3_cl_bands", "ocpg_lfm_dft_supported", "ocpg_lfm_dft_split", "ocpg_conv3x3_mfma_splits", "ocpg_gemm_set_tuning", "ocpg_gemm_export_picks", "ocpg_gemm_import_picks", "ocpg_colsum_blocks", "ocpg_mso_wgrad_rows", "ocpg_gemm_plans", "ocpg_gemm_tuned", "ocpg_gemm_tune_rejected", "ocpg_bias_relu_dropout_bwd_slots", "ocpg_dropout_add_ln_bwd_slots", "ocpg_groupnorm_cl_work")


def enable_kernel_timing(on=True):
    _TIMING["on"] = on
    _TIMING["events"] = []


def collect_kernel_timing(work=None):
    """-> {symbol: {"ms": total, "n": calls, "work": sum of work(symbol, args) over the calls}}; `work` maps a call's
    arguments (pointers and scalars as passed) to its algorithmic bytes / FLOPs, or None.  (MSDeformAttn is timed by its own
    wrapper, which knows the encoder / decoder shape: ops/functions/ms_deform_attn_func.py.)"""
    torch.cuda.synchronize()
    out = {}
    for name, args, e0, e1 in _TIMING["events"]:
        d = out.setdefault(name, {"ms": 0.0, "n": 0, "work": 0.0, "modelled": 0})
        d["ms"] += e0.elapsed_time(e1)
        d["n"] += 1
        w = work(name, args) if work is not None else None
        if isinstance(w, tuple):            # (FLOPs, low-precision operands): matrix-core symbols
            d["work_lowp"] = d.get("work_lowp", 0.0) + (w[0] if w[1] else 0.0)
            w = w[0]
        if w is not None:
            d["work"] += w
            d["modelled"] += 1
    _TIMING["events"] = []
    _TIMING["on"] = False
    return out


# ---- optional call census (tests: prove WHICH entry points served a module; bench.py: launches per step by symbol) ----
CENSUS = {"on": False, "calls": {}}


def census(on=True):
    """Start (and clear) / stop counting successful (rc == 0) calls per entry point; read CENSUS['calls']."""
    CENSUS["on"] = on
    if on:
        CENSUS["calls"] = {}
    return CENSUS["calls"]


class _Lib:
    """Attribute proxy over the CDLL: every kernel entry point can be bracketed by events when timing is on."""

    def __init__(self, cdll):
        self._cdll = cdll

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)
        if name not in SIGNATURES or name in _UNTIMED:
            setattr(self, name, fn)
            return fn
        timed = not name.startswith("ocpg_msda_")       # MSDeformAttn is timed by its own wrapper (knows enc / dec shape)

        def call(*a):
            if CENSUS["on"]:
                rc = fn(*a)
                if rc == 0:
                    CENSUS["calls"][name] = CENSUS["calls"].get(name, 0) + 1
                return rc
            if not (timed and _TIMING["on"]):
                return fn(*a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            _TIMING["events"].append((name, tuple(getattr(x, "value", x) for x in a), e0, e1))
            return rc
        setattr(self, name, call)
        return call


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m ocpg_amd.csrc.build` "
                "(hipcc --offload-arch=gfx950). The HIP extension is mandatory; there is no CPU/PyTorch fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        L.ocpg_hip_version.restype = ctypes.c_char_p
        L.ocpg_gemm_plans.restype = ctypes.c_longlong
        L.ocpg_gemm_tuned.restype = ctypes.c_longlong
        L.ocpg_gemm_tune_rejected.restype = ctypes.c_longlong
        L.ocpg_gemm_export_picks.restype = ctypes.c_longlong
        L.ocpg_gemm_set_tuning.restype = None
        L.ocpg_bias_relu_dropout_bwd_slots.restype = ctypes.c_longlong
        L.ocpg_dropout_add_ln_bwd_slots.restype = ctypes.c_longlong
        L.ocpg_groupnorm_cl_work.restype = ctypes.c_longlong
        L.ocpg_colsum_blocks.restype = ctypes.c_longlong
        _lib = _Lib(L)
    return _lib


def check(status, what):
    if status != 0:
        if status <= -1000:
            raise RuntimeError(f"{what}: invalid argument #{-status - 1000}")
        raise RuntimeError(f"{what}: HIP error {-status}")


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(name, t):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor: Not implemented on the CPU")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} tensor has to be contiguous")
