"""ctypes binding of libcontextflow_hip.so (C ABI: include/contextflow_hip.h).

This is the only place the product touches native code.  There is NO CPU fallback: if the shared
library is missing, or a tensor is not on a ROCm device, the call raises.  (The CPU oracle lives in
`oracle/` and is test infrastructure only.)
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# CONTEXTFLOW_HIP_LIB: developer override (A/B builds, probe builds of tools/dev); the default is the in-tree library
LIB_PATH = os.environ.get("CONTEXTFLOW_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "libcontextflow_hip.so")
ABI_VERSION = 13

_c_int, _c_i64, _c_f, _c_p = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/contextflow_hip.h declares
SIGNATURES = {
    "cf_abi_version": (_c_int, []),
    "cf_last_error": (ctypes.c_char_p, []),
    "cf_dequant_fwd": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p]),
    "cf_affine": (_c_int, [_c_p, _c_p, _c_i64, _c_f, _c_f, _c_int, _c_p]),
    "cf_logit_fwd": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_p]),
    "cf_sigmoid": (_c_int, [_c_p, _c_p, _c_i64, _c_p]),
    "cf_floor": (_c_int, [_c_p, _c_p, _c_i64, _c_p]),
    "cf_postprocess_inv": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_i64] + [_c_f] * 4 + [_c_p]),
    "cf_preprocess_fwd": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_i64, _c_f, _c_f, _c_f, _c_f, _c_f, _c_p]),
    "cf_preprocess_rng_fwd": (_c_int, [_c_p] * 4 + [ctypes.c_uint64, _c_int, _c_int, _c_int, _c_i64, _c_f, _c_f, _c_f, _c_f, _c_f, _c_int, _c_p]),
    "cf_std_normal_nll": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_i64, _c_p]),
    "cf_squeeze": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_i64, _c_i64, _c_int, _c_p]),
    "cf_conv1x1_fwd": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_i64, _c_i64, _c_p]),
    "cf_slogdet_inverse": (_c_int, [_c_p, _c_int, _c_p, _c_p, _c_p]),
    "cf_slogdet_inverse_batch": (_c_int, [_c_int, _c_p, _c_int, _c_p, _c_p, _c_p]),
    "cf_actnorm_stats_ws_bytes": (_c_i64, [_c_int]),
    "cf_actnorm_stats": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_i64, _c_p]),
    "cf_actnorm_sums": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_i64, _c_p]),
    "cf_actnorm_from_sums": (_c_int, [_c_p, _c_p, ctypes.c_double, _c_p, _c_p, _c_int, _c_p]),
    "cf_actnorm": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "cf_conv2d_reflect": (_c_int, [_c_p, _c_p, _c_p, _c_p] + [_c_int] * 10 + [_c_i64, _c_p]),
    "cf_conv2d_zero": (_c_int, [_c_p, _c_p, _c_p, _c_p] + [_c_int] * 10 + [_c_i64, _c_p]),
    "cf_reflect_pad_adjoint": (_c_int, [_c_p, _c_p] + [_c_int] * 5 + [_c_p]),
    "cf_coupling_apply": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "cf_gmm_prepare": (_c_int, [_c_p] * 6 + [_c_int] * 3 + [_c_p]),
    "cf_gmm_ws_bytes": (_c_i64, [_c_int] * 4),
    "cf_gmm_logprob": (_c_int, [_c_p] * 6 + [_c_int] * 4 + [_c_i64, _c_int, _c_p]),
    "cf_gmm_bwd_sums_supported": (_c_int, [_c_int] * 2),
    "cf_gmm_bwd_sums_ws_bytes": (_c_i64, [_c_int] * 3),
    "cf_gmm_bwd_sums": (_c_int, [_c_p] * 6 + [_c_int] * 3 + [_c_i64, _c_p]),
    "cf_gmm_keyed_ws_bytes": (_c_i64, [_c_int] * 5),
    "cf_gmm_logprob_keyed": (_c_int, [_c_p] * 9 + [_c_int] * 5 + [_c_i64, _c_int, _c_p]),
    "cf_gmm_levels_ws_bytes": (_c_i64, [_c_int, _c_p] + [_c_int] * 3),
    "cf_gmm_logprob_levels": (_c_int, [_c_int] + [_c_p] * 10 + [_c_int] * 3 + [_c_p]),
    "cf_gmm_sample": (_c_int, [_c_p] * 5 + [_c_int, _c_int, _c_p]),
    "cf_flow_step_supported": (_c_int, [_c_int] * 5),
    "cf_flow_step_ws_bytes": (_c_i64, [_c_int] * 3),
    "cf_flow_step_prepare": (_c_int, [_c_p] * 10 + [_c_int] * 3 + [_c_p]),
    "cf_flow_step_prepare_train": (_c_int, [_c_p] * 11 + [_c_int] * 3 + [_c_p]),
    "cf_flow_step_prepare_batch": (_c_int, [_c_int] + [_c_p] * 11 + [_c_int] * 3 + [_c_p]),
    "cf_flow_step_bwd_prepare_batch": (_c_int, [_c_int] + [_c_p] * 6 + [_c_int] * 3 + [_c_p]),
    "cf_flow_step_fwd": (_c_int, [_c_p] * 4 + [_c_int] * 4 + [_c_i64, _c_int, _c_p]),
    "cf_flow_step_inv_ws_bytes": (_c_i64, [_c_int] * 3),
    "cf_flow_step_inv_prepare": (_c_int, [_c_p] * 4 + [_c_int] * 3 + [_c_p]),
    "cf_flow_step_inv": (_c_int, [_c_p] * 4 + [_c_int] * 4 + [_c_i64, _c_int, _c_p]),
    "cf_gmm_bwd_coeffs": (_c_int, [_c_p] * 4 + [_c_int, _c_int, _c_int, _c_p]),
    "cf_gmm_bwd_gx": (_c_int, [_c_p] * 4 + [_c_int, _c_int, _c_i64, _c_p]),
    "cf_gmm_bwd_params": (_c_int, [_c_p] * 8 + [_c_int, _c_int, _c_p]),
    "cf_gmm_bwd_params_w": (_c_int, [_c_p] * 11 + [_c_int] * 3 + [_c_p]),
    "cf_gmm_resp_ws_bytes": (_c_i64, [_c_int] * 4),
    "cf_gmm_resp": (_c_int, [_c_p] * 7 + [_c_int] * 4 + [_c_i64, _c_p]),
    "cf_gmm_quad": (_c_int, [_c_p] * 4 + [_c_int] * 4 + [_c_i64, _c_p]),
    "cf_flow_step_bwd_ws_bytes": (_c_i64, [_c_int] * 3),
    "cf_flow_step_bwd_prepare": (_c_int, [_c_p] * 6 + [_c_int] * 3 + [_c_p]),
    "cf_flow_step_macs": (_c_i64, [_c_int] * 5),
    "cf_flow_step_chain_max_batch": (_c_int, [_c_int] * 3),
    "cf_flow_step_fwd_chain": (_c_int, [_c_p] * 4 + [_c_int] * 5 + [_c_i64, _c_int, _c_p]),
    "cf_step_wgrads_macs": (_c_i64, [_c_int] * 4),
    "cf_flow_step_tape_aux_bytes": (_c_i64, [_c_int] * 4),
    "cf_flow_step_bwd_taped": (_c_int, [_c_p] * 9 + [_c_int] * 5 + [_c_p]),
    "cf_flow_step_fwd_taped": (_c_int, [_c_p] * 8 + [_c_int] * 4 + [_c_i64, _c_int, _c_p]),
    "cf_step_param_grads": (_c_int, [_c_p] * 7 + [_c_int] + [_c_p] * 3 + [_c_int, _c_p]),
    "cf_bf16_split": (_c_int, [_c_int]),
    "cf_adamw_step_batch": (_c_int, [_c_int] + [_c_p] * 6 + [ctypes.c_double] * 5 + [_c_int, _c_p]),
    "cf_step_param_grads_batch": (_c_int, [_c_int] + [_c_p] * 7 + [_c_int] + [_c_p] * 3 + [_c_int, _c_p]),
    "cf_wgrad_ws_bytes": (_c_i64, [_c_int] * 6),
    "cf_wgrad": (_c_int, [_c_p] * 5 + [_c_int] * 6 + [_c_p]),
    "cf_step_wgrads_ws_bytes": (_c_i64, [_c_int] * 4),
    "cf_step_wgrads": (_c_int, [_c_p] * 17 + [_c_int] * 4 + [_c_i64, _c_int, _c_p]),
    "cf_step_wgrads_batch": (_c_int, [_c_int] + [_c_p] * 17 + [_c_int] * 4 + [_c_p, _c_p, _c_p]),
    "cf_linear_wgrad_ws_bytes": (_c_i64, [_c_int] * 3),
    "cf_linear_wgrad": (_c_int, [_c_p] * 5 + [_c_int] * 3 + [_c_p]),
    "cf_linear_wgrad_x2": (_c_int, [_c_p] * 4 + [_c_int] * 3 + [_c_p]),
    "cf_linear": (_c_int, [_c_p] * 5 + [_c_int] * 4 + [_c_p]),
    "cf_linear_group": (_c_int, [_c_int] + [_c_p] * 10 + [_c_int] * 4 + [_c_p]),
    "cf_linear_tn": (_c_int, [_c_p] * 3 + [_c_int] * 3 + [_c_p]),
    "cf_layernorm": (_c_int, [_c_p] * 5 + [_c_int] * 3 + [_c_f, _c_p]),
    "cf_attention": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_f, _c_p]),
    "cf_patchify": (_c_int, [_c_p, _c_p] + [_c_int] * 6 + [_c_i64, _c_int, _c_p]),
    "cf_vit_supported": (_c_int, [_c_int] * 8),
    "cf_vit_ws_bytes": (_c_i64, [_c_int] * 3),
    "cf_vit_flat_params": (_c_i64, [_c_int] * 3),
    "cf_vit_prepare": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_p]),
    "cf_vit_coupling": (_c_int, [_c_p] * 5 + [_c_int] * 8 + [_c_i64, _c_int, _c_p]),
    "cf_vit_step_supported": (_c_int, [_c_int] * 8),
    "cf_vit_step_ws_bytes": (_c_i64, [_c_int] * 2),
    "cf_vit_step_prepare": (_c_int, [_c_p] * 6 + [_c_int] * 2 + [_c_p]),
    "cf_vit_step_fwd": (_c_int, [_c_p] * 5 + [_c_int] * 3 + [_c_i64, _c_p]),
    "cf_vit_step_macs": (_c_i64, [_c_int] * 3),
    "cf_vit_step_rs_prepare_batch": (_c_int, [_c_int] + [_c_p] * 8 + [_c_int] * 2 + [_c_p]),
    "cf_vit_step_bwd_prepare_batch": (_c_int, [_c_int] + [_c_p] * 4 + [_c_int] * 2 + [_c_p]),
    "cf_vit_step_rs_supported": (_c_int, [_c_int] * 8),
    "cf_vit_step_rs_ws_bytes": (_c_i64, [_c_int] * 2),
    "cf_vit_step_rs_prepare": (_c_int, [_c_p] * 6 + [_c_int] * 2 + [_c_p]),
    "cf_vit_step_rs_fwd": (_c_int, [_c_p] * 5 + [_c_int] * 3 + [_c_i64, _c_p]),
    "cf_vit_step_bwd_ws_bytes": (_c_i64, [_c_int] * 2),
    "cf_vit_step_bwd_plane_floats": (_c_i64, [_c_int] * 3),
    "cf_vit_step_bwd_ln_floats": (_c_i64, [_c_int] * 3),
    "cf_vit_step_bwd_prepare": (_c_int, [_c_p] * 4 + [_c_int] * 2 + [_c_p]),
    "cf_vit_step_bwd": (_c_int, [_c_p] * 8 + [_c_int] * 3 + [_c_i64, _c_p]),
    "cf_vit_step_bwd_taped": (_c_int, [_c_p] * 9 + [_c_int] * 3 + [_c_i64, _c_p]),
    "cf_vit_step_tape_tokens": (_c_i64, [_c_int]),
    "cf_vit_step_tape_floats": (_c_i64, [_c_int] * 3),
    "cf_vit_step_fwd_taped": (_c_int, [_c_p] * 5 + [_c_int] * 3 + [_c_i64, _c_p]),
    "cf_vit_step_rs_chain_max_steps": (_c_int, []),
    "cf_vit_step_rs_fwd_chain": (_c_int, [_c_p] * 4 + [_c_int] * 4 + [_c_i64, _c_p]),
    "cf_vit_step_rs_fwd_taped": (_c_int, [_c_p] * 5 + [_c_int] * 3 + [_c_i64, _c_p]),
    "cf_linear_wgrad_group_ws_bytes": (_c_i64, [_c_p] * 3 + [_c_int]),
    "cf_linear_wgrad_group": (_c_int, [_c_p] * 7 + [_c_int, _c_p, _c_p]),
    "cf_spline_table_floats": (_c_i64, [_c_int, _c_int]),
    "cf_spline_prepare": (_c_int, [_c_p] * 4 + [_c_int, _c_int, _c_f, _c_p]),
    "cf_spline": (_c_int, [_c_p] * 4 + [_c_int] * 4 + [_c_f, _c_int, _c_p]),
    "cf_layernorm_bwd_parts": (_c_int, []),
    "cf_layernorm_bwd": (_c_int, [_c_p] * 5 + [_c_int, _c_int, _c_f, _c_p]),
    "cf_attention_bwd": (_c_int, [_c_p] * 3 + [_c_int] * 3 + [_c_f, _c_p]),
    "cf_gelu": (_c_int, [_c_p] * 3 + [_c_i64, _c_int, _c_p]),
    "cf_coupling_apply_bwd": (_c_int, [_c_p] * 6 + [_c_int] * 3 + [_c_i64, _c_i64, _c_p]),
    "cf_channel_sums_ws_bytes": (_c_i64, [_c_int] * 2),
    "cf_channel_sums": (_c_int, [_c_p] * 4 + [_c_int] * 3 + [_c_i64, _c_i64, _c_p]),
    "cf_flow_step_fwd_ctx_taped": (_c_int, [_c_p] * 9 + [_c_int] * 4 + [_c_i64, _c_p]),
    "cf_flow_step_fwd_ctx": (_c_int, [_c_p] * 5 + [_c_int] * 5 + [_c_i64, _c_p]),
    "cf_flow_step_bwd_ctx": (_c_int, [_c_p] * 14 + [_c_int] * 4 + [_c_i64, _c_p]),
    "cf_conv1x1_ctx_bwd": (_c_int, [_c_p] * 7 + [_c_int] * 3 + [_c_i64, _c_i64, _c_p]),
    "cf_actnorm_ctx_bwd": (_c_int, [_c_p] * 8 + [_c_int] * 3 + [_c_i64, _c_i64, _c_p]),
    "cf_sample_channel_sums": (_c_int, [_c_p] * 2 + [_c_int] * 3 + [_c_p]),
    "cf_relu_bwd": (_c_int, [_c_p] * 3 + [_c_i64, _c_p]),
    "cf_gmm_ctx_tables": (_c_int, [_c_p] * 5 + [_c_int] * 4 + [_c_p]),
    "cf_gmm_ctx_logprob_tab": (_c_int, [_c_p] * 10 + [_c_int] * 5 + [_c_i64, _c_int, _c_p]),
    "cf_gmm_ctx_bwd_tab": (_c_int, [_c_p] * 12 + [_c_int] * 5 + [_c_i64, _c_p]),
    "cf_gmm_ctx_pgrad_tab": (_c_int, [_c_p] * 9 + [_c_int] * 5 + [_c_i64, _c_int, _c_p]),
    "cf_gmm_ctx_bwd": (_c_int, [_c_p] * 9 + [_c_int] * 5 + [_c_i64, _c_p]),
    "cf_add_repeat": (_c_int, [_c_p] * 2 + [_c_int] * 4 + [_c_p]),
    "cf_cond_gauss_sample": (_c_int, [_c_p] * 4 + [_c_int, _c_int, _c_p]),
    "cf_activation": (_c_int, [_c_p] * 3 + [_c_i64, _c_int, _c_int, _c_f, _c_f, _c_p, _c_int, _c_p]),
    "cf_sigmoid_ldj": (_c_int, [_c_p] * 3 + [_c_int, _c_int, _c_p]),
    "cf_affine_ctx_blocked_floats": (_c_int, [_c_int] * 3),
    "cf_affine_ctx_fwd": (_c_int, [_c_p] * 7 + [ctypes.c_float] + [_c_p] * 2 + [_c_int] * 4 + [_c_i64, _c_int, _c_int, _c_int, _c_p]),
    "cf_ctx_encode": (_c_int, [_c_p] * 5 + [_c_int] * 4 + [_c_p]),
    "cf_conv1x1_ctx": (_c_int, [_c_p] * 5 + [_c_int] * 3 + [_c_i64, _c_p]),
    "cf_actnorm_ctx": (_c_int, [_c_p] * 6 + [_c_int] * 3 + [_c_i64, _c_p]),
    "cf_add_sample_bias": (_c_int, [_c_p] * 2 + [_c_int] * 4 + [_c_p]),
    "cf_gmm_ctx_logprob": (_c_int, [_c_p] * 7 + [_c_int] * 5 + [_c_i64, _c_int, _c_p]),
    "cf_logdet_combine": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_p]),
    "cf_nll_sum": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_p]),
}

_lib = None


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "contextflow_amd: %s not found. Build it first: "
                "python -c 'import __graft_entry__ as g; g.build()' (needs hipcc, gfx950)." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)       # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        got = handle.cf_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError("contextflow_amd: ABI version %d != expected %d" % (got, ABI_VERSION))
        _lib = handle
    return _lib


class _PtrArray(ctypes.c_void_p):
    """void* to a HOST array of device pointers; keeps the array and its tensors alive (as _Ptr does for one tensor)."""
    _keep = None
    _all = None


def ptr_array(tensors):
    """HOST array of the tensors' device pointers: the `*_batch` entry points take their per-item operands this way (the
    array travels inside the kernel arguments, so it only has to live until the call returns).  `call` launches on the
    device of the first tensor."""
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    out = _PtrArray(ctypes.addressof(arr))
    out._keep = tensors[0]
    out._all = (arr, list(tensors))
    return out


def check(rc, what=""):
    if rc != 0:
        msg = lib().cf_last_error().decode(errors="replace")
        raise RuntimeError("libcontextflow_hip %s failed (code %d): %s" % (what, rc, msg))


class _Stream(ctypes.c_void_p):
    """hipStream_t argument produced by `stream()`: `call` re-derives it from the device of the tensor arguments."""


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)       # (device index) -> hipStream_t as an integer
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream(device=None):
    """torch's current stream on `device` (default: the current device).  Passed to `call`, it is only a placeholder:
    `call` substitutes the current stream of the device the tensor arguments live on.
    (torch.cuda.current_stream() builds a Stream object through three layers of device-index resolution: 8 us per call, 65 calls
    per eager training step; the raw query torch itself uses for its compiled kernels returns the same handle in 0.3 us.)"""
    if _raw_stream is not None and _raw_device is not None:
        if device is None:
            idx = _raw_device()
        elif isinstance(device, int):
            idx = device
        else:
            idx = torch.device(device).index
            if idx is None:
                idx = _raw_device()
        return _Stream(_raw_stream(idx))
    return _Stream(torch.cuda.current_stream(device).cuda_stream)


def device_of(args):
    """The one device every tensor argument (wrapped by `p`) lives on; None for host-only entry points.  Tensors on two
    different devices in one call are a caller bug and raise."""
    dev = None
    for a in args:
        if a.__class__ not in _WRAPPED:              # ints, floats, streams, None: nothing to look at (a failing getattr costs more)
            continue
        t = getattr(a, "_keep", None)
        if t is None:
            continue
        d = t.device
        if dev is None:
            dev = d
        elif d != dev:
            raise RuntimeError("contextflow_amd: tensors of one call live on different devices (%s and %s)" % (dev, d))
    return dev


# indirection for the host-side test of the device selection (tests/test_host.py)
_current_device = torch.cuda.current_device
_device_ctx = torch.cuda.device
_current_stream = torch.cuda.current_stream


def call(name, *args):
    """Enqueue one entry point.  The kernels are launched on the device that OWNS the tensors, on torch's current stream
    of that device - not on whatever device happens to be current (the reference selects `cuda:N` without
    torch.cuda.set_device, model.py:170)."""
    fn = getattr(lib(), name)
    dev = device_of(args)
    if dev is None or dev.index is None or dev.index == _current_device():
        return check(fn(*args), name)
    with _device_ctx(dev):
        st = None
        args = list(args)
        for i, a in enumerate(args):
            if isinstance(a, _Stream):
                if st is None:
                    st = _Stream(_current_stream(dev).cuda_stream)
                args[i] = st
        return check(fn(*args), name)


def require_device(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "contextflow_amd layers run on a ROCm device only (got a %s tensor); there is no CPU path" % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("contextflow_amd: tensors on different devices (%s and %s)" % (dev, t.device))


def f32(t):
    """fp32 + fully contiguous view/copy of a parameter or activation."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def bview(t):
    """(tensor, batch_stride) with everything but the batch dim dense (so channel slices pass un-copied)."""
    if t.dtype != torch.float32:
        t = t.float()
    inner = 1
    ok = True
    for d in range(t.dim() - 1, 0, -1):
        if t.shape[d] != 1 and t.stride(d) != inner:
            ok = False
            break
        inner *= t.shape[d]
    if not ok or (t.shape[0] > 1 and t.stride(0) < inner):
        t = t.contiguous()
    bs = t.stride(0) if t.shape[0] > 1 else inner
    return t, int(bs)


class _Ptr(ctypes.c_void_p):
    """void* that keeps its tensor alive: a temporary built inside an argument list (`p(x.contiguous())`) would
    otherwise be freed - and its block handed to the NEXT temporary of the same argument list - before the launch that
    reads it is even enqueued.  The pointer object lives exactly until the C call has returned."""
    _keep = None


def p(t):
    if t is None:
        return _c_p(0)
    ptr = _Ptr(t.data_ptr())
    ptr._keep = t
    return ptr


_WRAPPED = (_Ptr, _PtrArray)             # the argument wrappers that carry tensors (device_of)
