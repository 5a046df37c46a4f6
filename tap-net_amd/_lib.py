"""ctypes binding of libtapenv.so (C ABI in include/tapenv.h).

There is no CPU fallback: if the shared library has not been built, or no HIP device is visible
when a context is requested, this module raises.
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TAP_LIB_PATH") or os.path.join(_HERE, "libtapenv.so")   # TAP_LIB_PATH: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "tapenv.h")

TAP_OK = 0
TAP_E_INVALID, TAP_E_UNSUPPORTED, TAP_E_HIP, TAP_E_OVERFLOW, TAP_E_NODEVICE, TAP_E_STEPS = -1, -2, -3, -4, -5, -6
TAP_LB_GREEDY, TAP_MACS, TAP_LB = 0, 1, 2
TAP_DT_F32, TAP_DT_I32 = 0, 1
TAP_R_C, TAP_R_CxS, TAP_R_CP, TAP_R_CPxS, TAP_R_CPS, TAP_R_2CPS, TAP_R_CxPxS, TAP_R_CP_HALF = range(8)
TAP_T_FRESH, TAP_T_RATIO = 1, 2
TAP_SB_INITIAL_MASK, TAP_SB_CONTINUE = 1, 2


_vp_t = C.c_void_p


class TapError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("libtapenv: %s (status %d)" % (message, status))
        self.status = status


class TapOverflowError(TapError, IndexError):
    """A placement reached above the container height (the reference raises IndexError)."""


class StepperBuffers(C.Structure):
    """tapenv.h: tap_stepper_buffers"""
    _fields_ = [("bits", _vp_t * 2), ("dyn", _vp_t * 2), ("current", _vp_t * 2), ("mask", _vp_t * 2),
                ("feature", _vp_t), ("decoder_static", _vp_t), ("ratio", _vp_t), ("tour", _vp_t), ("nonbinary", _vp_t),
                ("tour_stride", C.c_int32), ("tour_col0", C.c_int32), ("colsum", _vp_t * 2)]


class RollerBuffers(C.Structure):
    """tapenv.h: tap_roller_buffers"""
    _fields_ = [("static_", _vp_t * 2), ("nodes", _vp_t * 2), ("dynamic", _vp_t), ("bits", _vp_t), ("colsum", _vp_t),
                ("current_mask", _vp_t), ("err", _vp_t), ("feature", _vp_t), ("decoder_static", _vp_t), ("tour", _vp_t),
                ("picked", _vp_t), ("tour_stride", C.c_int32)]


class EnvDesc(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("B", "D", "W", "L", "H", "n_max", "strategy", "flags", "ratio_mode", "feature")]


_lib = None
_lock = threading.RLock()
_ctxs = {}

_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
_PROTOS = {
    "tap_abi_version": (_i, []),
    "tap_status_string": (C.c_char_p, [_i]),
    "tap_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "tap_ctx_destroy": (None, [_vp]),
    "tap_last_error": (C.c_char_p, [_vp]),
    "tap_env_desc_init": (_i, [C.POINTER(EnvDesc), _i, _i, C.POINTER(C.c_int32), _i,
                               C.c_char_p, C.c_char_p, C.c_char_p]),
    "tap_env_state_bytes": (_sz, [C.POINTER(EnvDesc)]),
    "tap_env_feature_len": (_i, [C.POINTER(EnvDesc)]),
    "tap_env_reset": (_i, [_vp, C.POINTER(EnvDesc), _vp, _vp]),
    "tap_env_step": (_i, [_vp, C.POINTER(EnvDesc), _vp, _vp, _i, _vp, _vp, _vp]),
    "tap_env_step_gather": (_i, [_vp, C.POINTER(EnvDesc), _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "tap_env_feature": (_i, [_vp, C.POINTER(EnvDesc), _vp, _vp, _vp]),
    "tap_env_ratio": (_i, [_vp, C.POINTER(EnvDesc), _vp, _vp, _vp, _vp, _vp]),
    "tap_env_export": (_i, [_vp, C.POINTER(EnvDesc), _vp, _vp, _vp, _vp, _vp, _vp]),
    "tap_stable3d_eval": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp]),
    "tap_env_check": (_i, [_vp, C.POINTER(EnvDesc), _vp, C.POINTER(C.c_int32), _vp]),
    "tap_env_errors": (_i, [_vp, C.POINTER(EnvDesc), _vp, _vp, _vp]),
    "tap_bits_words": (_i, [_i, _i]),
    "tap_episode_reward": (_i, [_vp, C.POINTER(EnvDesc), _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "tap_episode_scores": (_i, [_vp, C.POINTER(EnvDesc), _i, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tap_pack_blocks": (_i, [_vp, C.POINTER(EnvDesc), _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tap_precedence": (_i, [_vp, _i, _i, _i, C.POINTER(C.c_int32), _i, _vp, _vp, _vp, _vp, _vp]),
    "tap_ppsg_gt": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, C.c_uint64, _vp, C.c_int64, _i, C.c_int64, _vp, _vp, _vp, _vp]),
    "tap_ppsg_order": (_i, [_vp, _i, _i, _vp, _vp, C.c_uint64, _vp, C.c_int64, _i, _i, _vp, _vp]),
    "tap_ppsg_order2d": (_i, [_vp, _i, _i, _vp, _vp, C.c_uint64, _vp, C.c_int64, _i, _i, _vp, _vp]),
    "tap_ppsg_gt2d": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, C.c_uint64, _vp, C.c_int64, _i, C.c_int64,
                           _vp, _vp, _vp, _vp]),
    "tap_ppsg_check": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "tap_rolling_init": (_i, [_vp, _i, _i, _i, C.POINTER(C.c_int32), _i, _vp, _vp, _vp, _vp, _vp]),
    "tap_rolling_window": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tap_rolling_step": (_i, [_vp, C.POINTER(EnvDesc), _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                              _vp, _vp, _vp, _vp, _vp]),
    "tap_dyn_colsum": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "tap_update_dynamic": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "tap_update_mask": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tap_transition": (_i, [_vp, C.POINTER(EnvDesc), _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp,
                            _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "tap_mask_step": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tap_dyn_bits": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "tap_mask_step_bits": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tap_mask_step_first": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tap_transition_first": (_i, [_vp, C.POINTER(EnvDesc), _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "tap_transition_bits": (_i, [_vp, C.POINTER(EnvDesc), _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp,
                                 _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "tap_transition_launches": (_i, [_vp, C.POINTER(EnvDesc), _i, _i, _i, _i]),
    "tap_bw_probe": (_i, [_vp, _i, _vp, _vp, _sz, _vp]),
    "tap_stepper_create": (_i, [_vp, C.POINTER(EnvDesc), _vp, _i, _i, _i, _i, _i, _i, _vp, C.POINTER(_vp)]),
    "tap_stepper_destroy": (None, [_vp]),
    "tap_stepper_begin": (_i, [_vp, _vp, _vp, _i, _vp]),
    "tap_stepper_begin_shadow": (_i, [_vp, _vp, _vp, _i]),
    "tap_stepper_step": (_i, [_vp, _vp, _vp]),
    "tap_stepper_steps_done": (_i, [_vp]),
    "tap_roller_create": (_i, [_vp, C.POINTER(EnvDesc), _vp, _i, _i, _vp, C.POINTER(_vp)]),
    "tap_roller_destroy": (None, [_vp]),
    "tap_roller_begin": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "tap_roller_step": (_i, [_vp, _vp, _vp]),
    "tap_roller_steps_done": (_i, [_vp]),
}
EXPORTS = tuple(_PROTOS)


def lib():
    """Load libtapenv.so (after torch, so both share torch's HIP runtime)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise ImportError(
                        "%s not found -- build it with `python -c 'import __graft_entry__ as g; "
                        "g.build()'` or `make -C tap-net_amd/csrc`; there is no CPU fallback" % LIB_PATH)
                L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
                for name, (res, args) in _PROTOS.items():
                    f = getattr(L, name)
                    f.restype = res
                    f.argtypes = args
                if L.tap_abi_version() != 1:
                    raise ImportError("libtapenv ABI version mismatch")
                _lib = L
    return _lib


_resolved = {}   # torch.device with an index -> itself, checked once (the seams call this several times per step)


def resolve_device(device):
    """-> torch.device('cuda', index); raises TapError (no CPU fallback) when that is impossible."""
    hit = _resolved.get(device) if isinstance(device, torch.device) else None
    if hit is not None:
        return hit
    dev = torch.device(device)
    if dev.type != "cuda":
        raise TapError(TAP_E_NODEVICE, "tensors must live on a ROCm device, got %s; there is no CPU path" % dev)
    if not torch.cuda.is_available():
        raise TapError(TAP_E_NODEVICE, "no HIP device is visible; there is no CPU path")
    if dev.index is not None:
        _resolved[dev] = dev
        return dev
    return torch.device("cuda", torch.cuda.current_device())


def ctx(device):
    """The per-device context (created on first use)."""
    dev = resolve_device(device)
    idx = dev.index
    c = _ctxs.get(idx)
    if c is None:
        with _lock:
            c = _ctxs.get(idx)
            if c is None:
                h = _vp()
                st = lib().tap_ctx_create(idx, C.byref(h))
                if st != TAP_OK:
                    raise TapError(st, lib().tap_status_string(st).decode())
                c = _ctxs[idx] = h
    return c


def check(status, context):
    if status == TAP_OK:
        return
    msg = lib().tap_last_error(context).decode() or lib().tap_status_string(status).decode()
    if status == TAP_E_OVERFLOW:
        raise TapOverflowError(status, msg)
    raise TapError(status, msg)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_of(device):
    """The HIP stream torch is currently issuing to on ``device`` (read at every call: callers switch streams)."""
    if _raw_stream is not None and isinstance(device, torch.device) and device.index is not None:
        return _vp(_raw_stream(device.index))          # no Stream object, no device look-up: ~0.3 us instead of ~5
    return _vp(torch.cuda.current_stream(device).cuda_stream)


def raw_stream(index):
    """The current HIP stream of device ``index`` as an integer handle (what a c_void_p argument accepts)."""
    if _raw_stream is not None:
        return _raw_stream(index)
    return torch.cuda.current_stream(index).cuda_stream


def ptr(t):
    """Device pointer of a tensor for the C ABI.  A host tensor here would hand the kernels a host
    address (a GPU memory fault, not an exception), so it is refused loudly instead."""
    if t is None:
        return None
    if not t.is_cuda:
        raise TapError(TAP_E_INVALID, "tensor on %s passed to a device entry point (move it to the GPU first)" % (t.device,))
    return _vp(t.data_ptr())


def make_desc(batch_size, container_size, blocks_num, reward_type, heightmap_type, packing_strategy):
    d = EnvDesc()
    cs = (C.c_int32 * len(container_size))(*[int(v) for v in container_size])
    st = lib().tap_env_desc_init(C.byref(d), int(batch_size), len(container_size), cs, int(blocks_num),
                                 reward_type.encode(), heightmap_type.encode(), packing_strategy.encode())
    if st != TAP_OK:
        raise TapError(st, "cannot describe container %s / %s / %s / %s: %s" % (
            list(container_size), reward_type, heightmap_type, packing_strategy,
            lib().tap_status_string(st).decode()))
    return d
