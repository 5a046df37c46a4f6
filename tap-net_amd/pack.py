"""Device counterparts of the reference's ``pack.py`` seams, same names and signatures:

* ``update_dynamic`` (pack.py:333-376) and ``update_mask`` (pack.py:276-331) -- the callables
  trainer.py:475-476 injects into ``DRL`` as ``update_fn`` / ``mask_fn``;
* ``reward`` (pack.py:378-473) -- ``kwargs['reward_fn']``;
* ``PACKDataset`` (pack.py:25-273) -- the ``Dataset`` the trainer wraps in a ``DataLoader``.

All tensors stay on the ROCm device; results are fresh tensors (the input ``dynamic`` is never
mutated: the trainer re-uses it for the critic, trainer.py:214).
"""
import itertools
import math
import os
import weakref

import numpy as np
import torch
from torch.utils.data import Dataset

from . import _lib

_UPDATE_ROWS = {  # pack.py:338-357
    'simple': 1, 'rot': 1, 'rot-old': 1, 'bot': 3, 'bot-rot': 3, 'use-static': 3, 'use-pnet': 3,
    'mul': 3, 'mul-with': 3,
}


def _block_dim(static, input_type):
    if input_type not in _UPDATE_ROWS:
        raise ValueError("unknown input_type %r" % (input_type,))
    return int(static.shape[1]) - (2 if input_type in ('mul', 'mul-with') else 1)   # pack.py:288-302


def _rotate_types(block_dim, allow_rot):
    return math.factorial(block_dim) if allow_rot else 1                           # pack.py:306-309


def _f32c(t):
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.to(torch.float32).contiguous()
    return t


# ---- (B, 3, nR) column-sum shadow of a dynamic tensor -------------------------------------------
# update_mask only needs the per-section column sums of `dynamic` (pack.py:324-326).  update_dynamic
# produces them for its output as a by-product (old sums minus the zeroed rows), so the mask_fn
# that follows it in model.py:376-384 does not have to read the 3n x nR slab again.
_shadow = {}


def _shadow_put(dynamic, colsum):
    key = id(dynamic)
    _shadow[key] = (weakref.ref(dynamic, lambda _r, k=key: _shadow.pop(k, None)), dynamic._version, colsum)


def _shadow_get(dynamic):
    hit = _shadow.get(id(dynamic))
    if hit is not None and hit[0]() is dynamic and hit[1] == dynamic._version:
        return hit[2]
    return None


def dynamic_colsum(dynamic, blocks_num):
    """(B, 3, nR) float32 column sums of the move / small / large sections of ``dynamic``."""
    cs = _shadow_get(dynamic)
    if cs is not None:
        return cs
    dyn = _f32c(dynamic)
    B, rows, nR = dyn.shape
    cs = torch.empty(B, 3, nR, dtype=torch.float32, device=dyn.device)
    c = _lib.ctx(dyn.device)
    with torch.cuda.device(dyn.device):
        _lib.check(_lib.lib().tap_dyn_colsum(c, B, blocks_num, nR, rows, _lib.ptr(dyn), _lib.ptr(cs),
                                             _lib.stream_of(dyn.device)), c)
    _shadow_put(dynamic, cs)
    return cs


# ---- (B, nR) bit shadow of a 0/1-valued dynamic tensor (tapenv.h: tap_dyn_bits) -------------------
# Preferred over the column sums when the tensor allows it: the step then writes the new fp32 tensor
# from the bits and never reads the old one.  The cache is keyed by tensor identity and version: tensors
# mutated out of band (.data writes, foreign kernels) must not be passed again.
#   entry = shadow tensor | 'binary' (0/1 but the shape has no shadow) | 'nonbinary'
_bshadow = {}
_binary_mode = 'check'
_deferred = {}          # device index -> int32 counter of non-0/1 elements seen in 'trust' mode
_steppers = weakref.WeakSet()   # live EpisodeSteppers: each counts the non-0/1 elements of the tensors it was given


def set_binary_check(mode):
    """How the seams learn that a fresh ``dynamic`` tensor holds only 0 and 1 (what PACKDataset builds and
    update_dynamic preserves):

    'check' (default)  count the other values on the device and READ the count: one device->host sync per
                       new tensor (once per episode), exact fallback for tensors with other values;
    'trust'            no host read -- the seams stay asynchronous and capturable in a HIP graph; the count is
                       accumulated on the device and ``check_binary()`` reports it when the caller asks."""
    global _binary_mode
    if mode not in ('check', 'trust'):
        raise ValueError("mode must be 'check' or 'trust'")
    _binary_mode = mode


def check_binary(device=None):
    """Deferred form of the 0/1 test in 'trust' mode (synchronises): raises ValueError if any tensor handed to
    the seams since the last call held a value other than 0 or 1 -- results computed from it are invalid."""
    bad = 0
    want = None if device is None else _lib.resolve_device(device).index
    for idx, cnt in list(_deferred.items()) + [(sp._idx, sp._nonbinary) for sp in list(_steppers)]:
        if want is None or want == idx:
            bad += int(cnt.item())
            cnt.zero_()
    if bad:
        raise ValueError("%d element(s) of a `dynamic` tensor were neither 0 nor 1 while pack.set_binary_check('trust') "
                         "was in force" % bad)


def _bits_put(dynamic, bits):
    key = id(dynamic)
    _bshadow[key] = (weakref.ref(dynamic, lambda _r, k=key: _bshadow.pop(k, None)), dynamic._version, bits)


def _bits_state(dynamic, build=False):
    """-> shadow tensor | 'binary' | 'nonbinary' | None (unknown and not asked to find out)."""
    hit = _bshadow.get(id(dynamic))
    if hit is not None and hit[0]() is dynamic and hit[1] == dynamic._version:
        return hit[2]
    if not build or dynamic.dim() != 3:
        return None
    ok_shape = bits_supported(int(dynamic.shape[1]), int(dynamic.shape[2]))
    if _binary_mode == 'trust':
        dev = _lib.resolve_device(dynamic.device)
        cnt = _deferred.get(dev.index)
        if cnt is None:
            cnt = _deferred[dev.index] = torch.zeros(1, dtype=torch.int32, device=dev)
        shadow, _ = dynamic_bits(dynamic, counter=cnt, want_bits=ok_shape)
        state = shadow if ok_shape else 'binary'
    else:
        shadow, bad = dynamic_bits(dynamic, want_bits=ok_shape)
        if int(bad.item()) != 0:                     # one device->host read per episode
            state = 'nonbinary'
        else:
            state = shadow if ok_shape else 'binary'
    _bits_put(dynamic, state)
    return state


def _bits_get(dynamic, build=False):
    st = _bits_state(dynamic, build)
    return st if isinstance(st, torch.Tensor) else None


def _mask_step_bits(bits_in, st, ptr, n, R, rows, update_rows, mask_in, bits_out, dyn_out, cur, new):
    c = _lib.ctx(st.device)
    with torch.cuda.device(st.device):
        _lib.check(_lib.lib().tap_mask_step_bits(
            c, st.shape[0], n, R, rows, update_rows, _lib.ptr(bits_in), _lib.ptr(st), st.shape[1], _lib.ptr(ptr),
            _lib.ptr(mask_in), _lib.ptr(bits_out), _lib.ptr(dyn_out), _lib.ptr(cur), _lib.ptr(new),
            _lib.stream_of(st.device)), c)


def update_dynamic(dynamic, static, chosen_idx, input_type, allow_rot):
    """pack.update_dynamic (pack.py:333-376): zero the chosen block's rows, out of place."""
    block_dim = _block_dim(static, input_type)
    R = _rotate_types(block_dim, allow_rot)
    dyn, st = _f32c(dynamic), _f32c(static)
    B, rows, nR = dyn.shape
    n = nR // R                                                                    # pack.py:367
    ptr = chosen_idx.to(torch.int64).contiguous()
    out = torch.empty_like(dyn)
    bits_in = _bits_get(dynamic, build=True)
    if bits_in is not None:                          # 0/1 tensor: expand the new one from its bit shadow
        bits_out = torch.empty_like(bits_in)
        _mask_step_bits(bits_in, st, ptr, n, R, rows, _UPDATE_ROWS[input_type], None, bits_out, out, None, None)
        _bits_put(out, bits_out)
        return out
    c = _lib.ctx(dyn.device)
    if _bits_state(dynamic) == 'nonbinary':
        # values other than 0/1: "old column sum - cleared row" in fp32 is not the re-summed value the
        # reference tests (pack.py:323-329), so no incremental shadow -- update_mask re-reduces the tensor
        with torch.cuda.device(dyn.device):
            _lib.check(_lib.lib().tap_update_dynamic(
                c, B, n, nR, rows, _UPDATE_ROWS[input_type], _lib.ptr(dyn), _lib.ptr(st), st.shape[1],
                _lib.ptr(ptr), _lib.ptr(out), None, None, _lib.stream_of(dyn.device)), c)
        _bits_put(out, 'nonbinary')
        return out
    # first call of an episode: build the column-sum shadow once (one extra read of the slab) so this
    # and every later step run the single-round-trip streaming kernel and update_mask never re-reads
    cs_in = dynamic_colsum(dynamic, n)
    cs_out = torch.empty_like(cs_in)
    with torch.cuda.device(dyn.device):
        _lib.check(_lib.lib().tap_update_dynamic(
            c, B, n, nR, rows, _UPDATE_ROWS[input_type], _lib.ptr(dyn), _lib.ptr(st), st.shape[1],
            _lib.ptr(ptr), _lib.ptr(out), _lib.ptr(cs_in), _lib.ptr(cs_out),
            _lib.stream_of(dyn.device)), c)
    _shadow_put(out, cs_out)
    _bits_put(out, 'binary')
    return out


def update_mask(mask, dynamic, static, chosen_idx, input_type, allow_rot):
    """pack.update_mask (pack.py:276-331) -> (new_mask.float(), chosen_mask)."""
    block_dim = _block_dim(static, input_type)
    R = _rotate_types(block_dim, allow_rot)
    nR = int(dynamic.shape[-1])
    n = nR // R                                                                    # pack.py:311
    m = _f32c(mask)
    ptr = chosen_idx.to(torch.int64).contiguous()
    B = m.shape[0]
    cur = torch.empty_like(m)
    new = torch.empty_like(m)
    bits = _bits_get(dynamic)
    if bits is not None:                             # column sums = popcounts of the shadow
        _mask_step_bits(bits, _f32c(static), ptr, n, R, int(dynamic.shape[1]), 0, m, None, None, cur, new)
        return cur, new
    cs = dynamic_colsum(dynamic, n)
    c = _lib.ctx(m.device)
    with torch.cuda.device(m.device):
        _lib.check(_lib.lib().tap_update_mask(c, B, n, R, _lib.ptr(m), _lib.ptr(cs), _lib.ptr(ptr),
                                              _lib.ptr(cur), _lib.ptr(new), _lib.stream_of(m.device)), c)
    return cur, new


def initial_mask(dynamic, blocks_num, bits=None):
    """The mask DRL.forward builds before its loop (model.py:297-307) -> (current_mask, mask).
    ``bits``: the tensor's bit shadow when the caller holds it (column sums = popcounts, the fp32 tensor is
    not read)."""
    dyn = _f32c(dynamic)
    B, rows, nR = dyn.shape
    cur = torch.empty(B, nR, dtype=torch.float32, device=dyn.device)
    mask = torch.empty(B, nR, dtype=torch.float32, device=dyn.device)
    if bits is None:
        bits = _bits_get(dynamic)
    if bits is not None:
        _mask_step_bits_raw(dyn.device, B, blocks_num, nR // blocks_num, rows, 0, bits, None, 0, None, None, None, None,
                            cur, mask)
        return cur, mask
    cs = dynamic_colsum(dynamic, blocks_num)
    c = _lib.ctx(dyn.device)
    with torch.cuda.device(dyn.device):
        _lib.check(_lib.lib().tap_update_mask(c, B, blocks_num, nR // blocks_num, None, _lib.ptr(cs), None,
                                              _lib.ptr(cur), _lib.ptr(mask), _lib.stream_of(dyn.device)), c)
    return cur, mask


def _mask_step_bits_raw(device, B, n, R, rows, update_rows, bits_in, st, st_rows, ptr, mask_in, bits_out, dyn_out, cur, new):
    c = _lib.ctx(device)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().tap_mask_step_bits(
            c, B, n, R, rows, update_rows, _lib.ptr(bits_in), _lib.ptr(st), st_rows, _lib.ptr(ptr),
            _lib.ptr(mask_in), _lib.ptr(bits_out), _lib.ptr(dyn_out), _lib.ptr(cur), _lib.ptr(new),
            _lib.stream_of(device)), c)


def first_shadow_and_mask(dynamic, blocks_num, counter=None):
    """Bit shadow AND initial mask (model.py:297-307) of a fresh 0/1 ``dynamic`` in one launch that reads the
    tensor once (tap_mask_step_first, ptr = NULL) -> (bits, current_mask, mask).  The count of elements that
    are neither 0 nor 1 goes to ``counter`` (default: the deferred per-device counter check_binary() reads)."""
    dyn = _f32c(dynamic)
    B, rows, nR = dyn.shape
    dev = _lib.resolve_device(dyn.device)
    if counter is None:
        counter = _deferred.get(dev.index)
        if counter is None:
            counter = _deferred[dev.index] = torch.zeros(1, dtype=torch.int32, device=dev)
    bits = torch.empty(B, _bit_planes(rows) * nR, dtype=torch.int64, device=dev)
    cur = torch.empty(B, nR, dtype=torch.float32, device=dev)
    mask = torch.empty(B, nR, dtype=torch.float32, device=dev)
    c = _lib.ctx(dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().tap_mask_step_first(
            c, B, blocks_num, nR // blocks_num, rows, 0, _lib.ptr(dyn), None, 0, None, None, _lib.ptr(bits), None,
            _lib.ptr(cur), _lib.ptr(mask), _lib.ptr(counter), _lib.stream_of(dev)), c)
    return bits, cur, mask


def bits_supported(rows, nR):
    """Shapes the bit shadow of `dynamic` covers (tapenv.h: tap_bits_words): one word per column up to 64
    rows (windows of at most 21 nodes), two up to 128 rows (42 nodes)."""
    return _lib.lib().tap_bits_words(int(rows), int(nR)) > 0


def _bit_planes(rows):
    return 2 if rows > 64 else 1


def dynamic_bits(dynamic, counter=None, want_bits=True):
    """Bit shadow of a 0/1-valued ``dynamic`` (B, rows <= 128, nR): -> (bits (B, nR) int64 with bit r of
    word j = dynamic[b, r, j] != 0 -- above 64 rows (B, 2*nR): plane 0 = rows 0..63, plane 1 = rows 64.. --,
    nonbinary (1,) int32 = number of elements that are neither 0 nor 1;
    the shadow stands for the tensor only when that count is 0).  ``counter``: accumulate the count into
    this device int32 instead of a fresh one; ``want_bits`` False: count only (any number of rows)."""
    dyn = _f32c(dynamic)
    B, rows, nR = dyn.shape
    bits = torch.empty(B, _bit_planes(rows) * nR, dtype=torch.int64, device=dyn.device) if want_bits else None
    bad = counter if counter is not None else torch.zeros(1, dtype=torch.int32, device=dyn.device)
    c = _lib.ctx(dyn.device)
    with torch.cuda.device(dyn.device):
        _lib.check(_lib.lib().tap_dyn_bits(c, B, nR, rows, _lib.ptr(dyn), _lib.ptr(bits), _lib.ptr(bad),
                                           _lib.stream_of(dyn.device)), c)
    return bits, bad


class MaskStepper(object):
    """update_dynamic + update_mask fused into one launch per step (model.py:376-386), for callers
    that drive the episode loop themselves (tap-net_amd.rollout, bench.py).

    ``bits``: carry ``dynamic`` as its bit shadow between steps (the fp32 tensor is then written,
    never re-read).  None = whenever the shape allows and the tensor is 0/1-valued (one device->host
    read of a counter at construction), True = required (ValueError otherwise), False = never, a
    tensor = the shadow itself (dynamic_bits layout) when the caller already has it."""

    def __init__(self, static, dynamic, input_type='bot', allow_rot=True, bits=None):
        self.static = _f32c(static)
        self.dynamic = _f32c(dynamic)
        self.block_dim = _block_dim(static, input_type)
        self.R = _rotate_types(self.block_dim, allow_rot)
        self.B, self.rows, self.nR = self.dynamic.shape
        self.n = self.nR // self.R
        self.update_rows = _UPDATE_ROWS[input_type]
        self.bits, self.colsum, self.nonbinary = None, None, False
        first = None
        if isinstance(bits, torch.Tensor):               # a shadow the caller already holds (rolling windows)
            self.bits = bits
        elif (bits is not False and _binary_mode == 'trust' and _bits_state(dynamic) is None and
              bits_supported(self.rows, self.nR)):
            first = first_shadow_and_mask(self.dynamic, self.n)   # shadow + initial mask: one read, no host sync
            self.bits = first[0]
            _bits_put(dynamic, self.bits)
        else:
            st = _bits_state(dynamic, build=True)
            if isinstance(st, torch.Tensor) and bits is not False:
                self.bits = st
            self.nonbinary = isinstance(st, str) and st == 'nonbinary'
        if bits is True and self.bits is None:
            raise ValueError("dynamic cannot be carried as a bit shadow (needs rows <= 128, nR % 4 == 0, "
                             "nR <= 256 and only 0/1 values)")
        if self.bits is None:                            # the column sums are only needed without the shadow
            self.colsum = dynamic_colsum(self.dynamic, self.n)
        if first is not None:
            self.current_mask, self.mask = first[1], first[2]
        else:
            self.current_mask, self.mask = initial_mask(self.dynamic, self.n, bits=self.bits)
        # per-step plumbing resolved once (an eager decoding loop is host-bound: every look-up here is paid per step)
        self._dev = _lib.resolve_device(self.dynamic.device)
        self._ctx = _lib.ctx(self._dev)
        self._static_ptr, self._static_rows = _lib.ptr(self.static), int(self.static.shape[1])

    def _as_ptr(self, ptr):
        if ptr.dtype is torch.int64 and ptr.device == self._dev and ptr.is_contiguous():
            return ptr
        return ptr.to(device=self._dev, dtype=torch.int64).contiguous()

    def _check_step_args(self, ptr, dyn_out):
        if tuple(ptr.shape) != (self.B,):
            raise ValueError("ptr must be (%d,), got %s" % (self.B, tuple(ptr.shape)))
        if dyn_out is not None and (dyn_out.shape != self.dynamic.shape or dyn_out.dtype != torch.float32 or
                                    not dyn_out.is_contiguous() or dyn_out.data_ptr() == self.dynamic.data_ptr()):
            raise ValueError("dyn_out must be a distinct contiguous float32 tensor of shape %s" % (tuple(self.dynamic.shape),))

    def step(self, ptr, dyn_out=None):
        """-> (new_dynamic, current_mask, mask).  ``dyn_out`` lets a caller recycle buffers."""
        ptr = self._as_ptr(ptr)
        self._check_step_args(ptr, dyn_out)
        out = dyn_out if dyn_out is not None else torch.empty_like(self.dynamic)
        cur = torch.empty_like(self.mask)
        new = torch.empty_like(self.mask)
        c = self._ctx
        if self.bits is not None:
            nb = torch.empty_like(self.bits)
            with torch.cuda.device(self._dev):
                _lib.check(_lib.lib().tap_mask_step_bits(
                    c, self.B, self.n, self.R, self.rows, self.update_rows, _lib.ptr(self.bits),
                    self._static_ptr, self._static_rows, _lib.ptr(ptr), _lib.ptr(self.mask),
                    _lib.ptr(nb), _lib.ptr(out), _lib.ptr(cur), _lib.ptr(new), _lib.stream_of(self._dev)), c)
            self.dynamic, self.bits, self.current_mask, self.mask = out, nb, cur, new
            return out, cur, new
        cs = torch.empty_like(self.colsum)
        if self.nonbinary:
            self._step_resum(ptr, out, cs, cur, new)
            return out, cur, new
        with torch.cuda.device(out.device):
            _lib.check(_lib.lib().tap_mask_step(
                c, self.B, self.n, self.R, self.rows, self.update_rows, _lib.ptr(self.dynamic),
                _lib.ptr(self.static), self.static.shape[1], _lib.ptr(ptr), _lib.ptr(self.mask),
                _lib.ptr(self.colsum), _lib.ptr(out), _lib.ptr(cs), _lib.ptr(cur), _lib.ptr(new),
                _lib.stream_of(out.device)), c)
        self.dynamic, self.colsum, self.current_mask, self.mask = out, cs, cur, new
        return out, cur, new

    def _step_resum(self, ptr, out, cs, cur, new):
        """Values other than 0/1: re-sum the new tensor like the reference (pack.py:323-326) instead of
        "old sum - cleared row", which is only exact for 0/1 data -- three launches, exact."""
        c = _lib.ctx(out.device)
        L, S = _lib.lib(), _lib.stream_of(out.device)
        with torch.cuda.device(out.device):
            _lib.check(L.tap_update_dynamic(c, self.B, self.n, self.nR, self.rows, self.update_rows,
                                            _lib.ptr(self.dynamic), _lib.ptr(self.static), self.static.shape[1],
                                            _lib.ptr(ptr), _lib.ptr(out), None, None, S), c)
            _lib.check(L.tap_dyn_colsum(c, self.B, self.n, self.nR, self.rows, _lib.ptr(out), _lib.ptr(cs), S), c)
            _lib.check(L.tap_update_mask(c, self.B, self.n, self.R, _lib.ptr(self.mask), _lib.ptr(cs), _lib.ptr(ptr),
                                         _lib.ptr(cur), _lib.ptr(new), S), c)
        self.dynamic, self.colsum, self.current_mask, self.mask = out, cs, cur, new


class EnvTransition(MaskStepper):
    """One call per decoding step: update_dynamic + update_mask + gather + add_new_block
    (tap_transition*), optionally starting from a fresh container and optionally emitting
    calc_ratio -- ONE kernel for the lane-per-cell shapes, the same step as its two launches behind the
    same entry point for the others (``env.fused_ok`` tells which)."""

    def __init__(self, static, dynamic, env, input_type='bot', allow_rot=True, bits=None):
        super(EnvTransition, self).__init__(static, dynamic, input_type, allow_rot, bits)
        self.env = env
        if env.batch_size != self.B or env.block_dim != self.block_dim:
            raise ValueError("container batch / dimension does not match the instance tensors")
        import ctypes as C
        self._desc_ref, self._state_ptr = C.byref(env.desc), _lib.ptr(env._state)

    def step(self, ptr, fresh=False, want_ratio=False, want_feature=True, dyn_out=None):
        """-> (new_dynamic, current_mask, mask, feature, ratio)."""
        import ctypes as C
        ptr = self._as_ptr(ptr)
        self._check_step_args(ptr, dyn_out)
        out = dyn_out if dyn_out is not None else torch.empty_like(self.dynamic)
        cur = torch.empty_like(self.mask)
        new = torch.empty_like(self.mask)
        feat = self.env._new_feature() if want_feature else None
        ratio = torch.empty(self.B, dtype=torch.float32, device=self._dev) if want_ratio else None
        flags = (_lib.TAP_T_FRESH if fresh else 0) | (_lib.TAP_T_RATIO if want_ratio else 0)
        c = self._ctx
        if self.bits is not None:
            nb = torch.empty_like(self.bits)
            with torch.cuda.device(self._dev):
                _lib.check(_lib.lib().tap_transition_bits(
                    c, self._desc_ref, self._state_ptr, self.n, self.R, self.rows,
                    self.update_rows, _lib.ptr(self.bits), self._static_ptr, self._static_rows,
                    _lib.ptr(ptr), _lib.ptr(self.mask), _lib.ptr(nb), _lib.ptr(out), _lib.ptr(cur),
                    _lib.ptr(new), _lib.ptr(feat), _lib.ptr(ratio), flags, _lib.stream_of(self._dev)), c)
            self.dynamic, self.bits, self.current_mask, self.mask = out, nb, cur, new
            return out, cur, new, feat, ratio
        cs = torch.empty_like(self.colsum)
        if self.nonbinary:                               # exact, unfused (see MaskStepper._step_resum)
            if fresh:
                self.env.reset()
            self._step_resum(ptr, out, cs, cur, new)
            feat = self.env.add_new_blocks_gather(self.static, ptr, want_feature=want_feature)
            ratio = self.env.calc_ratios() if want_ratio else None
            return out, cur, new, feat, ratio
        with torch.cuda.device(out.device):
            _lib.check(_lib.lib().tap_transition(
                c, C.byref(self.env.desc), _lib.ptr(self.env._state), self.n, self.R, self.rows,
                self.update_rows, _lib.ptr(self.dynamic), _lib.ptr(self.static), self.static.shape[1],
                _lib.ptr(ptr), _lib.ptr(self.mask), _lib.ptr(self.colsum), _lib.ptr(out), _lib.ptr(cs),
                _lib.ptr(cur), _lib.ptr(new), _lib.ptr(feat), _lib.ptr(ratio), flags,
                _lib.stream_of(out.device)), c)
        self.dynamic, self.colsum, self.current_mask, self.mask = out, cs, cur, new
        return out, cur, new, feat, ratio


class NonBinaryDynamic(ValueError):
    """A ``dynamic`` tensor handed to an EpisodeStepper held values other than 0 and 1 (it carries the tensor as
    its bit shadow); rollout.run_episode falls back to the general EnvTransition path."""


class EpisodeStepper(object):
    """The step object of a decoding loop (tapenv.h: tap_stepper) -- EnvTransition for callers that pay per step
    on the host: it OWNS two phases of every per-step output (new ``dynamic``, its bit shadow, both masks), the
    feature, ``decoder_static`` (the gather of model.py:404-406, written by the step's own launch), the tour and the
    ratio; an episode is ``begin(static, dynamic)`` + ``steps`` calls of ``step(ptr)``, each ONE C call with two
    arguments -- no tensor is allocated and no torch op is issued between the policy's calls.

    After ``begin`` / ``step`` the attributes ``dynamic``, ``current_mask``, ``mask``, ``decoder_dynamic``,
    ``decoder_static`` hold the values model.py's loop variables of the same names would; they are views of the
    stepper's buffers: a step overwrites the phase written two steps ago (``inplace_dynamic=True``: ``dynamic`` is ONE
    tensor that every step updates in place), the next ``begin`` everything -- clone what must outlive that.  ``static`` / ``dynamic`` given to ``begin`` are only read (the trainer re-uses them,
    trainer.py:214) and are kept referenced for the episode.

    Shapes are fixed at construction from the example tensors; needs a window with a bit shadow
    (``bits_supported``) and 0/1-valued ``dynamic`` tensors: under ``set_binary_check('check')`` ``begin`` reads
    the launch's counter of other values (one host sync per episode) and raises NonBinaryDynamic, under 'trust'
    the count stays on the device until ``check_binary()`` (or ``check()``) asks for it."""

    def __init__(self, static, dynamic, env, input_type='bot', allow_rot=True, steps=None, want_tour=True,
                 tour=None, tour_col0=0, expand_dynamic=True, inplace_dynamic=False):
        import ctypes as C
        self.block_dim = _block_dim(static, input_type)
        self.R = _rotate_types(self.block_dim, allow_rot)
        # (``dynamic`` may be the tensor's SHAPE (B, rows, nR) when none exists: windows carried as their bit shadow only)
        self.B, self.rows, self.nR = (int(v) for v in (dynamic if isinstance(dynamic, (tuple, list)) else dynamic.shape))
        self.n = self.nR // self.R
        self.static_rows = int(static.shape[1])
        self.update_rows = _UPDATE_ROWS[input_type]
        self.steps = self.n if steps is None else int(steps)
        # windows without a bit shadow (rows > 128, nR % 4 != 0, nR > 256) run tap_transition's fp32-copy form on the
        # column-sum shadow behind the same three calls
        self._copy = not bits_supported(self.rows, self.nR)
        if self._copy and not expand_dynamic:
            raise ValueError("expand_dynamic=False needs a window with a bit shadow (rows <= 128, nR % 4 == 0, nR <= 256)")
        if env.batch_size != self.B or env.block_dim != self.block_dim:
            raise ValueError("container batch / dimension does not match the instance tensors")
        self.env = env
        dev = self._dev = env.device
        self._idx = dev.index
        f32 = dict(dtype=torch.float32, device=dev)
        D = self.block_dim
        words = _bit_planes(self.rows) * self.nR
        self._bits = [torch.empty(self.B, words, dtype=torch.int64, device=dev) if not self._copy else None for _ in range(2)]
        self._colsum = [torch.empty(self.B, 3, self.nR, **f32) if self._copy else None for _ in range(2)]
        # expand_dynamic=False: update_dynamic's result stays in its bit shadow (``dynamic_bits``) and the fp32 tensor
        # of model.py:378 is not written -- 78 % of a c2 step's bytes; ``dynamic`` is then None after a step
        self.expand_dynamic = bool(expand_dynamic)
        # inplace_dynamic=True (windows with a bit shadow; loops under no_grad -- validation, serving): ONE fp32 tensor for
        # the episode instead of two alternating ones.  Step 0 writes all of it, every later step the rows it clears
        # (update_dynamic's result differs from its input in rows real + n*i only, pack.py:370-374; the reference clones,
        # pack.py:368, because autograd keeps each step's tensor): ``dynamic`` after a step is that one tensor, the
        # previous steps' values are gone.  tapenv.h: tap_stepper_buffers.dyn
        self.inplace_dynamic = bool(inplace_dynamic)
        if self.inplace_dynamic and (self._copy or not expand_dynamic):
            raise ValueError("inplace_dynamic=True needs the fp32 tensor (expand_dynamic=True) of a window with a bit shadow")
        if self.inplace_dynamic:
            one = torch.empty(self.B, self.rows, self.nR, **f32)
            self._dyn = [one, one]
        else:
            self._dyn = [torch.empty(self.B, self.rows, self.nR, **f32) if expand_dynamic else None for _ in range(2)]
        self._cur = [torch.empty(self.B, self.nR, **f32) for _ in range(2)]
        self._mask = [torch.empty(self.B, self.nR, **f32) for _ in range(2)]
        # the decoder inputs are zeros before step 0 (pack.py:258-264); one flat buffer so that begin() clears both
        # with one launch
        fshape = env._feature_shape()
        flen = int(np.prod(fshape[1:]))
        self._dec = torch.zeros(self.B * (flen + D), **f32)
        self.decoder_dynamic = self._dec[:self.B * flen].view(fshape)
        self.decoder_static = self._dec[self.B * flen:].view(self.B, D, 1)
        self.ratio = torch.empty(self.B, **f32)
        # ``tour``: write the picks into columns tour_col0 .. tour_col0 + steps - 1 of the caller's (B, stride) int64
        # tensor (a rolling episode's last window continues the roller's tour) instead of an own (B, steps) one
        self._tour_buf, col0 = tour, int(tour_col0)
        if tour is not None:
            if tour.dtype is not torch.int64 or not tour.is_contiguous() or tour.device != dev or tour.dim() != 2 or \
                    tour.shape[0] != self.B or col0 < 0 or col0 + self.steps > tour.shape[1]:
                raise ValueError("tour must be a contiguous (B, >= tour_col0 + steps) int64 tensor on %s" % (dev,))
            self.tour = tour[:, col0:col0 + self.steps]
        else:
            self.tour = self._tour_buf = torch.zeros(self.B, self.steps, dtype=torch.int64, device=dev) if want_tour else None
        cnt = self._nonbinary = torch.zeros(1, dtype=torch.int32, device=dev)   # read by begin() / check_binary()
        _steppers.add(self)
        buf = _lib.StepperBuffers()
        for w in range(2):
            buf.bits[w] = self._bits[w].data_ptr() if not self._copy else None
            buf.dyn[w] = self._dyn[w].data_ptr() if expand_dynamic else None
            buf.colsum[w] = self._colsum[w].data_ptr() if self._copy else None
            buf.current[w], buf.mask[w] = self._cur[w].data_ptr(), self._mask[w].data_ptr()
        buf.feature, buf.decoder_static = self.decoder_dynamic.data_ptr(), self.decoder_static.data_ptr()
        buf.ratio = self.ratio.data_ptr()
        buf.tour = self._tour_buf.data_ptr() if self._tour_buf is not None else None
        buf.tour_stride = int(self._tour_buf.shape[1]) if self._tour_buf is not None else 0
        buf.tour_col0 = col0 if tour is not None else 0
        buf.nonbinary = cnt.data_ptr()
        self._ctx = _lib.ctx(dev)
        L = _lib.lib()
        h = C.c_void_p()
        _lib.check(L.tap_stepper_create(self._ctx, C.byref(env.desc), _lib.ptr(env._state), self.n, self.R, self.rows,
                                        self.update_rows, self.static_rows, self.steps, C.byref(buf), C.byref(h)), self._ctx)
        self._h = h
        self._destroy = L.tap_stepper_destroy
        self._step_fn, self._begin_fn, self._begin_shadow_fn = L.tap_stepper_step, L.tap_stepper_begin, L.tap_stepper_begin_shadow
        self._views = [(self._dyn[w], self._cur[w], self._mask[w]) for w in range(2)]
        self._ones = None
        self.static = self.dynamic = self.current_mask = self.mask = self.dynamic_bits = None
        self.k = 0
        self._count_in_step0 = False

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            self._destroy(h)

    @property
    def launches_per_step(self):
        """1 when a step of this shape is ONE kernel, 2 when the entry point runs the precedence update and the
        placement as two launches (tapenv.h: tap_transition_launches)."""
        import ctypes as C
        return int(_lib.lib().tap_transition_launches(self._ctx, C.byref(self.env.desc), self.n, self.R, self.rows, 1))

    def _check_instances(self, static, what, shape):
        if (static.dtype is not torch.float32 or not static.is_contiguous() or static.device != self._dev or
                tuple(static.shape) != shape):
            raise ValueError("%s must be a contiguous float32 tensor of shape %s on %s" % (what, shape, self._dev))

    def begin(self, static, dynamic, initial_mask=True, keep_container=False, current_mask=None):
        """Bind the next instance batch.  ``initial_mask``: one launch builds the shadow and the masks of
        model.py:297-307 (``current_mask`` is then what the policy sees before step 0); False: no launch, step 0
        reads ``dynamic`` itself and ``current_mask`` / ``mask`` are None until then (a replayed tour needs neither).
        ``keep_container``: step 0 goes on with the containers as they are."""
        self._check_instances(static, "static", (self.B, self.static_rows, self.nR))
        self._check_instances(dynamic, "dynamic", (self.B, self.rows, self.nR))
        flags = (_lib.TAP_SB_INITIAL_MASK if initial_mask else 0) | (_lib.TAP_SB_CONTINUE if keep_container else 0)
        rc = self._begin_fn(self._h, static.data_ptr(), dynamic.data_ptr(), flags, _lib.raw_stream(self._idx))
        if rc:
            _lib.check(rc, self._ctx)
        self.static, self.dynamic, self.k = static, dynamic, 0
        self._count_in_step0 = not initial_mask and not self._copy
        self._dec.zero_()
        if self._copy:                                        # begin always makes its launches there: masks and the count exist
            self.current_mask, self.mask = self._cur[1], self._mask[1]
            if _binary_mode == 'check':
                self._raise_nonbinary()
        elif initial_mask:
            self.current_mask, self.mask = self._cur[1], self._mask[1]
            if _binary_mode == 'check':
                self._raise_nonbinary()
        else:
            self.current_mask, self.mask = current_mask, None     # the caller's, if it has one (rolling windows)
        return self

    def begin_shadow(self, static, dynamic, bits, current_mask=None, keep_container=False):
        """The same for a window whose bit shadow the caller holds (rolling windows emit it next to the tensor):
        no launch; ``dynamic`` / ``current_mask`` are only recorded as the values before step 0."""
        self._check_instances(static, "static", (self.B, self.static_rows, self.nR))
        if self._copy:
            raise ValueError("this window has no bit shadow: begin(static, dynamic)")
        if bits.dtype is not torch.int64 or not bits.is_contiguous() or bits.device != self._dev or \
                bits.numel() != self._bits[0].numel():
            raise ValueError("bits must be the (B, %d) int64 shadow of the window" % (self._bits[0].shape[1],))
        rc = self._begin_shadow_fn(self._h, static.data_ptr(), bits.data_ptr(), _lib.TAP_SB_CONTINUE if keep_container else 0)
        if rc:
            _lib.check(rc, self._ctx)
        if self._ones is None:
            self._ones = torch.ones(self.B, self.nR, dtype=torch.float32, device=self._dev)
        self._count_in_step0 = False                      # the caller's shadow: nothing is counted
        self.static, self.dynamic, self.k, self._bits0 = static, dynamic, 0, bits
        self.current_mask, self.mask = current_mask, self._ones
        return self

    def step(self, ptr):
        """One decoding step (model.py:376-465): ptr (B,) int64 on the device -> the phase (0 | 1) the step wrote.
        Afterwards ``dynamic``, ``current_mask``, ``mask``, ``decoder_dynamic``, ``decoder_static`` are the loop
        variables of model.py after that step; ``ratio`` is valid after the last step, ``tour[:, k]`` = ptr."""
        if ptr.dtype is not torch.int64 or not ptr.is_contiguous() or ptr.device != self._dev or ptr.numel() != self.B:
            ptr = ptr.to(device=self._dev, dtype=torch.int64).contiguous()
            if ptr.numel() != self.B:
                raise ValueError("ptr must be (%d,), got %s" % (self.B, tuple(ptr.shape)))
        rc = self._step_fn(self._h, ptr.data_ptr(), _lib.raw_stream(self._idx))
        if rc:
            _lib.check(rc, self._ctx)
        w = self.k & 1
        self.k += 1
        self.dynamic, self.current_mask, self.mask = self._views[w]
        self.dynamic_bits = self._bits[w]
        if self.k == 1 and self._count_in_step0 and _binary_mode == 'check':
            # begin(initial_mask=False) made no launch: step 0 built the shadow and counted the elements that are neither
            # 0 nor 1 -- under 'check' that count is read here (one host sync per episode, as begin() does otherwise)
            self._raise_nonbinary()
        return w

    def _raise_nonbinary(self):
        bad = int(self._nonbinary.item())
        if bad:
            self._nonbinary.zero_()
            raise NonBinaryDynamic("%d element(s) of `dynamic` are neither 0 nor 1: no bit shadow" % bad)

    def check(self):
        """Synchronises: container errors (TapOverflowError like the reference's IndexError) and the count of
        non-0/1 elements seen since the last check (NonBinaryDynamic)."""
        self.env.check()
        self._raise_nonbinary()


def _reward_mul(static, tour_indices, reward_type, input_type, allow_rot, container_width, container_height):
    """The two-container branch of pack.reward (pack.py:451-466): the last row of ``static`` is each
    block's target id; the blocks of id 0 and of id 1 are packed separately, in tour order, each into
    its own empty container, and the score is the mean of the two (0 for an empty list)."""
    import ctypes as C
    block_dim = _block_dim(static, input_type)
    R = _rotate_types(block_dim, allow_rot)
    st = _f32c(static)
    B, rows, nR = st.shape
    n = nR // R
    tour = tour_indices.to(torch.int64)
    idx = tour.unsqueeze(1).repeat(1, rows, R)                                      # pack.py:438
    sample = torch.gather(st, 2, idx)[:, :, :n]                                     # pack.py:441
    blocks = sample[:, 1:1 + block_dim, :].transpose(1, 2).to(torch.int32)         # pack.py:449-450, astype(int)
    ids = sample[:, -1, :]
    cs = [container_width, container_height] if block_dim == 2 else \
        [container_width, container_width, container_height]
    desc = _lib.make_desc(B, cs, n, reward_type, 'full', 'LB_GREEDY')
    c = _lib.ctx(st.device)
    scores = []
    for target in (0, 1):
        # a zero side marks "not in this list" for tap_pack_blocks
        mine = (blocks * (ids == target).unsqueeze(2).to(torch.int32)).contiguous()
        s64 = torch.empty(B, dtype=torch.float64, device=st.device)
        with torch.cuda.device(st.device):
            _lib.check(_lib.lib().tap_pack_blocks(c, C.byref(desc), B, n, _lib.ptr(mine), None, None, None,
                                                  _lib.ptr(s64), _lib.stream_of(st.device)), c)
        scores.append(s64)
    return -((scores[0] + scores[1]) / 2).to(torch.float32)                          # pack.py:466, 473


def reward(static, tour_indices, reward_type, input_type, allow_rot, container_width, container_height,
           packing_strategy='LB_GREEDY'):
    """pack.reward (pack.py:378-473): pack every env's blocks in tour order from an empty
    container and return ``-scores`` (un-normalised C+P+S, tools.py:2442-2449), one launch."""
    if packing_strategy in ('MACS', 'MUL'):
        # pack.py:431 names tools.calc_positions_mus, which does not exist in the reference
        raise AttributeError("module 'tools' has no attribute 'calc_positions_mus'")
    if input_type in ('mul', 'mul-with'):
        return _reward_mul(static, tour_indices, reward_type, input_type, allow_rot, container_width, container_height)
    block_dim = _block_dim(static, input_type)
    R = _rotate_types(block_dim, allow_rot)
    st = _f32c(static)
    B, rows, nR = st.shape
    n = nR // R                                                                    # pack.py:438
    tour = tour_indices.to(torch.int64).contiguous()
    steps = tour.shape[1]
    if steps < n:
        raise ValueError("tour shorter than blocks_num")
    if steps > n:
        tour = tour[:, :n].contiguous()                                            # pack.py:444 [:,:,:n]
    cs = [container_width, container_height] if block_dim == 2 else \
        [container_width, container_width, container_height]                       # pack.py:408-411
    desc = _lib.make_desc(B, cs, n, reward_type, 'full', 'LB_GREEDY')
    out = torch.empty(B, dtype=torch.float32, device=st.device)
    import ctypes as C
    c = _lib.ctx(st.device)
    try:
        # one launch for every container size: lane-per-cell groups up to 64 cells, one wavefront per container with the
        # height-map in LDS above (big.hip: k_big_wave_episode)
        with torch.cuda.device(st.device):
            _lib.check(_lib.lib().tap_episode_reward(c, C.byref(desc), B, n, _lib.ptr(st), rows, nR,
                                                     _lib.ptr(tour), _lib.ptr(out), None, None,
                                                     _lib.stream_of(st.device)), c)
        return out
    except _lib.TapError as e:
        if e.status != _lib.TAP_E_UNSUPPORTED or not _beyond_lane_kernels(desc):
            raise
    # no whole-episode kernel for this shape (TAP_NO_WAVE_KERNELS): the same episode, one placement launch per block;
    # tools.calc_positions_lb_greedy returns C + P + S un-normalised (tools.py:2442-2449)
    from .env import BatchedContainer
    env = BatchedContainer(B, cs, n, reward_type, 'full', packing_strategy='LB_GREEDY', device=st.device)
    for t in range(n):
        env.add_new_blocks_gather(st, tour[:, t].contiguous(), want_feature=False)
    cps = env.calc_CPS()
    return -((cps[:, 0] + cps[:, 1]) + cps[:, 2]).to(torch.float32)


# ---- pack.render (pack.py:670-977): the metric files of a test run ---------------------------------------------

_LBG_RATIO_TYPES = ('C+P-lb-soft', 'C+P-lb-hard', 'C+P+S-lb-soft', 'C+P+S-lb-hard')                # tools.py:2442-2445
_MCS_RATIO_TYPES = ('comp', 'soft', 'hard', 'pyrm', 'pyrm-soft', 'pyrm-hard', 'mcs-soft', 'mcs-hard',  # tools.py:3285-3306
                    'pyrm-soft-sum', 'pyrm-soft-SUM', 'pyrm-hard-sum', 'pyrm-hard-SUM', 'CPS',
                    'C+P-mul-soft', 'C+P-mul-hard', 'C+P-mcs-soft', 'C+P-mcs-hard',
                    'C+P+S-mul-soft', 'C+P+S-mul-hard', 'C+P+S-mcs-soft', 'C+P+S-mcs-hard')
_NET_REWARD_TYPES = ('C+P+S-SL-soft', 'C+P+S-RL-soft', 'C+P+S-G-soft', 'C+P+S-LG-soft')             # pack.py:728-730
RENDER_FILES = ('ratio', 'valid_size', 'box_size', 'empty_size', 'stable_num', 'packing_height', 'time', 'ids')


def _beyond_lane_kernels(desc):
    """containers the lane-per-cell kernels do not take: above 64 cells, or a 3D side above 8 (tap_common.h: tap_is_big*)"""
    return desc.W * desc.L > 64 or (desc.D == 3 and (desc.W > 8 or desc.L > 8))


def _stepped_scores(st, tour, container_size, n, reward_type, strategy, target, check=True):
    """The whole-episode figures for shapes the one-launch kernels do not take (LB_GREEDY and MACS 3D above 64 cells or
    with a 3D side above 8, MACS 2D above 64 columns): the same
    episode as n placement launches on a state blob, ``active`` selecting one container's blocks.  ``check`` as in
    episode_scores: raise like the reference (one host sync), or report per container -- NaN ratio where the error
    word is set, like the one-launch path."""
    from .env import BatchedContainer
    B = st.shape[0]
    env = BatchedContainer(B, container_size, n, reward_type, 'full', packing_strategy=strategy, device=st.device)
    for t in range(n):
        col = tour[:, t].contiguous()
        act = None
        if target is not None:
            act = torch.gather(st[:, -1, :], 1, col.unsqueeze(1)).squeeze(1) == float(target)
        env.add_new_blocks_gather(st, col, active=act, want_feature=False)
    if check:
        env.check()
    cnt = env.counters.to(torch.int64)
    max_h = env.heightmap.reshape(B, -1).max(dim=1).values.to(torch.int64)
    box = max_h * int(np.prod(container_size[:-1]))
    valid, empty, nst, count = (cnt[:, k] for k in range(4))
    C = valid.double() / box.double()
    P = valid.double() / (empty + valid).double()
    S = nst.double() / count.double()
    live = count > 0
    mode = env.desc.ratio_mode if strategy == 'MACS' else _lib.TAP_R_CPS           # tools.py:2442-2445 / :3285-3308
    by_mode = {_lib.TAP_R_C: lambda: C, _lib.TAP_R_CxS: lambda: C * S, _lib.TAP_R_CP: lambda: C + P,
               _lib.TAP_R_CPxS: lambda: (C + P) * S, _lib.TAP_R_2CPS: lambda: (2 * C + P) + S,
               _lib.TAP_R_CxPxS: lambda: (C * P) * S}
    ratio = torch.where(live, by_mode.get(mode, lambda: (C + P) + S)(), torch.zeros_like(C))
    if not check:                                     # containers that raised an error bit report NaN (episode_finish)
        err = torch.empty(B, dtype=torch.int32, device=st.device)
        env._call(_lib.lib().tap_env_errors, _lib.ptr(env._state), _lib.ptr(err))
        ratio = torch.where(err != 0, torch.full_like(ratio, float('nan')), ratio)
    scores = torch.stack((valid, torch.where(live, box, torch.zeros_like(box)), empty, nst, max_h), 1)
    return ratio, scores


def episode_scores(static, tour_indices, reward_type, input_type, allow_rot, container_size, packing_strategy='LB_GREEDY',
                   target=None, check=True):
    """tools.calc_positions_lb_greedy (tools.py:2393-2449) / tools.calc_positions_mcs (tools.py:3213-3315) for
    every sample of a batch in one launch (MACS / MUL containers above 64 cells: one placement launch per block,
    _stepped_scores): blocks in tour order into an empty container.
    -> (ratio (B,) float64, scores (B, 5) int64 = valid_size, box_size, empty_size, stable_num, max height).
    ``target`` 0 | 1: only the blocks whose target id (last row of ``static``, the two-container input types) equals
    it; an empty list scores zeros (pack.py:760-769).  ``check``: raise like the reference when a container
    overflowed (one host read of the error words)."""
    import ctypes as C
    block_dim = _block_dim(static, input_type)
    R = _rotate_types(block_dim, allow_rot)
    st = _f32c(static)
    B, rows, nR = st.shape
    n = nR // R
    tour = tour_indices.to(device=st.device, dtype=torch.int64)
    if tour.shape[1] < n:
        raise ValueError("tour shorter than blocks_num")
    tour = tour[:, :n].contiguous()                                                # pack.py:693 [:,:,:blocks_num]
    mcs = packing_strategy in ('MACS', 'MUL')
    if reward_type not in (_MCS_RATIO_TYPES if mcs else _LBG_RATIO_TYPES):
        # tools.py:2446 / :3308 print 'Unknown reward type' and fall into `return ... ratio` with ratio unbound
        raise UnboundLocalError("local variable 'ratio' referenced before assignment (reward_type %r is not one %s "
                                "scores)" % (reward_type, 'calc_positions_mcs' if mcs else 'calc_positions_lb_greedy'))
    strategy = 'MACS' if mcs else 'LB_GREEDY'
    desc = _lib.make_desc(B, container_size, n, reward_type, 'full', strategy)
    if mcs and _beyond_lane_kernels(desc):
        # MACS / MUL beyond the whole-episode kernels' container size (above 64 cells -- 2D: 64 columns -- or a 3D side above 8)
        return _stepped_scores(st, tour, list(container_size), n, reward_type, strategy, target, check)
    ratio = torch.empty(B, dtype=torch.float64, device=st.device)
    scores = torch.empty(B, 5, dtype=torch.int64, device=st.device)
    err = torch.empty(B, dtype=torch.int32, device=st.device)
    c = _lib.ctx(st.device)
    try:
        with torch.cuda.device(st.device):
            _lib.check(_lib.lib().tap_episode_scores(c, C.byref(desc), B, n, _lib.ptr(st), rows, nR, _lib.ptr(tour),
                                                     -1 if target is None else int(target), _lib.ptr(ratio), _lib.ptr(scores),
                                                     None, None, _lib.ptr(err), _lib.stream_of(st.device)), c)
    except _lib.TapError as e:
        if e.status != _lib.TAP_E_UNSUPPORTED or not _beyond_lane_kernels(desc):
            raise
        return _stepped_scores(st, tour, list(container_size), n, reward_type, strategy, target, check)
    if check:
        bits = int(torch.bitwise_or(err, 0).max().item()) if B else 0
        if bits:
            bad = int((err != 0).sum().item())
            if int((err & 1).max().item()):
                raise _lib.TapOverflowError(_lib.TAP_E_OVERFLOW, "%d container(s) exceeded height H=%d" % (bad, desc.H))
            raise _lib.TapError(_lib.TAP_E_INVALID, "%d container(s) raised error bits 0x%x (tapenv.h: tap_env_check)" % (bad, int(err.max().item())))
    return ratio, scores


def render(static, tour_indices, save_path, dynamic, valid_time, **kwargs):
    """pack.render (pack.py:670-977), the ``render_fn`` of a test run (trainer.py:493, called by validate,
    trainer.py:132): re-pack every sample in tour order with the whole-episode function of the packing strategy
    (pack.py:726-734, 792) and write the eight metric files next to ``save_path`` (pack.py:967-977; the drawing
    code of the reference is commented out, its only other effect is an empty matplotlib figure per sample).  One
    launch per container list instead of a Python loop over samples; the two-container input types pack the blocks of
    each target id into their own container and average the six figures (pack.py:754-776).  Values and file bytes
    equal the reference's (np.savetxt of float64).  ``dynamic`` is unused, as in the reference."""
    input_type = kwargs['input_type']
    if input_type not in _UPDATE_ROWS:
        print('Render OHHHH')                                                       # pack.py:688 (then NameError)
        raise NameError("name 'block_dim' is not defined")
    block_dim = _block_dim(static, input_type)
    mul = input_type in ('mul', 'mul-with')
    unit = kwargs['unit']
    container_width = int(np.ceil(kwargs['container_width'] * unit))                # pack.py:703-711
    container_height = int(np.ceil(kwargs['container_height'] * unit))
    initial_container_height = int(np.ceil(kwargs['initial_container_height'] * unit))
    if block_dim == 3:
        container_size = [container_width, container_width, container_height]
        container_size_ab = [container_width, container_width, initial_container_height]   # pack.py:718-721 (sic)
    else:
        container_size = [container_width, container_height]
        container_size_ab = [container_width, container_height]
    strategy = kwargs['packing_strategy']
    reward_type = kwargs['reward_type']
    if strategy not in ('MACS', 'MUL') and reward_type in _NET_REWARD_TYPES:
        raise NotImplementedError("tools.calc_positions_net (the pack-net back-ends) is outside this package")
    args = (static, tour_indices, reward_type, input_type, kwargs['allow_rot'])
    if mul:
        ra, sa = episode_scores(*args, container_size_ab, strategy, target=0)
        rb, sb = episode_scores(*args, container_size_ab, strategy, target=1)
        ratio = (ra + rb) / 2                                                       # pack.py:771-776
        scores = (sa.double() + sb.double()) / 2
    else:
        ratio, scores = episode_scores(*args, container_size, strategy)
        scores = scores.double()
    ratio = ratio.cpu().numpy()
    scores = scores.cpu().numpy()
    stem = save_path[:-13]                                                          # pack.py:967 (sic)
    np.savetxt(stem + '-ratio.txt', ratio)
    for k, name in enumerate(('valid_size', 'box_size', 'empty_size', 'stable_num', 'packing_height')):
        np.savetxt(stem + '-%s.txt' % name, scores[:, k])
    np.savetxt(stem + '-time.txt', np.array([valid_time]))
    np.savetxt(stem + '-ids.txt', tour_indices.cpu().numpy())


class PACKDataset(Dataset):
    """pack.PACKDataset (pack.py:25-273): the reference's six text files -> the four tensors of
    ``__getitem__``.  Layouts: static (N, 1+D, n*R) with row 0 = block id and
    column r*n+i = block i in rotation r; dynamic (N, 3n, n*R) = cat(move, small, large)."""

    def __init__(self, data_file, blocks_num, num_samples, seed, input_type, heightmap_type, allow_rot,
                 container_width, mix_data_file=None, unit=1, no_precedence=False):
        super(PACKDataset, self).__init__()
        if seed is None:
            seed = np.random.randint(123456)
        np.random.seed(seed)
        torch.manual_seed(seed)
        n = int(blocks_num)

        def load(prefix, name):
            return np.loadtxt(prefix + name + '.txt').astype('float32')

        names = ('dep_move', 'dep_small', 'dep_large', 'blocks', 'pos', 'container')
        arr = {k: load(data_file, k) for k in names}
        if mix_data_file is not None:                                              # pack.py:67-97
            half = int(num_samples / 2)
            rot_half = int(len(arr['blocks']) / 2)
            mix = {k: load(mix_data_file, k) for k in names}
            for k in ('dep_move', 'pos', 'container'):
                arr[k] = np.vstack((arr[k][:half], mix[k][:half]))
            for k in ('dep_small', 'dep_large', 'blocks'):
                arr[k] = np.vstack((arr[k][:rot_half], mix[k][:rot_half]))

        N = int(num_samples)
        positions = torch.from_numpy(arr['pos']).view(N, -1, n)
        D = positions.shape[1]
        R_all = math.factorial(D)
        deps_move = torch.from_numpy(arr['dep_move']).view(N, -1, n).transpose(2, 1)
        # blocks.txt: R lines per sample, each dimension-major (D*n)  -> (N, D, R*n), column r*n+i
        blocks = arr['blocks'].reshape(N, -1, D, n).transpose(0, 1, 3, 2).reshape(N, -1, D).transpose(0, 2, 1)
        blocks = (torch.from_numpy(np.ascontiguousarray(blocks)) * unit).ceil()    # pack.py:123-125

        def rot_deps(a):                                                           # pack.py:127-136
            a = a.reshape(N, -1, n, n).transpose(0, 1, 3, 2).reshape(N, n * R_all, n).transpose(0, 2, 1)
            return torch.from_numpy(np.ascontiguousarray(a))

        small, large = rot_deps(arr['dep_small']), rot_deps(arr['dep_large'])
        R = R_all
        if not allow_rot:                                                          # pack.py:140-142
            blocks = blocks[:, :, :n]
            R = 1
        index = torch.arange(n).view(1, 1, n).repeat(N, 1, R).float()              # pack.py:144-147
        container_index = torch.from_numpy(arr['container']).unsqueeze(1).repeat(1, 1, R).float()
        deps_move = deps_move.repeat(1, 1, R)
        if no_precedence:                                                          # pack.py:175-178
            deps_move, small, large = (torch.zeros_like(t) for t in (deps_move, small, large))

        if input_type in ('simple', 'rot'):
            self.static, self.dynamic = torch.cat((index, blocks), 1), deps_move
        elif input_type == 'bot':
            self.static = torch.cat((index, blocks), 1)
            self.dynamic = torch.cat((deps_move, small, large), 1)
        elif input_type in ('bot-rot', 'use-static', 'use-pnet'):
            self.static = torch.cat((index, blocks), 1)
            self.dynamic = torch.cat((deps_move, torch.zeros_like(small), torch.zeros_like(large)), 1)
        elif input_type in ('mul', 'mul-with'):
            self.static = torch.cat((index, blocks, container_index), 1)
            self.dynamic = torch.cat((deps_move, small, large), 1)
        elif input_type == 'rot-old':
            self.static = torch.cat((index, blocks), 1)
            self.dynamic = torch.cat((deps_move, torch.zeros_like(index)), 1)
        else:
            raise ValueError("unknown input_type %r" % (input_type,))
        self.static = self.static.contiguous()
        self.dynamic = self.dynamic.contiguous()

        # decoder inputs (pack.py:231-264)
        static_dim = D + (1 if input_type == 'mul-with' else 0)
        hm_num = 1
        if heightmap_type == 'diff':
            hm_w = container_width * unit - 1 if D == 2 else container_width * unit
            hm_num = 2 if D == 3 else 1
        else:
            hm_w = container_width * unit
        hm_l = int(np.ceil(container_width * unit))
        hm_w = int(np.ceil(hm_w))
        if input_type in ('mul', 'mul-with'):
            if D == 2:
                hm_w *= 2
            else:
                hm_num *= 2
        self.decoder_static = torch.zeros(N, static_dim, 1, requires_grad=True)
        if D == 2:
            self.decoder_dynamic = torch.zeros(N, hm_w, 1, requires_grad=True)
        else:
            self.decoder_dynamic = torch.zeros(N, hm_num, hm_w, hm_l, requires_grad=True)
        self.num_samples = N

    def __len__(self):
        return self.num_samples

    def __getitem__(self, idx):
        return (self.static[idx], self.dynamic[idx], self.decoder_static[idx], self.decoder_dynamic[idx])


# ---- dataset creation (pack.py:475-667): the entry points trainer.py calls before it builds PACKDataset ---------

def _dataset_dirs(kind, blocks_num, train_size, valid_size, obj_dim, initial_container_width, size_range):
    def one(split, size):
        return './data/%s_%dd/pack-%s-%d-%d-%d-%d-%d/' % (kind, obj_dim, split, blocks_num, size, initial_container_width,
                                                          size_range[0], size_range[1])
    return one('train', train_size), one('valid', valid_size)


def _have(data_dir):
    return os.path.exists(data_dir + 'blocks.txt')


def create_dataset(blocks_num, train_size, valid_size, obj_dim, initial_container_width, initial_container_height,
                   arm_size, size_range, seed=None, device='cuda'):
    """pack.create_dataset (pack.py:580-667): RAND instances -- random blocks packed into the initial container
    with hard LB_GREEDY, kept when every block is stable, precedence extracted from the packing
    (generate.generate_blocks) -- written as the reference's six text files under the reference's directory
    names; existing directories are reused.  -> (train_dir, valid_dir).  The instances come from the device-side
    generator (generate.generate_instances), the files from datafiles.write_dataset.
    ``initial_container_width`` <= -1 (the reference's container-free random-dependency sets,
    generate.generate_deps_prob) is not provided."""
    from . import generate, datafiles
    blocks_num = int(blocks_num)
    if initial_container_width <= -1:
        raise NotImplementedError("random-dependency data sets without an initial container (generate_deps_prob)")
    if seed is None:
        seed = np.random.randint(123456789)
    np.random.seed(seed)                                                          # pack.py:592
    train_dir, valid_dir = _dataset_dirs('rand', blocks_num, train_size, valid_size, obj_dim, initial_container_width, size_range)
    for k, (data_dir, size) in enumerate(((train_dir, train_size), (valid_dir, valid_size))):
        if _have(data_dir):
            continue
        static, dynamic, _, positions = generate.generate_instances(
            int(size), blocks_num, obj_dim, initial_container_width, initial_container_height, arm_size, tuple(size_range),
            seed=(int(seed) + 7919 * k) % (2 ** 31), device=device, return_aux=True)
        ids = np.random.RandomState((int(seed) + k) % (2 ** 31)).randint(0, 2, size=(int(size), blocks_num))  # pack.py:652
        datafiles.write_dataset(data_dir, static, dynamic, positions, container_ids=ids)
    return train_dir, valid_dir


def create_dataset_gt(blocks_num, train_size, valid_size, obj_dim, target_container_width, target_container_height,
                      initial_container_width, initial_container_height, input_type, arm_size, size_range, seed=None,
                      device='cuda'):
    """pack.create_dataset_gt (pack.py:475-566): perfect-packing (PPSG) instances -- generate_blocks_with_GT with
    the height of the perfect packing drawn per sample from generate_height_prob's distribution -- as the six text
    files under the reference's directory names.  -> (train_dir, valid_dir).  Device generators:
    generate.generate_ppsg_instances_2d / generate_ppsg_instances (see their notes on which heights and block
    counts the reference's acceptance loops can reach).  One container-id vector per data set, half zeros, shuffled
    (pack.py:516-518)."""
    from . import generate, datafiles
    blocks_num = int(blocks_num)
    if int(arm_size) < 1:
        raise ValueError("arm_size >= 1")
    if seed is None:
        seed = np.random.randint(123456789)
    np.random.seed(seed)                                                          # pack.py:486
    train_dir, valid_dir = _dataset_dirs('gt', blocks_num, train_size, valid_size, obj_dim, initial_container_width, size_range)
    rs = np.random.RandomState(int(seed) % (2 ** 31))
    for k, (data_dir, size) in enumerate(((train_dir, train_size), (valid_dir, valid_size))):
        ids = np.ones(blocks_num, dtype=np.int64)
        ids[:blocks_num // 2] = 0
        rs.shuffle(ids)
        if _have(data_dir):
            continue
        kw = dict(seed=(int(seed) + 7919 * k) % (2 ** 31), device=device, input_type=input_type, arm_size=int(arm_size))
        if obj_dim == 2:
            # the height restriction of generate_ppsg_instances_2d's default is tuned for size_range (1, 5) at 20 blocks;
            # any other request draws from the reference's whole height distribution
            area = (4.9, 8.4) if tuple(size_range) == (1, 5) and blocks_num == 20 else None
            blocks, positions = generate.generate_ppsg_instances_2d(int(size), blocks_num, initial_container_width,
                                                                    initial_container_height, target_container_width,
                                                                    tuple(size_range), mean_block_area=area, **kw)
            note = ("2D perfect-packing instances of tap-net_amd's device generator (generate_blocks_with_GT's steps)%s\n"
                    % ("; heights restricted to mean block areas 4.9 .. 8.4 (14 .. 24 at W = 7: 90 % of generate_height_prob)"
                       if area else ""))
        else:
            blocks, positions = generate.generate_ppsg_instances(int(size), blocks_num, initial_container_width,
                                                                 initial_container_height, target_container_width,
                                                                 tuple(size_range), **kw)
            note = ("3D perfect-packing instances of tap-net_amd's device generator: stacked 10-block BPP_Generator_3D "
                    "packings (the reference's single acceptance loop does not reach more than ~10 blocks)\n")
        cs = generate.initial_container(obj_dim, initial_container_width, initial_container_height)
        static, dynamic = generate.precedence_tensors(blocks, positions, cs, arm_size)
        datafiles.write_dataset(data_dir, static, dynamic, positions, container_ids=np.tile(ids, (int(size), 1)))
        with open(data_dir + 'GENERATOR.txt', 'w') as f:                           # tells these files from ones the reference wrote
            f.write(note)
    return train_dir, valid_dir


def get_mix_dataset(blocks_num, train_size, valid_size, obj_dim, initial_container_width, size_range, seed=None):
    """pack.get_mix_dataset (pack.py:568-578): the directory names of the PPSG and RAND sets a MIX run reads --
    names only, nothing is generated.  (sic) the reference returns the 2D directories for obj_dim 3 as well."""
    blocks_num = int(blocks_num)
    g = _dataset_dirs('gt', blocks_num, train_size, valid_size, 2, initial_container_width, size_range)
    r = _dataset_dirs('rand', blocks_num, train_size, valid_size, 2, initial_container_width, size_range)
    return g[0], r[0], g[1], r[1]


def rotation_permutations(block_dim):
    """Rotation r permutes a block's sides by itertools.permutations(range(D))[r] (generate.py:953-960)."""
    return list(itertools.permutations(range(block_dim)))
