"""Rolling precedence windows -- the batched counterpart of ``generate.InitialContainer``
(generate.py:1589-1839) and of the outer loop of ``rolling.validate`` (rolling.py:589-637), which
packs instances far larger than the policy's 10-node input by re-cutting a 10-node sub-graph of the
precedence DAG after every placement (batch size 1, networkx, in the reference).
"""
import ctypes as C

import torch

from . import _lib
from .env import BatchedContainer
from .pack import EnvTransition, EpisodeStepper, bits_supported
from .rollout import _flagged_reward


class RollingWindows(object):
    """B instances of N <= 4096 packed blocks each; ``next(ptr)`` drops the block picked in the
    previous window and returns the next window's network input.  Up to 64 blocks an instance is one
    wavefront (lane = node, graphs = 64-bit masks), up to 128 blocks with windows of at most 32 nodes still one
    wavefront (lane = two nodes, two-word masks), and ``step`` fuses the placement with the next window; above that
    the same steps run one thread per instance on multi-word masks and ``step`` is two launches."""

    def __init__(self, blocks, positions, initial_container_size, child_graph_size=10, arm_size=1):
        self.blocks = blocks.to(torch.int32).contiguous()
        self.device = _lib.resolve_device(self.blocks.device)
        positions = positions.to(device=self.device, dtype=torch.int32).contiguous()
        self.B, self.N, self.D = self.blocks.shape
        self.child = int(child_graph_size)
        self.R = 2 if self.D == 2 else 6
        nw = (self.N + 63) // 64                          # mask words per graph: 1 up to 64 blocks, up to 4 for 256
        self.rel = torch.empty(self.B, 5, self.N * nw, dtype=torch.int64, device=self.device)
        self.state = torch.empty(self.B, 2 * nw, dtype=torch.int64, device=self.device)
        self.steps_done = 0
        cs = (C.c_int32 * self.D)(*[int(v) for v in initial_container_size])
        self._ctx = _lib.ctx(self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().tap_rolling_init(self._ctx, self.B, self.D, self.N, cs, int(arm_size),
                                                   _lib.ptr(self.blocks), _lib.ptr(positions), _lib.ptr(self.rel),
                                                   _lib.ptr(self.state), _lib.stream_of(self.device)), self._ctx)

    @property
    def windows_total(self):
        """Number of convert_to_input() calls of one instance: N - child single-step windows plus the last one."""
        return self.N - self.child + 1

    def is_last_graph(self):
        """InitialContainer.is_last_graph (generate.py:1838-1839) for the window returned last; the
        instances run in lock-step, so this is a host-side count, not a device read."""
        return self.steps_done >= self.windows_total

    def next(self, remove_ptr=None, want_masks=True, expand_dynamic=True):
        """convert_to_input() after remove_block(sub_graph_nodes[ptr % child]).
        -> dict(static, dynamic, nodes, colsum, bits, current_mask): ``bits`` is the window tensor's bit
        shadow (pack.dynamic_bits layout) when its shape has one, else ``colsum`` holds the column sums."""
        if (remove_ptr is None) != (self.steps_done == 0):
            raise ValueError("pass the previous window's pick to every call but the first")
        f32 = dict(dtype=torch.float32, device=self.device)
        nRc = self.child * self.R
        static = torch.empty(self.B, 1 + self.D, nRc, **f32)
        nodes = torch.empty(self.B, self.child, dtype=torch.int32, device=self.device)
        bits = self._new_bits(nRc) if want_masks else None
        dynamic = self._new_dynamic(nRc, bits, expand_dynamic)
        colsum = torch.empty(self.B, 3, nRc, **f32) if want_masks and bits is None else None
        cur = torch.empty(self.B, nRc, **f32) if want_masks else None
        err = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        ptr = None if remove_ptr is None else remove_ptr.to(device=self.device, dtype=torch.int64).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().tap_rolling_window(
                self._ctx, self.B, self.D, self.N, self.child, _lib.ptr(self.blocks), _lib.ptr(self.rel),
                _lib.ptr(self.state), _lib.ptr(ptr), _lib.ptr(static), _lib.ptr(dynamic), _lib.ptr(colsum),
                _lib.ptr(bits), _lib.ptr(cur), _lib.ptr(nodes), _lib.ptr(err), _lib.stream_of(self.device)), self._ctx)
        self.steps_done += 1
        self._err = err
        return dict(static=static, dynamic=dynamic, nodes=nodes, colsum=colsum, bits=bits, current_mask=cur)

    def _new_dynamic(self, nRc, bits, expand_dynamic):
        """The window's fp32 precedence tensor, or None with ``expand_dynamic=False`` (tapenv.h: dynamic_out = NULL --
        the window then only exists as its bit shadow, which its shape must have)."""
        if expand_dynamic:
            return torch.empty(self.B, 3 * self.child, nRc, dtype=torch.float32, device=self.device)
        if bits is None:
            raise ValueError("expand_dynamic=False needs the window's bit shadow (want_masks=True and 3 * child <= 64, "
                             "child * R a multiple of 4)")
        return None

    def _new_bits(self, nRc):
        """Buffer for the window tensor's bit shadow (pack.dynamic_bits layout), None if the shape has none."""
        from .pack import bits_supported
        if 3 * self.child > 64 or not bits_supported(3 * self.child, nRc):   # the window kernels emit the one-word shadow
            return None
        return torch.empty(self.B, nRc, dtype=torch.int64, device=self.device)

    def step(self, ptr, env, static_cur, want_masks=True, want_feature=True, expand_dynamic=True):
        """One decoding step in ONE launch (tap_rolling_step): place the block picked in the current
        window (column ``ptr`` of ``static_cur``) into ``env`` and build the next window.
        -> (feature, next-window dict)."""
        f32 = dict(dtype=torch.float32, device=self.device)
        nRc = self.child * self.R
        static = torch.empty(self.B, 1 + self.D, nRc, **f32)
        nodes = torch.empty(self.B, self.child, dtype=torch.int32, device=self.device)
        bits = self._new_bits(nRc) if want_masks else None
        dynamic = self._new_dynamic(nRc, bits, expand_dynamic)
        colsum = torch.empty(self.B, 3, nRc, **f32) if want_masks and bits is None else None
        cur = torch.empty(self.B, nRc, **f32) if want_masks else None
        err = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        feat = env._new_feature() if want_feature else None
        ptr = ptr.to(device=self.device, dtype=torch.int64).contiguous()
        static_cur = static_cur.contiguous()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().tap_rolling_step(
                self._ctx, C.byref(env.desc), _lib.ptr(env._state), self.N, self.child, _lib.ptr(self.blocks),
                _lib.ptr(self.rel), _lib.ptr(self.state), _lib.ptr(ptr), _lib.ptr(static_cur), _lib.ptr(static),
                _lib.ptr(dynamic), _lib.ptr(colsum), _lib.ptr(bits), _lib.ptr(cur), _lib.ptr(nodes), _lib.ptr(err),
                _lib.ptr(feat), _lib.stream_of(self.device)), self._ctx)
        self.steps_done += 1
        self._err = err
        return feat, dict(static=static, dynamic=dynamic, nodes=nodes, colsum=colsum, bits=bits, current_mask=cur)

    def check(self):
        if int(self._err.sum().item()):
            raise _lib.TapError(_lib.TAP_E_INVALID, "a precedence window could not be filled")


class InitialContainer(object):
    """Drop-in for ``generate.InitialContainer`` (generate.py:1589-1839) as rolling.py drives it (rolling.py:501, 592, 597,
    637): ONE instance, numpy in / numpy out, on a ``RollingWindows`` of batch 1 -- the per-instance facade of this seam, as
    ``tools.Container`` is for S2 (one launch and one sync per call: it exists so that the unchanged loop runs, the batched
    ``RollingWindows`` / ``RollingStepper`` are the fast path).

    ``convert_to_input()`` -> (static (1 + D, child * R), dynamic (3 * child, child * R)) float64 arrays like the
    reference's; afterwards ``sub_graph_nodes`` is the window's node list (sorted: the order of ``static``'s columns, which
    is what rolling.py:637 indexes with the policy's pick).  ``remove_block(block_id)`` drops a node of the current window
    before the next ``convert_to_input()``; an id that is not in the window is ignored, as the reference's ``try`` does
    (generate.py:1832-1836).  The windows' set-order quirk (generate.py:1758-1761) is the kernels' (rolling.hip)."""

    def __init__(self, blocks, positions, blocks_num, initial_container_size, allow_bot=True, child_graph_size=10,
                 input_type='bot', device='cuda', arm_size=1):
        import numpy as np
        if input_type != 'bot' or not allow_bot:
            raise NotImplementedError("rolling windows are built for input_type 'bot' (rolling.py:501 passes allow_bot=True)")
        n = int(blocks_num)
        b = torch.as_tensor(np.asarray(blocks)[:n].astype('int32')).unsqueeze(0)
        p = torch.as_tensor(np.asarray(positions)[:n].astype('int32')).unsqueeze(0)
        dev = _lib.resolve_device(device)
        self._rw = RollingWindows(b.to(dev), p.to(dev), [int(v) for v in initial_container_size], int(child_graph_size), arm_size)
        self.blocks_num, self.child_graph_size, self.input_type = n, int(child_graph_size), input_type
        self.block_dim = self._rw.D
        self.rotate_types = self._rw.R
        self.sub_graph_nodes = []
        self._pending = None

    def remove_block(self, block_id):
        if block_id in self.sub_graph_nodes:
            if self._pending is not None:
                raise NotImplementedError("one block per window step (rolling.py:637 removes the policy's pick)")
            self._pending = self.sub_graph_nodes.index(block_id)

    def convert_to_input(self):
        rw = self._rw
        if rw.steps_done == 0:
            win = rw.next(None)
        else:
            if self._pending is None:
                raise NotImplementedError("convert_to_input() without a removed block: the window is unchanged -- keep the "
                                          "previous tensors (the reference would recompute the same ones)")
            win = rw.next(torch.tensor([self._pending], dtype=torch.int64, device=rw.device))
        self._pending = None
        rw.check()
        self.sub_graph_nodes = [int(v) for v in win['nodes'][0].tolist()]
        return win['static'][0].cpu().numpy().astype('float64'), win['dynamic'][0].cpu().numpy().astype('float64')

    def is_last_graph(self):
        return self._rw.is_last_graph()


class RollingStepper(object):
    """The step object of rolling.validate's loop (tapenv.h: tap_roller) -- ``RollingWindows.step`` for callers that
    pay per step on the host: it OWNS the window buffers (two phases of ``static`` and the node list, one of
    everything else), the feature, ``decoder_static``, the tour and the list of picked global block ids; a step is
    ONE C call with two arguments, whose launch also writes decoder_static, the tour column and the picked id.

    ``begin(windows)`` emits the first window of a freshly initialised ``RollingWindows`` (of the shape given at
    construction); after it and after every ``step(ptr)`` the attributes ``static``, ``dynamic``, ``nodes``,
    ``bits`` / ``colsum``, ``current_mask``, ``decoder_dynamic``, ``decoder_static`` describe the CURRENT window --
    views of the stepper's buffers, overwritten by later steps."""

    def __init__(self, windows, env, want_masks=True, expand_dynamic=True):
        rw = self.windows = windows
        self.env = env
        dev = self._dev = rw.device
        self._idx = dev.index
        B, N, D, child, R = rw.B, rw.N, rw.D, rw.child, rw.R
        if env.batch_size != B or env.block_dim != D:
            raise ValueError("container batch / dimension does not match the instances")
        self.B, self.N, self.D, self.child, self.R = B, N, D, child, R
        nRc = child * R
        f32 = dict(dtype=torch.float32, device=dev)
        self._static = [torch.empty(B, 1 + D, nRc, **f32) for _ in range(2)]
        self._nodes = [torch.empty(B, child, dtype=torch.int32, device=dev) for _ in range(2)]
        self.bits = rw._new_bits(nRc) if want_masks else None
        # expand_dynamic=False: no fp32 precedence tensor (``dynamic`` is None, the windows exist as ``bits`` only)
        self.dynamic = rw._new_dynamic(nRc, self.bits, expand_dynamic)
        self.colsum = torch.empty(B, 3, nRc, **f32) if want_masks and self.bits is None else None
        self.current_mask = torch.empty(B, nRc, **f32) if want_masks else None
        self._err = torch.zeros(B, dtype=torch.int32, device=dev)
        self._ones = torch.ones(B, nRc, **f32)            # the `mask` of a one-step window (rolling.py:353: a fresh forward)
        fshape = env._feature_shape()
        flen = 1
        for v in fshape[1:]:
            flen *= int(v)
        self._dec = torch.zeros(B * (flen + D), **f32)
        self.decoder_dynamic = self._dec[:B * flen].view(fshape)
        self.decoder_static = self._dec[B * flen:].view(B, D, 1)
        self.tour = torch.zeros(B, N, dtype=torch.int64, device=dev)
        self.picked = torch.zeros(B, N, dtype=torch.int32, device=dev)
        buf = _lib.RollerBuffers()
        for w in range(2):
            buf.static_[w], buf.nodes[w] = self._static[w].data_ptr(), self._nodes[w].data_ptr()
        buf.dynamic = self.dynamic.data_ptr() if self.dynamic is not None else None
        buf.bits = self.bits.data_ptr() if self.bits is not None else None
        buf.colsum = self.colsum.data_ptr() if self.colsum is not None else None
        buf.current_mask = self.current_mask.data_ptr() if self.current_mask is not None else None
        buf.err, buf.feature = self._err.data_ptr(), self.decoder_dynamic.data_ptr()
        buf.decoder_static, buf.tour, buf.picked = self.decoder_static.data_ptr(), self.tour.data_ptr(), self.picked.data_ptr()
        buf.tour_stride = N
        self._ctx = _lib.ctx(dev)
        L = _lib.lib()
        h = C.c_void_p()
        _lib.check(L.tap_roller_create(self._ctx, C.byref(env.desc), _lib.ptr(env._state), N, child, C.byref(buf), C.byref(h)),
                   self._ctx)
        self._h, self._destroy, self._step_fn, self._begin_fn = h, L.tap_roller_destroy, L.tap_roller_step, L.tap_roller_begin
        self.static = self.nodes = None
        self.k = -1

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            self._destroy(h)

    def begin(self, windows=None):
        """First window of ``windows`` (default: the RollingWindows given at construction; any freshly initialised
        one of the same shape) -- its ``state`` is consumed by the episode."""
        rw = self.windows if windows is None else windows
        if (rw.B, rw.N, rw.D, rw.child) != (self.B, self.N, self.D, self.child) or rw.device != self._dev:
            raise ValueError("windows of another shape / device than this stepper was built for")
        if rw.steps_done != 0:
            raise ValueError("the windows have been rolled already: build (or re-initialise) a RollingWindows per episode")
        self.windows = rw
        rc = self._begin_fn(self._h, rw.blocks.data_ptr(), rw.rel.data_ptr(), rw.state.data_ptr(), _lib.raw_stream(self._idx))
        if rc:
            _lib.check(rc, self._ctx)
        rw.steps_done = 1
        rw._err = self._err
        self._dec.zero_()
        self.k = 0
        self.static, self.nodes = self._static[0], self._nodes[0]
        return self

    def step(self, ptr):
        """Place the block picked in the current window (column ``ptr`` (B,) int64 of ``static``) into the containers
        and emit the next window (tap_rolling_step): one launch for the single-kernel shapes."""
        if ptr.dtype is not torch.int64 or not ptr.is_contiguous() or ptr.device != self._dev or ptr.numel() != self.B:
            ptr = ptr.to(device=self._dev, dtype=torch.int64).contiguous()
            if ptr.numel() != self.B:
                raise ValueError("ptr must be (%d,), got %s" % (self.B, tuple(ptr.shape)))
        rc = self._step_fn(self._h, ptr.data_ptr(), _lib.raw_stream(self._idx))
        if rc:
            _lib.check(rc, self._ctx)
        self.k += 1
        self.windows.steps_done += 1
        w = self.k & 1
        self.static, self.nodes = self._static[w], self._nodes[w]
        return w

    def is_last_graph(self):
        return self.k >= self.N - self.child

    def check(self):
        if int(self._err.sum().item()):
            raise _lib.TapError(_lib.TAP_E_INVALID, "a precedence window could not be filled")


class RollingDataset(object):
    """rolling.RollingDataset (rolling.py:462-536): the reference's data files of ``total_blocks_num``-block
    instances -> the initial containers ``rolling.validate`` rolls over, here ONE batched ``RollingWindows``
    (``.windows``, windows of ``net_blocks_num`` nodes) instead of a list of per-sample ``InitialContainer``s, plus
    the zero decoder inputs with the reference's shapes.  ``.blocks`` / ``.positions`` (N, total, D) int32 are
    rotation 0 of every block, as the reference hands them to ``InitialContainer``."""

    def __init__(self, data_file, total_blocks_num, net_blocks_num, num_samples, block_dim, seed, input_type,
                 heightmap_type, allow_rot, container_width, initial_container_width, initial_container_height,
                 mix_data_file=None, unit=1, device='cuda'):
        import numpy as np
        if seed is None:
            seed = np.random.randint(123456)
        np.random.seed(seed)
        torch.manual_seed(seed)
        N, n, D = int(num_samples), int(total_blocks_num), int(block_dim)
        blocks, positions = self.read_instances(data_file, n, D, N)
        self.blocks = torch.as_tensor(np.ascontiguousarray(blocks), dtype=torch.int32, device=device)
        self.positions = torch.as_tensor(np.ascontiguousarray(positions), dtype=torch.int32, device=device)
        init = ([initial_container_width, initial_container_height] if D == 2 else
                [initial_container_width, initial_container_width, initial_container_height])       # rolling.py:494-497
        self.initial_container_size = init
        self.windows = RollingWindows(self.blocks, self.positions, init, child_graph_size=int(net_blocks_num))
        static_dim, maps = D, 1                                                          # rolling.py:506-527
        if heightmap_type == 'diff':
            width = container_width * unit - 1 if D == 2 else container_width * unit
            maps = 1 if D == 2 else 2
        else:
            width = container_width * unit
        if input_type in ('mul', 'mul-with'):
            if D == 2:
                width *= 2
            else:
                maps *= 2
        if input_type == 'mul-with':
            static_dim += 1
        self.decoder_static = torch.zeros(1, static_dim, 1, device=device, requires_grad=True)   # rolling.py:529-534
        self.decoder_dynamic = (torch.zeros(1, width, 1, device=device, requires_grad=True) if D == 2 else
                                torch.zeros(1, maps, width, width, device=device, requires_grad=True))
        self.num_samples = N

    @staticmethod
    def read_instances(data_file, total_blocks_num, block_dim, num_samples):
        """blocks.txt / pos.txt -> (blocks, positions) (N, total, D) int64, rotation 0 (rolling.py:475-491)."""
        import numpy as np
        n, D = int(total_blocks_num), int(block_dim)
        blocks = np.loadtxt(data_file + 'blocks.txt', ndmin=2).astype('int64')
        positions = np.loadtxt(data_file + 'pos.txt', ndmin=2).astype('int64')
        R = 2 if D == 2 else 6
        data_size = len(blocks) // R
        blocks = blocks.reshape(data_size, -1, D, n).transpose(0, 1, 3, 2).reshape(data_size, -1, D)
        positions = positions.reshape(len(positions), -1, n).transpose(0, 2, 1)
        return blocks[:num_samples, :n], positions[:num_samples]


def run_rolling_episode(blocks, positions, initial_container_size, policy, container_width, container_height,
                        child_graph_size=10, reward_type='C+P+S-lb-soft', heightmap_type='diff',
                        packing_strategy='LB_GREEDY', record=False, fused=True, steppers=None, check='nan',
                        expand_dynamic=True):
    """rolling.validate's loop for a batch (rolling.py:589-637 around DRL.forward(one_step),
    rolling.py:294-460): N - child windows of ONE decoding step each, then a full episode on the
    last window; one long-lived target container per instance.

    ``policy(step=, static=, dynamic=, current_mask=, mask=, decoder_static=, decoder_dynamic=)``
    -> ptr (B,) int64 as in rollout.run_episode.  Returns dict(tour (B, N) of window-local picks,
    nodes (B, N) global block ids in packing order, reward, env).  ``check`` as in rollout.run_episode: containers
    that raised an error bit report NaN ('nan', default, no host sync), raise ('raise'), or keep the raw ratio (False).

    ``fused`` runs on a RollingStepper + pack.EpisodeStepper pair: per decoding step the policy's call and one C
    call (decoder_static, the tour and the picked ids are written by the step's own launch).  ``steppers``: the
    pair returned by an earlier call (``out['steppers']``) to re-use across episodes -- nothing but the relation
    masks is then allocated per episode, and the returned tensors are views of the pair's buffers.

    ``expand_dynamic=False`` (the step objects only; windows of at most 21 nodes): the windows' precedence tensors are
    never written as fp32 -- the policy gets ``dynamic=None`` and reads the pair's ``bits`` / ``dynamic_bits`` instead
    (7 200 of the 9 937 bytes a c5 step moves are that tensor)."""
    rw = RollingWindows(blocks, positions, initial_container_size, child_graph_size)
    if not expand_dynamic and not (fused and rw._new_bits(rw.child * rw.R) is not None):
        raise ValueError("expand_dynamic=False runs on the step objects (fused=True) and needs windows with a one-word bit shadow")
    if fused and bits_supported(3 * rw.child, rw.child * rw.R):
        return _run_rolling_steppers(rw, policy, container_width, container_height, reward_type, heightmap_type,
                                     packing_strategy, record, steppers, check, expand_dynamic)
    return _run_rolling_eager(rw, policy, container_width, container_height, reward_type, heightmap_type,
                              packing_strategy, record, fused, check)


def _run_rolling_steppers(rw, policy, container_width, container_height, reward_type, heightmap_type, packing_strategy,
                          record, steppers, check='nan', expand_dynamic=True):
    B, N, D, child = rw.B, rw.N, rw.D, rw.child
    dev = rw.device
    if steppers is None:
        cs = [container_width, container_height] if D == 2 else [container_width, container_width, container_height]
        env = BatchedContainer(B, cs, N, reward_type, heightmap_type, packing_strategy=packing_strategy, device=dev)
        roll = RollingStepper(rw, env, expand_dynamic=expand_dynamic)
        last = EpisodeStepper(roll._static[0], (B, 3 * child, child * rw.R), env, steps=child, tour=roll.tour,
                              tour_col0=N - child, expand_dynamic=expand_dynamic)
    else:
        roll, last = steppers
        env = roll.env
        # the pair's buffers fix whether the fp32 tensors exist: a call that asks for the other mode would hand the
        # policy a tensor it did not ask for (or None where it expects one)
        pair_expands = roll.dynamic is not None
        if pair_expands != bool(expand_dynamic) or bool(last.expand_dynamic) != bool(expand_dynamic):
            raise ValueError("steppers were built with expand_dynamic=%s, this call asks for expand_dynamic=%s: "
                             "build a new pair (steppers=None) for the other mode" % (pair_expands, bool(expand_dynamic)))
    env.reset()
    roll.begin(rw)
    feats = []
    step = 0
    for _ in range(N - child):                                   # one_step windows (rolling.py:353-354)
        ptr = policy(step=step, static=roll.static, dynamic=roll.dynamic, current_mask=roll.current_mask,
                     mask=roll._ones, decoder_static=roll.decoder_static, decoder_dynamic=roll.decoder_dynamic)
        roll.step(ptr)                                           # placement + next window: one launch
        if record:
            feats.append(roll.decoder_dynamic.clone())
        step += 1
    assert roll.is_last_graph()                                  # the current window is the last graph: a whole episode on it
    nodes_last = roll.nodes
    if roll.bits is not None:
        last.begin_shadow(roll.static, roll.dynamic, roll.bits, current_mask=roll.current_mask, keep_container=True)
    else:                                                        # two-word shadow: step 0 builds it from the tensor
        last.begin(roll.static, roll.dynamic, initial_mask=False, keep_container=True, current_mask=roll.current_mask)
        last.mask = roll._ones
    last._dec.copy_(roll._dec)                                   # the decoder inputs go on from the last rolling step
    static_last = roll.static
    for t in range(child):
        ptr = policy(step=step, static=static_last, dynamic=last.dynamic, current_mask=last.current_mask,
                     mask=last.mask, decoder_static=last.decoder_static, decoder_dynamic=last.decoder_dynamic)
        last.step(ptr)
        if record:
            feats.append(last.decoder_dynamic.clone())
        step += 1
    tour = roll.tour
    roll.picked[:, N - child:] = torch.gather(nodes_last, 1, (tour[:, N - child:] % child))   # sub_graph_nodes[ptr]
    out = dict(tour_idx=tour, nodes=roll.picked, reward=_flagged_reward(-last.ratio, check, env), env=env, windows=rw, steppers=(roll, last))
    if record:
        out['features'] = feats
    return out


def _run_rolling_eager(rw, policy, container_width, container_height, reward_type, heightmap_type, packing_strategy,
                       record, fused, check='nan'):
    """The same loop on fresh tensors per step (RollingWindows.step / next, pack.EnvTransition): every window shape,
    also the ones without a bit shadow."""
    from . import pack as tpack
    B, N, D, child = rw.B, rw.N, rw.D, rw.child
    dev = rw.device
    cs = [container_width, container_height] if D == 2 else [container_width, container_width, container_height]
    env = BatchedContainer(B, cs, N, reward_type, heightmap_type, packing_strategy=packing_strategy, device=dev)
    decoder_static = torch.zeros(B, D, 1, device=dev)
    decoder_dynamic = torch.zeros(env._feature_shape(), device=dev)
    ar = torch.arange(B, device=dev)
    tour, picked, feats = [], [], []
    # fused: tap_rolling_step per decoding step -- one kernel for LB_GREEDY on lane-per-cell containers and N <= 64, the
    # placement and the window launch behind the same entry point otherwise
    ptr, ratio, step = None, None, 0
    win = rw.next(None)
    for _ in range(N - child):                                   # one_step windows
        ones = torch.ones_like(win['current_mask'])
        ptr = policy(step=step, static=win['static'], dynamic=win['dynamic'], current_mask=win['current_mask'],
                     mask=ones, decoder_static=decoder_static, decoder_dynamic=decoder_dynamic).to(torch.int64)
        decoder_static = torch.gather(win['static'][:, 1:, :], 2, ptr.view(-1, 1, 1).expand(-1, D, 1))
        tour.append(ptr.unsqueeze(1)); picked.append(win['nodes'][ar, ptr % child].unsqueeze(1))
        if fused:                                                # placement + next window: one launch
            decoder_dynamic, win = rw.step(ptr, env, win['static'])
        else:
            decoder_dynamic = env.add_new_blocks_gather(win['static'], ptr)
            win = rw.next(ptr)
        if record:
            feats.append(decoder_dynamic)
        step += 1
    assert rw.is_last_graph()                                    # `win` is the last graph: a whole episode on it
    if win['colsum'] is not None:
        tpack._shadow_put(win['dynamic'], win['colsum'])
    # windows of 22 .. 42 nodes (66 .. 126 rows): the window kernels emit the one-word shadow only, so the last
    # window's episode builds the two-word shadow from the tensor itself (bits=None) instead of copying fp32 slabs
    last_bits = win['bits'] if win['bits'] is not None else (None if tpack.bits_supported(3 * child, child * rw.R) else False)
    trans = EnvTransition(win['static'], win['dynamic'], env, bits=last_bits)
    for t in range(child):
        ptr = policy(step=step, static=trans.static, dynamic=trans.dynamic, current_mask=trans.current_mask,
                     mask=trans.mask, decoder_static=decoder_static, decoder_dynamic=decoder_dynamic).to(torch.int64)
        decoder_static = torch.gather(trans.static[:, 1:, :], 2, ptr.view(-1, 1, 1).expand(-1, D, 1))
        _, _, _, decoder_dynamic, r = trans.step(ptr, fresh=False, want_ratio=(t == child - 1))
        ratio = r if r is not None else ratio
        tour.append(ptr.unsqueeze(1)); picked.append(win['nodes'][ar, ptr % child].unsqueeze(1))
        if record:
            feats.append(decoder_dynamic)
        step += 1
    out = dict(tour_idx=torch.cat(tour, 1), nodes=torch.cat(picked, 1), reward=_flagged_reward(-ratio, check, env), env=env, windows=rw)
    if record:
        out['features'] = feats
    return out
