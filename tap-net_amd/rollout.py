"""The episode loop around the batched environment -- this package's counterpart of the loop in
the reference's ``DRL.forward`` (model.py:342-515), with the policy network factored out.

The reference interleaves its pointer network with a per-env Python loop and two host<->device
copies per step (model.py:407-465).  Here a step is two kernel launches on the current stream
(precedence update, placement) and nothing leaves the device until the caller asks.
"""
import math

import torch

from . import _lib
from .env import BatchedContainer
from .pack import EnvTransition, EpisodeStepper, MaskStepper, NonBinaryDynamic, bits_supported


class TapePolicy(object):
    """Replays a recorded tour (B, steps) -- e.g. the reference actor's greedy choices.  The tour is kept
    step-major, so a step's picks are a contiguous (B,) view: no copy kernel between the steps."""

    def __init__(self, tour):
        self.tour = tour
        self._steps = tour.to(torch.int64).t().contiguous()

    def __call__(self, step, **_):
        return self._steps[step]


class RandomFeasiblePolicy(object):
    """Uniformly random selectable column of ``current_mask``: an exponential race -- the column with the smallest
    Exp(1) draw among the selectable ones, argmax(mask / q) -- which is what torch.multinomial(mask, 1) computes
    after its input checks (three torch ops on buffers the policy keeps, instead of ~85 us of host per
    torch.multinomial call on this stack).  Where NO column is selectable argmax returns column 0 (torch.multinomial
    raises there): ``strict=True`` checks for that, at one host sync per call; a recorded run checks the tour against
    the masks afterwards (tests), a trainer's episodes end before the mask empties."""

    def __init__(self, generator=None, strict=False):
        self.generator = generator
        self.strict = strict          # True: raise when an env has no selectable column, as torch.multinomial does (one host sync per call)
        self._q = self._r = None

    def __call__(self, step, current_mask, **_):
        if self.strict and not bool((current_mask.sum(1) > 0).all()):
            raise RuntimeError("RandomFeasiblePolicy: an env has no selectable column (current_mask is all zero there); "
                               "argmax would return column 0")
        if self._q is None or self._q.shape != current_mask.shape or self._q.device != current_mask.device:
            self._q = torch.empty_like(current_mask)
            self._r = torch.empty_like(current_mask)
        self._q.exponential_(generator=self.generator)
        torch.div(current_mask, self._q, out=self._r)
        return torch.argmax(self._r, dim=1)


class MultinomialPolicy(object):
    """The same distribution through torch.multinomial on ``current_mask`` (round 3's stand-in policy)."""

    def __init__(self, generator=None):
        self.generator = generator

    def __call__(self, step, current_mask, **_):
        return torch.multinomial(current_mask, 1, generator=self.generator).squeeze(1)


class UniformPickPolicy(object):
    """Uniformly random selectable column from PRE-DRAWN uniforms ``u`` (B, steps) in [0, 1): the k-th selectable
    column of ``current_mask`` with k = floor(u * count), by cumulative sum -- plain deterministic torch ops, so
    an episode with this policy can be captured in a HIP graph and replayed on refreshed ``u`` (torch's own
    in-graph RNG, ``torch.multinomial`` with a graph-registered generator, aborted under back-to-back replays on
    this ROCm stack)."""

    def __init__(self, u):
        self.u = u

    def __call__(self, step, current_mask, **_):
        cnt = current_mask.sum(1)
        k = torch.minimum((self.u[:, step] * cnt).floor(), cnt - 1).clamp_(min=0)
        return (current_mask.cumsum(1) > k.unsqueeze(1)).to(torch.float32).argmax(1)


class UniformKeysPolicy(object):
    """Uniformly random selectable column from PRE-DRAWN iid keys ``v`` (steps, B, nR) in [0, 1): the selectable
    column with the largest key -- the argmax of iid keys over a set is uniform over the set.  Two torch ops per step
    (a multiply and an argmax), deterministic given ``v``, so an episode with it can be captured in a HIP graph and
    replayed on refreshed keys.  (A key of exactly 0 on the only selectable column would tie with the masked ones;
    torch.rand draws from [0, 1), so add a tiny offset when that matters.)"""

    def __init__(self, v):
        self.v = v

    def __call__(self, step, current_mask, **_):
        return torch.argmax(current_mask * self.v[step], dim=1)


def _flagged_reward(reward, check, *envs):
    """What an episode reports for containers whose sticky error word is set (a placement above the container's
    height -- the reference's IndexError --, a block the container cannot take, a footprint beyond the big-container
    kernels' support masks, tapenv.h: tap_env_errors): their placements were skipped, so their score is not the
    episode's.  check='nan' (default): NaN reward for exactly those containers, no host sync (two small launches,
    capturable in a hipGraph); 'raise': env.check() -- one host sync, TapOverflowError / TapError like the reference's
    raise; False / None: the raw calc_ratio (the caller checks, e.g. once per epoch with stepper.check())."""
    if not check:
        return reward
    if check == 'raise':
        for e in envs:
            e.check()
        return reward
    if check != 'nan':
        raise ValueError("check must be 'nan', 'raise' or False, not %r" % (check,))
    err = envs[0].errors
    for e in envs[1:]:
        err = err | e.errors
    return torch.where(err != 0, torch.full_like(reward, float('nan')), reward)


def run_episode(static, dynamic, policy, container_width, container_height,
                reward_type='C+P+S-lb-soft', heightmap_type='diff', packing_strategy='LB_GREEDY',
                input_type='bot', allow_rot=True, env=None, record=False, steps=None, fused=True, bits=None,
                container_length=None, stepper=None, check='nan'):
    """One episode for a batch (model.py:254-515 minus the network).

    ``policy(step=, static=, dynamic=, current_mask=, mask=, decoder_static=, decoder_dynamic=)``
    returns ptr (B,) int64.  With ``fused`` every step is ONE launch (tap_transition*; the two launches behind the
    same entry for the shapes without a single kernel), the first one starting from a fresh container and the last
    one emitting calc_ratio; otherwise a step is two launches (tap_mask_step, tap_env_step_gather).  Returns a dict:
    tour_idx (B, steps), reward = -scores (B,) fp32 (model.py:515), env, and with ``record`` the per-step features /
    masks.  ``bits``: carry ``dynamic`` as its bit shadow between the steps, None = when possible.

    The fused loop runs on a ``pack.EpisodeStepper`` (windows without a bit shadow: on its fp32-copy form): between two calls of the
    policy the host makes ONE C call and issues no torch op (the step's launch also writes ``decoder_static`` and
    the tour column).  ``stepper``: a stepper to re-use across episodes (a trainer builds one per run: nothing is
    allocated per episode; the returned tensors are then views of its buffers, valid until its next ``begin``);
    None builds one for this call.  ``check``: what ``reward`` holds for containers that raised an error bit
    (_flagged_reward: 'nan' -- NaN, no host sync --, 'raise', or False for the raw ratio).
    """
    if input_type in ('mul', 'mul-with'):
        return _run_episode_mul(static, dynamic, policy, container_width, container_height, reward_type,
                                heightmap_type, packing_strategy, input_type, allow_rot, record, steps, check)
    block_dim = int(static.shape[1]) - 1
    n = int(dynamic.shape[-1]) // (math.factorial(block_dim) if allow_rot else 1)
    B, D = int(static.shape[0]), block_dim
    nsteps = n if steps is None else steps
    dev = _lib.resolve_device(static.device)
    # (windows without a bit shadow -- rows > 128, nR % 4 != 0, nR > 256 -- run on the stepper too since round 5: its
    #  fp32-copy form; ``bits=True`` still asks for the shadow and fails there)
    if stepper is not None or (fused and bits is not False and not isinstance(bits, torch.Tensor) and
                               (bits is None or bits_supported(int(dynamic.shape[1]), int(dynamic.shape[2])))):
        try:
            return _run_episode_stepper(static, dynamic, policy, container_width, container_height, reward_type,
                                        heightmap_type, packing_strategy, input_type, allow_rot, env, record, nsteps,
                                        container_length, stepper, dev, n, D, check)
        except NonBinaryDynamic:
            if bits is True or stepper is not None:
                raise ValueError("dynamic cannot be carried as a bit shadow (it holds values other than 0 and 1)")
            if env is not None:
                env.reset()
    # model.py:279 builds square 3D containers (L = W); container_length lifts that for callers that want it
    cs = [container_width, container_height] if D == 2 else \
        [container_width, container_length or container_width, container_height]
    if env is None:
        env = BatchedContainer(B, cs, n, reward_type, heightmap_type, packing_strategy=packing_strategy, device=dev)
    if fused:
        masks = EnvTransition(static.to(dev), dynamic.to(dev), env, input_type, allow_rot, bits)
    else:
        masks = MaskStepper(static.to(dev), dynamic.to(dev), input_type, allow_rot, bits)
        env.reset()
    static_part = masks.static[:, 1:, :]
    decoder_static = torch.zeros(B, D, 1, device=dev)
    decoder_dynamic = torch.zeros(env._feature_shape(), device=dev)
    tour, feats, curs, msks = [], [], [], []
    ratio = None
    for step in range(nsteps):
        ptr = policy(step=step, static=masks.static, dynamic=masks.dynamic,
                     current_mask=masks.current_mask, mask=masks.mask,
                     decoder_static=decoder_static, decoder_dynamic=decoder_dynamic)
        ptr = ptr.to(torch.int64)
        decoder_static = torch.gather(static_part, 2, ptr.view(-1, 1, 1).expand(-1, D, 1))  # model.py:404-406
        if fused:                                                 # model.py:376-386 + 451-465, one launch
            _, _, _, decoder_dynamic, r = masks.step(ptr, fresh=(step == 0), want_ratio=(step == nsteps - 1))
            ratio = r if r is not None else ratio
        else:
            masks.step(ptr)                                       # model.py:376-386
            decoder_dynamic = env.add_new_blocks_gather(masks.static, ptr)                  # model.py:451-465
        tour.append(ptr.unsqueeze(1))
        if record:
            feats.append(decoder_dynamic); curs.append(masks.current_mask); msks.append(masks.mask)
    if ratio is None:
        ratio = env.calc_ratios()                                 # model.py:499-510
    out = {'tour_idx': torch.cat(tour, dim=1), 'reward': _flagged_reward(-ratio, check, env), 'env': env,
           'dynamic': masks.dynamic, 'mask': masks.mask}
    if record:
        out.update(features=feats, current_masks=curs, masks=msks)
    return out


def _as_instances(t, dev):
    if t.dtype is not torch.float32 or not t.is_contiguous() or t.device != dev:
        t = t.to(device=dev, dtype=torch.float32).contiguous()
    return t


def _run_episode_stepper(static, dynamic, policy, container_width, container_height, reward_type, heightmap_type,
                         packing_strategy, input_type, allow_rot, env, record, nsteps, container_length, stepper, dev, n, D,
                         check='nan'):
    """run_episode's loop on a pack.EpisodeStepper: per decoding step the policy's call and one C call."""
    static, dynamic = _as_instances(static, dev), _as_instances(dynamic, dev)
    if stepper is None:
        if env is None:
            cs = [container_width, container_height] if D == 2 else \
                [container_width, container_length or container_width, container_height]   # model.py:279: L = W
            env = BatchedContainer(int(static.shape[0]), cs, n, reward_type, heightmap_type,
                                   packing_strategy=packing_strategy, device=dev)
        stepper = EpisodeStepper(static, dynamic, env, input_type, allow_rot, steps=nsteps)
    elif stepper.steps != nsteps:
        raise ValueError("the stepper was built for %d steps per episode, not %d" % (stepper.steps, nsteps))
    sp = stepper
    sp.begin(static, dynamic)
    feats, curs, msks = [], [], []
    for step in range(nsteps):
        ptr = policy(step=step, static=static, dynamic=sp.dynamic, current_mask=sp.current_mask, mask=sp.mask,
                     decoder_static=sp.decoder_static, decoder_dynamic=sp.decoder_dynamic)
        sp.step(ptr)                                              # model.py:376-465, one launch
        if record:
            feats.append(sp.decoder_dynamic.clone()); curs.append(sp.current_mask.clone()); msks.append(sp.mask.clone())
    out = {'tour_idx': sp.tour, 'reward': _flagged_reward(-sp.ratio, check, sp.env), 'env': sp.env, 'dynamic': sp.dynamic,
           'mask': sp.mask, 'stepper': sp}
    if record:
        out.update(features=feats, current_masks=curs, masks=msks)
    return out


def _run_episode_mul(static, dynamic, policy, container_width, container_height, reward_type,
                     heightmap_type, packing_strategy, input_type, allow_rot, record, steps, check='nan'):
    """The two-container variant (input types 'mul' / 'mul-with', model.py:290-292, 396-447,
    503-507): the last row of ``static`` holds each block's target container id; per step the chosen
    block is placed in container a (id 0) or b (id 1) and the other one only reports its height-map
    (the C ABI's ``active`` mask); decoder features are concatenated on dim 1 and the score is the
    mean of the two ``calc_ratio`` values."""
    D = int(static.shape[1]) - 2
    n = int(dynamic.shape[-1]) // (math.factorial(D) if allow_rot else 1)
    B = int(static.shape[0])
    cs = [container_width, container_height] if D == 2 else [container_width, container_width, container_height]
    dev = _lib.resolve_device(static.device)
    env_a = BatchedContainer(B, cs, n, reward_type, heightmap_type, packing_strategy=packing_strategy, device=dev)
    env_b = BatchedContainer(B, cs, n, reward_type, heightmap_type, packing_strategy=packing_strategy, device=dev)
    masks = MaskStepper(static.to(dev), dynamic.to(dev), input_type, allow_rot)
    static_part = masks.static[:, 1:-1, :] if input_type == 'mul' else masks.static[:, 1:, :]   # model.py:389-394
    ssize = static_part.shape[1]
    decoder_static = torch.zeros(B, ssize, 1, device=dev)
    decoder_dynamic = torch.cat((torch.zeros(env_a._feature_shape(), device=dev),
                                 torch.zeros(env_b._feature_shape(), device=dev)), 1)
    tour, feats, curs, msks = [], [], [], []
    for step in range(n if steps is None else steps):
        ptr = policy(step=step, static=masks.static, dynamic=masks.dynamic,
                     current_mask=masks.current_mask, mask=masks.mask,
                     decoder_static=decoder_static, decoder_dynamic=decoder_dynamic).to(torch.int64)
        masks.step(ptr)
        target = torch.gather(masks.static[:, -1, :], 1, ptr.view(-1, 1)).squeeze(1)           # model.py:396-401
        decoder_static = torch.gather(static_part, 2, ptr.view(-1, 1, 1).expand(-1, ssize, 1))
        fa = env_a.add_new_blocks_gather(masks.static, ptr, active=(target == 0))               # model.py:419-427
        fb = env_b.add_new_blocks_gather(masks.static, ptr, active=(target == 1))
        decoder_dynamic = torch.cat((fa, fb), 1)                                               # model.py:429-447
        tour.append(ptr.unsqueeze(1))
        if record:
            feats.append(decoder_dynamic); curs.append(masks.current_mask); msks.append(masks.mask)
    ratio = (env_a.calc_ratios() + env_b.calc_ratios()) / 2.0                                  # model.py:503-507
    out = {'tour_idx': torch.cat(tour, dim=1), 'reward': _flagged_reward(-ratio, check, env_a, env_b), 'env': env_a, 'env_b': env_b,
           'dynamic': masks.dynamic, 'mask': masks.mask}
    if record:
        out.update(features=feats, current_masks=curs, masks=msks)
    return out
