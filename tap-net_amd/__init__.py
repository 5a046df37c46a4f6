"""tap-net_amd -- MI355X-native batched Transport-and-Pack environment.

A drop-in for the packing hot path of Juzhan/TAP-Net (tools.Container, pack.update_dynamic /
update_mask / reward, PACKDataset) backed by hand-written HIP kernels behind the C ABI in
include/tapenv.h.  The directory name is not a Python identifier; import it as ``tap_net_amd``
(the alias module at the repository root) or via importlib.
"""
from . import _lib, build, datafiles, dist, env, generate, pack, rolling, rollout, synth, tools   # noqa: F401
from ._lib import TapError, TapOverflowError                   # noqa: F401
from .env import BatchedContainer, Container                   # noqa: F401
from .generate import generate_instances                       # noqa: F401
from .pack import (EnvTransition, EpisodeStepper, MaskStepper, PACKDataset, episode_scores, initial_mask, render, reward,   # noqa: F401
                   update_dynamic, update_mask)
from .rolling import RollingDataset, RollingStepper, RollingWindows, run_rolling_episode            # noqa: F401
from .rollout import RandomFeasiblePolicy, TapePolicy, UniformKeysPolicy, UniformPickPolicy, run_episode   # noqa: F401

__all__ = ["BatchedContainer", "Container", "MaskStepper", "EnvTransition", "EpisodeStepper", "PACKDataset", "initial_mask", "reward", "render", "episode_scores", "RollingDataset",
           "update_dynamic", "update_mask", "run_episode", "TapePolicy", "RandomFeasiblePolicy", "UniformPickPolicy", "UniformKeysPolicy",
           "generate_instances", "RollingWindows", "RollingStepper", "run_rolling_episode", "TapError", "TapOverflowError"]
