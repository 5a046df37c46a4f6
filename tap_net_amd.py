"""Import alias: the package directory is ``tap-net_amd`` (not a Python identifier), so
``import tap_net_amd`` loads it under this name and registers its submodules too."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("tap-net_amd")
for _name, _mod in list(sys.modules.items()):
    if _name == "tap-net_amd" or _name.startswith("tap-net_amd."):
        sys.modules["tap_net_amd" + _name[len("tap-net_amd"):]] = _mod
sys.modules[__name__] = _pkg
