"""Loader for the upstream reference (only present in the build container, never on the GPU box).

Used by tests/golden/make_golden.py (fixture generation) and by the `reference`-marked tests that
compare the oracle live against the reference.  Nothing here is imported by the product.
"""
import math
import os
import sys

REFERENCE_DIR = os.environ.get("TAP_REFERENCE_DIR", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REFERENCE_DIR, "tools.py"))


_mods = None


def load():
    """-> (tools, pack, generate) reference modules, or None when the checkout is absent."""
    global _mods
    if _mods is not None:
        return _mods
    if not available():
        return None
    import numpy
    numpy.math = math              # np.math was removed in numpy 2 (pack.py:109, generate.py:993 ...)
    sys.dont_write_bytecode = True  # the checkout is read-only
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REFERENCE_DIR)
    try:
        import tools as ref_tools
        import pack as ref_pack
        import generate as ref_generate
    finally:
        sys.path.remove(REFERENCE_DIR)
    _mods = (ref_tools, ref_pack, ref_generate)
    return _mods
