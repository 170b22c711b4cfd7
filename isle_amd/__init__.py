"""isle_amd — MI355X (gfx950) implementation of ISLE's training hot path.

The product is the C-ABI shared library ``isle_amd/libisle_hip.so`` (sources in ``isle_amd/csrc``,
interface in ``include/isle_hip.h``).  This package is only the thin Python binding used by the
tests and the benchmark; it fails loudly when the HIP library is missing — there is no CPU fallback.
"""
from ._lib import load_library, library_path, IsleHipError  # noqa: F401
from .hot_path import HotPath, TIMING_FAMILIES  # noqa: F401
