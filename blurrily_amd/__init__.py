"""blurrily_amd -- MI355X-native trigram find behind the Blurrily::Map surface.

The product is ``libblurrily_hip.so`` (C ABI in ``include/blurrily_storage.h``,
hand-written HIP kernels for gfx950).  This package is the host-side mirror of
the reference's Ruby layer (``lib/blurrily/map.rb``) over that C ABI via
ctypes, because the image has no Ruby toolchain (see INTEGRATION.md for the
Ruby binding).  ``find`` has no CPU fallback: without a GPU it raises.
"""
from .client import Client
from .command_processor import CommandProcessor
from .defaults import LIMIT_DEFAULT, LIMIT_RANGE, REF_RANGE, WEIGHT_RANGE
from .map import ClosedError, Map, RawMap, normalize_string
from .map_group import MapGroup

__all__ = [
    "Map", "RawMap", "ClosedError", "normalize_string", "MapGroup", "CommandProcessor", "Client",
    "LIMIT_DEFAULT", "LIMIT_RANGE", "REF_RANGE", "WEIGHT_RANGE",
]
