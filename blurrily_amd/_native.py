"""ctypes binding of libblurrily_hip.so (include/blurrily_storage.h).

Fails loudly when the library is missing: there is no pure-Python or CPU path.
"""
import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# BLURRILY_LIB selects another build of the same library (same-box A/B of two builds, tools/ab_probe.py)
LIB_PATH = os.environ.get("BLURRILY_LIB") or os.path.join(_HERE, "libblurrily_hip.so")


class TrigramMatch(C.Structure):
    """storage.h:18-24 -- packed 12-byte row."""
    _pack_ = 1
    _fields_ = [("reference", C.c_uint32), ("matches", C.c_uint32), ("weight", C.c_uint32)]


class TrigramStat(C.Structure):
    """storage.h:26-30."""
    _fields_ = [("references", C.c_uint32), ("trigrams", C.c_uint32)]


class DeviceInfo(C.Structure):
    _fields_ = [
        ("device_ordinal", C.c_int32), ("n_refs", C.c_uint32), ("n_windows", C.c_uint32),
        ("window_bits", C.c_uint32), ("n_entries", C.c_uint64), ("device_bytes", C.c_uint64),
        ("last_find_kernel_ms", C.c_double), ("last_tokenise_kernel_ms", C.c_double),
        ("n_pending", C.c_uint32), ("n_tombstones", C.c_uint32), ("base_builds", C.c_uint64),
        ("mean_hit_slice", C.c_double), ("n_bitmaps", C.c_uint32), ("reserved_", C.c_uint32),
        ("dense_share", C.c_double), ("ws_gain", C.c_double),
        ("n_replicas", C.c_uint32), ("distinct_devices", C.c_uint32), ("peer_access_mask", C.c_uint32),
        ("same_device_mask", C.c_uint32), ("pci_bus_id", C.c_char * 16),
    ]


_lib = None


def _one_hip_runtime():
    """A process must run on ONE copy of the HIP runtime.  PyTorch-ROCm ships its own
    libamdhip64.so.7 and loads it by path; if this library came first, bound to /opt/rocm's copy,
    the two runtimes would not see each other's streams and device pointers (torch tensors handed to
    blurrily_storage_find_batch_device, as bench.py does).  The dynamic linker resolves our
    DT_NEEDED by soname against what is already loaded, so: when torch is installed but not yet
    imported, load its copy first -- whichever order the application imports things in, everybody
    then shares it.  torch itself is not imported."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def hip_runtime():
    """ctypes handle of the HIP runtime the library is bound to (tests allocate through it)."""
    lib()
    return C.CDLL("libamdhip64.so.7")


def lib():
    """Load the shared library once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  blurrily_amd has no CPU fallback.")
    _one_hip_runtime()
    L = C.CDLL(LIB_PATH, use_errno=True)
    vp, vpp = C.c_void_p, C.POINTER(C.c_void_p)
    sig = {
        "blurrily_storage_new": (C.c_int, [vpp]),
        "blurrily_storage_load": (C.c_int, [vpp, C.c_char_p]),
        "blurrily_storage_close": (C.c_int, [vpp]),
        "blurrily_storage_mark": (None, [vp]),
        "blurrily_storage_save": (C.c_int, [vp, C.c_char_p]),
        "blurrily_storage_put": (C.c_int, [vp, C.c_char_p, C.c_uint32, C.c_uint32]),
        "blurrily_storage_delete": (C.c_int, [vp, C.c_uint32]),
        "blurrily_storage_find": (C.c_int, [vp, C.c_char_p, C.c_uint16, C.c_void_p]),
        "blurrily_storage_stats": (C.c_int, [vp, C.POINTER(TrigramStat)]),
        "blurrily_storage_put_many": (C.c_long, [vp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
        "blurrily_storage_find_batch": (C.c_int, [vp, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint16,
                                                  C.c_void_p, C.c_void_p]),
        "blurrily_storage_find_batch_device": (C.c_int, [vp, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                         C.c_uint16, C.c_void_p, C.c_void_p, C.c_void_p,
                                                         C.c_void_p]),
        "blurrily_storage_find_batch_raw": (C.c_int, [vp, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint16,
                                                      C.c_void_p, C.c_void_p, C.c_void_p]),
        "blurrily_normalize_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                                      C.c_void_p]),
        "blurrily_storage_sync_device": (C.c_int, [vp]),
        "blurrily_tokeniser_parse_string": (C.c_int, [C.c_char_p, C.c_void_p]),
        "blurrily_storage_device_info": (C.c_int, [vp, C.POINTER(DeviceInfo)]),
        "blurrily_storage_set_timing": (None, [vp, C.c_int]),
        "blurrily_storage_set_stats": (None, [vp, C.c_int]),
        "blurrily_storage_find_stats": (C.c_int, [vp, C.c_void_p]),
        "blurrily_storage_find_path_flags": (C.c_int, [vp, C.c_void_p, C.c_size_t]),
        "blurrily_storage_set_option": (C.c_int, [vp, C.c_char_p, C.c_longlong]),
        "blurrily_storage_get_option": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_longlong)]),
        "blurrily_storage_device_info_sized": (C.c_size_t, [vp, C.c_void_p, C.c_size_t]),
        "blurrily_storage_tune": (C.c_int, [vp, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint16]),
        "blurrily_storage_last_kernels": (C.c_size_t, [vp, C.c_char_p, C.c_size_t]),
    }
    for name, (res, args) in sig.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            if os.environ.get("BLURRILY_LIB"):      # (an older build of the library under A/B: it lacks the newer entry points)
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTED_SYMBOLS = (
    "blurrily_storage_new", "blurrily_storage_load", "blurrily_storage_close", "blurrily_storage_mark",
    "blurrily_storage_save", "blurrily_storage_put", "blurrily_storage_delete", "blurrily_storage_find",
    "blurrily_storage_stats", "blurrily_storage_put_many", "blurrily_storage_find_batch",
    "blurrily_storage_find_batch_device", "blurrily_storage_find_batch_raw", "blurrily_normalize_batch_device",
    "blurrily_storage_sync_device", "blurrily_tokeniser_parse_string",
    "blurrily_storage_device_info", "blurrily_storage_set_timing",
    "blurrily_storage_set_stats", "blurrily_storage_find_stats",
    "blurrily_storage_set_option", "blurrily_storage_get_option", "blurrily_storage_find_path_flags",
    "blurrily_storage_device_info_sized", "blurrily_storage_tune", "blurrily_storage_last_kernels",
)
