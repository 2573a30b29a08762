"""Host-side mirror of the reference's Ruby API over the C ABI.

``RawMap`` mirrors the C glue class ``Blurrily::RawMap`` (ext/blurrily/map_ext.c:206-229:
``new``/``load``, ``put/3``, ``delete/1``, ``save/1``, ``find/2``, ``stats/0``, ``close/0``,
``ClosedError``); ``Map`` mirrors ``Blurrily::Map`` (lib/blurrily/map.rb:8-47: string
normalisation, default weight/limit, clean-path save elision).  Same names, same argument
meaning, same error behaviour -- so tests read like spec/blurrily/map_spec.rb.

The image has no Ruby; INTEGRATION.md shows the Ruby side of the same boundary.
"""
import ctypes as C
import os
import re
import unicodedata

import numpy as np

from . import _native
from .defaults import LIMIT_DEFAULT

_U32_MAX = 0xFFFFFFFF


class ClosedError(RuntimeError):
    """Blurrily::RawMap::ClosedError (map_ext.c:216)."""


def _raise_errno(path=None):
    err = C.get_errno()
    raise OSError(err, os.strerror(err), path)


def _u32(value, what):
    """NUM2UINT (map_ext.c:85-86): reject what does not fit an unsigned 32-bit integer."""
    v = int(value)
    if not 0 <= v <= _U32_MAX:
        raise OverflowError(f"{what} {value!r} out of range of unsigned int")
    return v


def _pack(needles):
    """list of bytes -> (packed bytes buffer, uint64 offsets[n+1])."""
    offsets = np.zeros(len(needles) + 1, dtype=np.uint64)
    if needles:
        offsets[1:] = np.cumsum([len(b) for b in needles], dtype=np.uint64)
    return b"".join(needles), offsets


class RawMap:
    """Thin object wrapper over a ``trigram_map`` handle (map_ext.c)."""

    ClosedError = ClosedError

    def __init__(self, _handle=None):
        self._lib = _native.lib()
        if _handle is None:
            h = C.c_void_p()
            if self._lib.blurrily_storage_new(C.byref(h)) < 0:      # map_ext.c:49-50
                _raise_errno()
            _handle = h
        self._h = _handle
        self._closed = False

    # -- construction ---------------------------------------------------------------
    @classmethod
    def load(cls, path):
        """map_ext.c:59-71 -- Errno::* on failure (ENOENT, EPROTO ...)."""
        lib = _native.lib()
        h = C.c_void_p()
        if lib.blurrily_storage_load(C.byref(h), os.fsencode(path)) < 0:
            _raise_errno(path)
        obj = cls.__new__(cls)
        RawMap.__init__(obj, _handle=h)
        return obj

    def __del__(self):                                              # blurrily_free, map_ext.c:25-32
        try:
            if not getattr(self, "_closed", True) and self._h:
                self._lib.blurrily_storage_close(C.byref(self._h))
                self._closed = True
        except Exception:
            pass

    def _check_open(self):
        if self._closed:
            raise ClosedError("Map was freed")                      # map_ext.c:11-16

    # -- reference surface ----------------------------------------------------------
    def put(self, needle, reference, weight):
        """map_ext.c:81-95 -> blurrily_storage_put.  Returns #trigrams added (0 for a dup ref)."""
        self._check_open()
        res = self._lib.blurrily_storage_put(self._h, _as_bytes(needle), _u32(reference, "reference"),
                                             _u32(weight, "weight"))
        assert res >= 0
        return res

    def delete(self, reference):
        """map_ext.c:99-111."""
        self._check_open()
        res = self._lib.blurrily_storage_delete(self._h, _u32(reference, "reference"))
        assert res >= 0
        return res

    def save(self, path):
        """map_ext.c:115-127."""
        self._check_open()
        if self._lib.blurrily_storage_save(self._h, os.fsencode(path)) < 0:
            _raise_errno(path)
        return None

    def find(self, needle, limit):
        """map_ext.c:131-162: limit <= 0 -> LIMIT_DEFAULT; rows ``[ref, matches, weight]``."""
        self._check_open()
        limit = int(limit)
        if not -(1 << 31) <= limit <= _U32_MAX:
            raise OverflowError("limit out of range")
        if limit > 0x7FFFFFFF:
            limit -= 1 << 32                      # NUM2UINT into an `int` (map_ext.c:135)
        if limit <= 0:
            limit = LIMIT_DEFAULT                 # map_ext.c:142-146
        c_limit = limit & 0xFFFF                  # uint16_t parameter (storage.h:110)
        rows = (_native.TrigramMatch * max(c_limit, 1))()
        res = self._lib.blurrily_storage_find(self._h, _as_bytes(needle), c_limit, rows)
        if res < 0:
            _raise_errno()
        return [[rows[k].reference, rows[k].matches, rows[k].weight] for k in range(res)]

    def stats(self):
        """map_ext.c:167-184."""
        self._check_open()
        st = _native.TrigramStat()
        res = self._lib.blurrily_storage_stats(self._h, C.byref(st))
        assert res >= 0
        return {"references": st.references, "trigrams": st.trigrams}

    def close(self):
        """map_ext.c:188-202."""
        self._check_open()
        if self._lib.blurrily_storage_close(C.byref(self._h)) < 0:
            _raise_errno()
        self._closed = True
        return None

    # -- batched extensions (no reference counterpart) -------------------------------
    def put_many(self, needles, references, weights=None):
        """Same as ``put`` for each element, in order; returns total trigrams added."""
        self._check_open()
        packed, offsets = _pack([_as_bytes(s) for s in needles])
        refs = np.ascontiguousarray(references, dtype=np.uint32)
        wts = None if weights is None else np.ascontiguousarray(weights, dtype=np.uint32)
        return self.put_many_packed(packed, offsets, refs, wts)

    def put_many_packed(self, packed, offsets, refs, weights=None):
        self._check_open()
        buf = np.frombuffer(packed, dtype=np.uint8) if not isinstance(packed, np.ndarray) else packed
        res = self._lib.blurrily_storage_put_many(
            self._h, buf.ctypes.data if buf.size else None, offsets.ctypes.data, refs.ctypes.data,
            None if weights is None else weights.ctypes.data, len(refs))
        if res < 0:
            _raise_errno()
        return res

    def find_batch_packed(self, packed, offsets, limit):
        """n finds in one GPU batch.  Returns (rows[n, limit, 3] uint32, counts[n] uint32)."""
        self._check_open()
        n = len(offsets) - 1
        limit = int(limit) & 0xFFFF
        rows = np.zeros((n, max(limit, 1), 3), dtype=np.uint32)
        counts = np.zeros(n, dtype=np.uint32)
        buf = np.frombuffer(packed, dtype=np.uint8) if not isinstance(packed, np.ndarray) else packed
        res = self._lib.blurrily_storage_find_batch(
            self._h, buf.ctypes.data if buf.size else None, offsets.ctypes.data, n, limit,
            rows.ctypes.data, counts.ctypes.data)
        if res < 0:
            _raise_errno()
        return rows[:, :limit, :], counts

    def find_batch_raw_packed(self, packed, offsets, limit):
        """find_batch_packed over un-normalised ASCII needles: normalize_string runs on the device
        (blurrily_storage_find_batch_raw).  Returns (rows, counts, non_ascii[n] uint32)."""
        self._check_open()
        n = len(offsets) - 1
        limit = int(limit) & 0xFFFF
        rows = np.zeros((n, max(limit, 1), 3), dtype=np.uint32)
        counts = np.zeros(n, dtype=np.uint32)
        flags = np.zeros(n, dtype=np.uint32)
        buf = np.frombuffer(packed, dtype=np.uint8) if not isinstance(packed, np.ndarray) else packed
        res = self._lib.blurrily_storage_find_batch_raw(
            self._h, buf.ctypes.data if buf.size else None, offsets.ctypes.data, n, limit,
            rows.ctypes.data, counts.ctypes.data, flags.ctypes.data)
        if res < 0:
            _raise_errno()
        return rows[:, :limit, :], counts, flags

    def sync_device(self):
        self._check_open()
        if self._lib.blurrily_storage_sync_device(self._h) < 0:
            _raise_errno()

    def device_info(self):
        self._check_open()
        info = _native.DeviceInfo()
        self._lib.blurrily_storage_device_info(self._h, C.byref(info))
        return {f: getattr(info, f) for f, _ in info._fields_}

    def set_timing(self, enabled):
        self._check_open()
        self._lib.blurrily_storage_set_timing(self._h, 1 if enabled else 0)

    STAT_NAMES = ("posting_entries", "steps", "table_words", "tasks", "compactions", "resweeps",
                  "units", "probes")

    def set_stats(self, enabled):
        """Request counters of the find kernels on/off (include/blurrily_storage.h)."""
        self._check_open()
        self._lib.blurrily_storage_set_stats(self._h, 1 if enabled else 0)

    # bits of find_path_flags() (csrc/find_kernels.h: kPath*)
    PATH_FLAGS = ("nibble", "byte", "cold_start", "resweep", "compaction", "skipped", "ring_overflow", "pipelined",
                  "wide", "chunked", "ranged", "multi_pass", "tombstone", "own_only", "ws_task", "ws_left_out",
                  "ws_robust", "ws_cand_overflow", "ws_pool_overflow", "ws_wide", "ws_table_walk", "nm_left_out", "small")

    def find_path_flags(self, n):
        """Per needle of the last find call made while set_stats(True): which kernel paths its find took
        (uint32 array; bit i = PATH_FLAGS[i])."""
        self._check_open()
        out = np.zeros(n, dtype=np.uint32)
        if self._lib.blurrily_storage_find_path_flags(self._h, out.ctypes.data, n) < 0:
            _raise_errno()
        return out

    def last_kernels(self):
        """Names of the find kernels the last batched find launched, in launch order (blurrily_storage_last_kernels)."""
        self._check_open()
        buf = C.create_string_buffer(512)
        self._lib.blurrily_storage_last_kernels(self._h, buf, len(buf))
        return [k for k in buf.value.decode().split("+") if k]

    def tune(self, packed, offsets, n, limit):
        """Measure now which sweep serves batches of n needles at this limit (blurrily_storage_tune): the needles
        given, repeated up to n, go through every sweep the class can take."""
        self._check_open()
        buf = np.frombuffer(packed, dtype=np.uint8) if not isinstance(packed, np.ndarray) else packed
        if self._lib.blurrily_storage_tune(self._h, buf.ctypes.data, offsets.ctypes.data, len(offsets) - 1, n, limit) < 0:
            _raise_errno()

    def set_option(self, key, value):
        """A tunable of this map (include/blurrily_storage.h: blurrily_storage_set_option)."""
        self._check_open()
        if self._lib.blurrily_storage_set_option(self._h, key.encode(), int(value)) < 0:
            _raise_errno()

    def get_option(self, key):
        self._check_open()
        out = C.c_longlong(0)
        if self._lib.blurrily_storage_get_option(self._h, key.encode(), C.byref(out)) < 0:
            _raise_errno()
        return int(out.value)

    def find_stats(self):
        """Counters of the last find call made while set_stats(True): a dict by STAT_NAMES."""
        self._check_open()
        out = (C.c_uint64 * 8)()
        if self._lib.blurrily_storage_find_stats(self._h, out) < 0:
            _raise_errno()
        return dict(zip(self.STAT_NAMES, (int(v) for v in out)))

    @property
    def handle(self):
        self._check_open()
        return self._h


def set_process_option(key, value):
    """A process-wide tunable ("host_threads", "build_trace"; include/blurrily_storage.h)."""
    if _native.lib().blurrily_storage_set_option(None, key.encode(), int(value)) < 0:
        _raise_errno()


def _as_bytes(s):
    """StringValuePtr: the C side sees the bytes up to the first NUL."""
    if isinstance(s, bytes):
        return s
    return str(s).encode("utf-8")


_PLAIN = re.compile(r"^([a-z ])+$", re.M)       # Ruby's ^ and $ are line anchors (map.rb:42)
_ASCII_UPPER = {c: c + 32 for c in range(ord("A"), ord("Z") + 1)}


def normalize_string(needle):
    """lib/blurrily/map.rb:40-47.

    downcase -> unless only ``[a-z ]``: NFKD, drop non-ASCII, non ``[a-z]`` -> space ->
    squeeze whitespace, strip.  ``downcase`` is ASCII-only, as on the Rubies the reference
    supports (.travis.yml:1-5: 1.9.3-2.2.0).  NFKD comes from Python's ``unicodedata``;
    the reference uses ActiveSupport 4.2's tables (Gemfile.lock:11) -- parity on non-ASCII
    input is pinned only by spec/blurrily/map_spec.rb:55-59 (``'@€%é'`` -> ``'e'``).
    """
    result = str(needle).translate(_ASCII_UPPER)
    if not _PLAIN.search(result):
        result = unicodedata.normalize("NFKD", result)
        result = re.sub(r"[^\x00-\x7F]", "", result)
        result = re.sub(r"[^a-z]", " ", result)
    result = re.sub(r"[ \t\r\n\f\v]+", " ", result)
    return result.strip(" \t\r\n\f\v").rstrip("\0 \t\r\n\f\v")


class Map(RawMap):
    """Blurrily::Map (lib/blurrily/map.rb:6-48)."""

    def __init__(self, _handle=None):
        super().__init__(_handle)
        self._clean_path = None

    def put(self, needle, reference, weight=None):                  # map.rb:8-13
        weight = 0 if weight is None else weight
        needle = normalize_string(needle)
        self._clean_path = None
        return super().put(needle, reference, weight)

    def find(self, needle, limit=LIMIT_DEFAULT):                    # map.rb:15-18
        return super().find(normalize_string(needle), limit)

    def delete(self, reference):                                    # map.rb:20-23
        self._clean_path = None
        return super().delete(reference)

    def save(self, path):                                           # map.rb:25-30
        path = os.fspath(path)
        if self._clean_path == path:
            return None
        super().save(path)
        self._clean_path = path
        return None

    @classmethod
    def load(cls, path):                                            # map.rb:32-36
        obj = super().load(path)
        obj._clean_path = os.fspath(path)
        return obj

    # batched counterparts: each element behaves exactly like the single call
    def put_many(self, needles, references, weights=None):
        self._clean_path = None
        return super().put_many([normalize_string(s) for s in needles], references, weights)

    def find_batch(self, needles, limit=LIMIT_DEFAULT):
        """``[self.find(s, limit) for s in needles]`` in one GPU batch."""
        limit = int(limit)
        if limit <= 0:
            limit = LIMIT_DEFAULT
        # ASCII needles go to the GPU as they are (normalize_string runs there); the others are
        # normalised here first -- NFKD is host work -- and pass through the device step unchanged
        raw = []
        for s in needles:
            b = _as_bytes(s)
            raw.append(b if b.isascii() else _as_bytes(normalize_string(b.decode("utf-8", "replace")
                                                                        if isinstance(s, bytes) else s)))
        packed, offsets = _pack(raw)
        rows, counts, flags = self.find_batch_raw_packed(packed, offsets, limit)
        assert not flags.any()
        return [rows[i, :counts[i]].tolist() for i in range(len(needles))]
