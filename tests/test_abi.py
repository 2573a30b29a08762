"""The C-ABI library loads and exports every symbol include/*.h declares (no GPU needed)."""
import ctypes
import errno
import glob
import os
import re

import pytest

from blurrily_amd import Map, _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names.update(re.findall(r"\b(blurrily_[a-z_]+)\s*\(", text))
    return sorted(names)


def test_header_declares_the_reference_abi():
    want = {"blurrily_storage_" + n for n in
            ("new", "load", "close", "mark", "save", "put", "delete", "find", "stats")}   # storage.h:36-117
    assert want <= set(declared_functions())


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared_functions():
        assert getattr(lib, name) is not None, name
    assert set(declared_functions()) == set(_native.EXPORTED_SYMBOLS)


def test_tokeniser_entry_point():
    lib = _native.lib()
    out = (ctypes.c_uint16 * 7)()
    assert lib.blurrily_tokeniser_parse_string(b"london", out) == 7
    assert list(out) == [407, 3543, 9408, 11400, 11408, 11886, 12096]


def test_find_fails_loudly_without_a_gpu(has_gpu, capfd):
    if has_gpu:
        pytest.skip("a GPU is present: the HIP path runs instead")
    m = Map()
    m.put("london", 123)
    with pytest.raises(OSError) as e:
        m.find("london")
    assert e.value.errno == errno.ENODEV
    assert "no usable HIP device" in capfd.readouterr().err
    with pytest.raises(OSError):
        m.find_batch(["london", "paris"])


def test_product_does_not_reference_the_oracle():
    """Nothing under blurrily_amd/ may import, link or open anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "blurrily_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)
