"""include/blurrily_storage.h against the reference's own ext/blurrily/storage.h, in one translation unit
(tests/c/header_compat.c) under -Werror: incompatible declarations of the nine functions of storage.h:36-117
do not compile.  Needs /root/reference (this container); skipped elsewhere -- the header is never copied."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_EXT = "/root/reference/ext/blurrily"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_EXT, "storage.h")), reason="no reference tree on this box")
@pytest.mark.parametrize("order", ["reference_first", "ours_alone"])
def test_the_two_headers_agree_in_one_translation_unit(tmp_path, order):
    src = os.path.join(ROOT, "tests", "c", "header_compat.c")
    if order == "ours_alone":
        # our header alone must define the same types itself (what a C client without the gem sees)
        text = open(src).read().replace('#include "storage.h"', "/* (reference header left out) */")
        src = str(tmp_path / "ours_alone.c")
        open(src, "w").write(text)
    # flags of ext/blurrily/extconf.rb:4-13
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-DPLATFORM_LINUX", "-D_XOPEN_SOURCE=700", "-D_GNU_SOURCE=1",
           "-D_FILE_OFFSET_BITS=64", "-I", REF_EXT, "-I", os.path.join(ROOT, "include"), "-c", src,
           "-o", str(tmp_path / "compat.o")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_the_ruby_shim_is_committed_as_files():
    for rel in ("ruby/ext/blurrily/extconf.rb", "ruby/ext/blurrily/map_ext_batch.c", "ruby/lib/blurrily/map_batch.rb"):
        assert os.path.getsize(os.path.join(ROOT, rel)) > 500, rel
    # the batch glue binds part 2 of the header by the names the library exports
    from blurrily_amd import _native
    text = open(os.path.join(ROOT, "ruby/ext/blurrily/map_ext_batch.c")).read()
    lib = _native.lib()
    for sym in ("blurrily_storage_find_batch", "blurrily_storage_find_batch_raw", "blurrily_storage_put_many",
                "blurrily_storage_sync_device", "blurrily_storage_set_option", "blurrily_storage_get_option"):
        assert sym in text and hasattr(lib, sym), sym
