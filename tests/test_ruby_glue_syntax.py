"""The Ruby-side glue has never met a Ruby toolchain (the image has none): at least the C compiler's FRONT END
sees it.  `gcc -fsyntax-only -std=c99 -Wall -Wextra -Werror` over

  * ruby/ext/blurrily/map_ext_reference.c -- the gem's OWN glue (ext/blurrily/map_ext.c:1-230), #included where
    it lies in the reference tree with its initialiser renamed, and
  * ruby/ext/blurrily/map_ext_batch.c -- the batched methods, against the gem's storage.h AND
    include/blurrily_storage.h in one translation unit,

with the declarations-only headers of tests/c/mock_ruby/ in place of ruby.h.  No object code is produced and
nothing is linked or run.  A second check holds the two translation units to ONE definition of Init_map_ext
(what the rename is for): the preprocessed gem glue must define Init_map_ext_reference and not Init_map_ext."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEM_EXT = "/root/reference/ext/blurrily"
MOCK = os.path.join(ROOT, "tests", "c", "mock_ruby")
GLUE = os.path.join(ROOT, "ruby", "ext", "blurrily")
# the gem's own flags (ext/blurrily/extconf.rb:4-16), -Werror kept: the glue must be warning-free
FLAGS = ["-std=c99", "-Wall", "-Wextra", "-Werror", "-DPLATFORM_LINUX", "-D_XOPEN_SOURCE=700", "-D_GNU_SOURCE=1",
         "-D_FILE_OFFSET_BITS=64", "-I", MOCK, "-I", GEM_EXT, "-I", os.path.join(ROOT, "include")]

needs_reference = pytest.mark.skipif(not os.path.exists(os.path.join(GEM_EXT, "map_ext.c")),
                                     reason="the reference tree is not on this box")


def _gcc(*args):
    return subprocess.run(["gcc", *args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


@needs_reference
@pytest.mark.parametrize("src", ["map_ext_reference.c", "map_ext_batch.c"])
def test_glue_passes_the_front_end(src):
    res = _gcc("-fsyntax-only", *FLAGS, os.path.join(GLUE, src))
    assert res.returncode == 0, res.stdout


@needs_reference
def test_one_definition_of_the_initialiser():
    ref = _gcc("-E", "-P", *FLAGS, os.path.join(GLUE, "map_ext_reference.c"))
    bat = _gcc("-E", "-P", *FLAGS, os.path.join(GLUE, "map_ext_batch.c"))
    assert ref.returncode == 0 and bat.returncode == 0, ref.stdout + bat.stdout
    defines = lambda text, name: re.search(r"\bvoid\s+%s\s*\(\s*void\s*\)\s*\{" % name, text) is not None
    assert defines(ref.stdout, "Init_map_ext_reference") and not defines(ref.stdout, "Init_map_ext")
    assert defines(bat.stdout, "Init_map_ext") and not defines(bat.stdout, "Init_map_ext_reference")
    # ... and the batch glue calls the gem's initialiser first
    assert re.search(r"Init_map_ext_reference\s*\(\s*\)\s*;", bat.stdout)


def test_the_mock_headers_define_nothing():
    """declarations only: no function body, no object definition (a stand-in Ruby would be something else)"""
    for rel in ("ruby.h", os.path.join("ruby", "thread.h")):
        text = open(os.path.join(MOCK, rel)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        body = re.sub(r"struct\s+\w+\s*\{[^}]*\}\s*;", "", text)          # a struct's fields are not a body
        body = re.sub(r"enum\s+\w+\s*\{[^}]*\}\s*;", "", body)
        assert "{" not in body, rel


@needs_reference
def test_every_guarded_method_exists_in_the_gem():
    """the wrappers re-define exactly methods the gem's Init_map_ext defines (map_ext.c:219-228)"""
    gem = open(os.path.join(GEM_EXT, "map_ext.c")).read()
    batch = open(os.path.join(GLUE, "map_ext_batch.c")).read()
    guarded = re.findall(r'guard_method\("(\w+)"', batch)
    assert sorted(guarded) == ["close", "delete", "find", "put", "save", "stats"]
    for name in guarded:
        assert re.search(r'rb_define_method\(klass,\s*"%s"' % name, gem), name
