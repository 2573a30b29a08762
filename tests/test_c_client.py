"""The boundary is a C ABI: a plain C99 program (tests/c/abi_client.c) includes include/blurrily_storage.h,
links against libblurrily_hip.so and drives it the way ext/blurrily/map_ext.c drives the reference."""
import os
import subprocess

import pytest

from blurrily_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def client(tmp_path_factory):
    exe = tmp_path_factory.mktemp("c") / "abi_client"
    libdir = os.path.dirname(_native.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "abi_client.c"), "-o", str(exe), "-L", libdir,
                    "-l:libblurrily_hip.so", "-Wl,-rpath," + libdir], check=True)
    return str(exe)


def test_c_client_host_side(client, tmp_path, has_gpu):
    if has_gpu:
        pytest.skip("a GPU is present: the gpu variant runs instead")
    r = subprocess.run([client, str(tmp_path / "c.trigrams"), "host"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr


@pytest.mark.gpu
def test_c_client_on_the_gpu(client, tmp_path):
    r = subprocess.run([client, str(tmp_path / "c.trigrams"), "gpu"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr
