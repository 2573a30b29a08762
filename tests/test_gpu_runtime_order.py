"""One HIP runtime per process, whichever of blurrily_amd and torch is imported first
(blurrily_amd/_native.py: _one_hip_runtime).  Each order runs in its own interpreter."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = r"""
import ctypes as C, numpy as np
m = RawMap()
for i, s in enumerate([b"london", b"londres", b"paris"], start=1):
    m.put(s, i, 0)
assert m.find(b"london", 10)[0] == [1, 7, 6]
# device-resident batch on torch's buffers and stream
lib = _native.lib()
packed = torch.tensor(list(b"londonparis"), dtype=torch.uint8, device="cuda")
off = torch.tensor([0, 6, 11], dtype=torch.int64, device="cuda")
rows = torch.zeros((2, 10, 3), dtype=torch.int32, device="cuda")
counts = torch.zeros(2, dtype=torch.int32, device="cuda")
nb = torch.zeros(2, dtype=torch.int32, device="cuda")
res = lib.blurrily_storage_find_batch_device(m.handle, packed.data_ptr(), 11, off.data_ptr(), 2, 10, rows.data_ptr(),
                                             counts.data_ptr(), nb.data_ptr(), torch.cuda.current_stream().cuda_stream)
assert res == 0, C.get_errno()
torch.cuda.synchronize()
assert counts.tolist() == [2, 1], counts.tolist()
assert rows[0, 0].tolist() == [1, 7, 6] and rows[1, 0].tolist() == [3, 6, 5]
maps = set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l)
assert len(maps) == 1, maps
print("ok")
"""


HEADS = {"blurrily": "from blurrily_amd import RawMap, _native\n_native.lib()\nimport torch\n",
         "torch": "import torch\nfrom blurrily_amd import RawMap, _native\n"}


@pytest.fixture(scope="module")
def interpreters():
    """both orders, each in its own interpreter -- STARTED TOGETHER: a fresh interpreter's `import torch` is most of a
    minute on a fresh box, and the two do not depend on each other"""
    procs = {k: subprocess.Popen([sys.executable, "-c", h + BODY], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for k, h in HEADS.items()}
    yield procs
    for p in procs.values():
        if p.poll() is None:
            p.kill()


@pytest.mark.parametrize("first", ["blurrily", "torch"])
def test_import_order(first, interpreters):
    p = interpreters[first]
    out, err = p.communicate(timeout=400)
    assert p.returncode == 0 and out.strip().endswith("ok"), out + err
