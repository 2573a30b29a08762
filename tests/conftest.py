import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _gpu_available():
    try:
        import ctypes
        from blurrily_amd import _native
        hip = _native.hip_runtime()                 # the one runtime of this process (see _native.py)
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except (OSError, ImportError):
        return False


@pytest.fixture(scope="session")
def has_gpu():
    return _gpu_available()


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The suites need the built artefacts; build() is idempotent and quick when up to date."""
    import __graft_entry__ as g
    g.build()


class _FullGeonames:
    """configs[2]'s haystack (8 423 769 strings: tools/workloads.py bench_haystack("geonames")) ONCE per session, and the
    oracle over it -- two full-size tests check against it, and it is the one costly build of the suite.  The oracle's
    ten seconds of put_many run on a thread of their own from the start (the C call releases the interpreter lock), under
    the product map's build and first finds; `oracle` waits for it."""

    def __init__(self):
        import threading
        import workloads as W
        from helpers import Oracle
        self.hay, self.off = W.bench_haystack("geonames")
        self._oracle = Oracle()
        self._building = threading.Thread(target=self._oracle.put_many, args=(self.hay, self.off), daemon=True)
        self._building.start()

    @property
    def oracle(self):
        self._building.join()
        return self._oracle


@pytest.fixture(scope="session")
def geonames_full():
    return _FullGeonames()


TWO_RANKS_DETAIL = "/tmp/blurrily_two_ranks_detail_%d.json" % os.getpid()


def two_ranks_command(port, detail=TWO_RANKS_DETAIL):
    """bench.py's N > 1 path as two ranks sharing this box's GPU over gloo (tests/test_gpu_bench_batch.py:
    test_bench_two_ranks_plumbing)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
            "--gpus", "2", "--steps", "2", "--warmup", "1", "--scale", "0.05", "--detail", detail]


@pytest.hookimpl(trylast=True)
def pytest_collection_modifyitems(config, items):
    """That run is a minute of two fresh interpreters importing torch: started when the session starts, it overlaps the
    first tests instead of standing in the driver's clock; the test collects it (a plumbing test: nothing in it, or in
    the tests it overlaps, asserts on a time).  Last of the collection hooks, so that `items` is what -k / -m / --deselect
    left; never under --collect-only; pytest_sessionfinish kills it if the test that collects it never ran."""
    if config.getoption("collectonly", False):
        return
    if not any(it.name == "test_bench_two_ranks_plumbing" for it in items) or not _gpu_available():
        return
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BLURRILY_DIST_BACKEND="gloo")
    try:
        import __graft_entry__ as g
        g.build()                                            # (the ranks load the built library)
        config._two_ranks = subprocess.Popen(two_ranks_command(port), env=env, cwd=ROOT, stdout=subprocess.PIPE,
                                             stderr=subprocess.PIPE, text=True, start_new_session=True)   # (its own process group: killed whole)
    except Exception:
        config._two_ranks = None


def pytest_sessionfinish(session, exitstatus):
    """The two-rank bench run started with the session (above) does not outlive it: an aborted or -x-stopped session
    would otherwise leave a minute of GPU work behind."""
    proc = getattr(session.config, "_two_ranks", None)
    if proc is not None and proc.poll() is None:
        import signal
        try:
            os.killpg(os.getpgid(proc.pid), signal.SIGTERM)
        except Exception:
            proc.terminate()
        try:
            proc.wait(timeout=20)
        except Exception:
            proc.kill()
    try:
        os.unlink(TWO_RANKS_DETAIL)
    except OSError:
        pass
