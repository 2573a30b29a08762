"""Front-end mirror -- the parts that need no GPU.  Reads like
spec/blurrily/command_processor_spec.rb, map_group_spec.rb, client_spec.rb and server_spec.rb."""
import os
import signal
import socket
import subprocess
import sys
import threading
import time

import pytest

from blurrily_amd import Client, CommandProcessor, Map, MapGroup
from blurrily_amd.command_processor import ruby_to_i, split_fields

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- CommandProcessor (command_processor_spec.rb:10-62; the FIND vectors run in test_gpu_frontend.py) ----
@pytest.fixture
def processor(tmp_path):
    return CommandProcessor(MapGroup(tmp_path))


def test_put_is_ok(processor):
    assert processor.process_command("PUT\tlocations_en\tgreat london\t12") == "OK"
    assert processor.process_command("PUT\tdb\tWhatever string\t12\t1") == "OK"


@pytest.mark.parametrize("line,reply", [
    ("Some stuff", "ERROR\tUnknown command"),
    ("", "ERROR\tUnknown command"),
    ("FIND\tbad db name\tWhatever string", "ERROR\tInvalid database name"),
    ("FIND", "ERROR\tInvalid database name"),
    ("FIND\tdb\tWhatever string\tlimit", "ERROR\tLimit must be a number"),
    ("FIND\tdb\tWhatever string\t0", "ERROR\tLimit must be a number"),
    ("FIND\tdb\tWhatever string\t1025", "ERROR\tLimit must be a number"),
    ("PUT\tdb\tWhatever string\t12\tweight", "ERROR\tInvalid weight"),
    ("PUT\tdb\tWhatever string\tref", "ERROR\tInvalid reference"),
    ("PUT\tdb\tWhatever string\t0", "ERROR\tInvalid reference"),
    ("PUT\tdb\tWhatever string\t2147483649", "ERROR\tInvalid reference"),
    ("DELETE\tdb\tx1", "ERROR\tInvalid reference"),
])
def test_errors(processor, line, reply):
    assert processor.process_command(line) == reply


def test_too_many_arguments(processor):
    out = processor.process_command("PUT\tdb\tWhatever string\tref\tweight\targument too much")
    assert out.startswith("ERROR\twrong number ")
    assert processor.process_command("CLEAR\tdb\textra").startswith("ERROR\twrong number ")
    assert processor.process_command("PUT\tdb").startswith("ERROR\twrong number ")


def test_put_delete_clear_reach_the_map(tmp_path):
    group = MapGroup(tmp_path)
    cp = CommandProcessor(group)
    assert cp.process_command("PUT\tdb\tLondon\t7") == "OK"
    assert cp.process_command("PUT\tdb\tParis\t8\t3\t") == "OK"             # trailing tab: split drops it
    assert group.map("db").stats() == {"references": 2, "trigrams": 13}
    assert cp.process_command("DELETE\tdb\t7") == "OK"
    assert group.map("db").stats()["references"] == 1
    assert cp.process_command("CLEAR\tdb") == "OK"
    assert group.map("db").stats() == {"references": 0, "trigrams": 0}


def test_ruby_string_helpers():
    assert split_fields("a\tb\t\t") == ["a", "b"]
    assert split_fields("a\t\tb") == ["a", "", "b"]
    assert split_fields("") == []
    assert [ruby_to_i(s) for s in ("12", " 12", "12abc", "abc", "", "1_000", "-3", "+4", "1__0")] == \
        [12, 12, 12, 0, 0, 1000, -3, 4, 1]


# ---- MapGroup (map_group_spec.rb:9-54) -------------------------------------------------------------
def test_map_group_returns_one_map_per_name(tmp_path):
    group = MapGroup(tmp_path)
    en, fr = group.map("location_en"), group.map("location_fr")
    assert isinstance(en, Map)
    assert group.map("location_en") is en and group.map("location_en") is not fr


def test_map_group_loads_from_file_if_it_exists(tmp_path):
    group = MapGroup(tmp_path)
    group.map("location_en").put("aaa", 123, 0)
    group.save()
    loaded = MapGroup(tmp_path).map("location_en")
    assert loaded.stats() == {"references": 1, "trigrams": 4}
    assert loaded.put("bbb", 123, 0) == 0                                     # the reference is known


def test_map_group_saves_all_maps_in_the_chosen_directory(tmp_path):
    group = MapGroup(tmp_path / "nested" / "data")
    group.map("location_en")
    group.map("location_fr")
    group.save()
    assert (tmp_path / "nested" / "data" / "location_en.trigrams").exists()
    assert (tmp_path / "nested" / "data" / "location_fr.trigrams").exists()


# ---- Client (client_spec.rb:15-68): argument checks and the wire format ------------------------------
class _OneShotServer:
    """mock_tcp_next_request: accepts one connection, records the request line, sends `reply`."""

    def __init__(self, reply):
        self.sock = socket.socket()
        self.sock.bind(("127.0.0.1", 0))
        self.sock.listen(1)
        self.port = self.sock.getsockname()[1]
        self.request = None
        self._thread = threading.Thread(target=self._serve, args=(reply,), daemon=True)
        self._thread.start()

    def _serve(self, reply):
        conn, _ = self.sock.accept()
        f = conn.makefile("rwb")
        self.request = f.readline().decode()
        if reply is not None:
            f.write(reply.encode() + b"\n")
            f.flush()
        conn.close()
        self.sock.close()

    def join(self):
        self._thread.join(5)


def _client(port=1):
    return Client(host="127.0.0.1", port=port, db_name="location_en")


def test_client_find_argument_checks():
    c = _client()
    with pytest.raises(TypeError):
        c.find()
    with pytest.raises(ValueError):
        c.find("needle\twith\ttabs")
    with pytest.raises(ValueError):
        c.find("london", "blah")
    with pytest.raises(ValueError):
        c.find("")


def test_client_find_returns_records():
    srv = _OneShotServer("OK\t1337\t1\t2")
    assert _client(srv.port).find("london") == [[1337, 1, 2]]
    srv.join()
    assert srv.request == "FIND\tlocation_en\tlondon\t10\n"


def test_client_handles_no_records_and_errors():
    srv = _OneShotServer("OK")
    assert _client(srv.port).find("blah") == []
    srv = _OneShotServer("ERROR")
    with pytest.raises(Client.Error):
        _client(srv.port).find("blah")
    srv = _OneShotServer("ERROR\tUnknown command")
    with pytest.raises(Client.Error, match="Unknown command"):
        _client(srv.port).find("blah")
    srv = _OneShotServer(None)
    with pytest.raises(Client.Error, match="Server disconnected"):
        _client(srv.port).find("blah")


def test_client_put_argument_checks_and_request():
    c = _client()
    with pytest.raises(TypeError):
        c.put()
    with pytest.raises(ValueError):
        c.put("South\tLondon", 123, 0)
    with pytest.raises(TypeError):
        c.put("London")
    with pytest.raises(ValueError):
        c.put("London", "abc", 0)
    with pytest.raises(ValueError):
        c.put("London", 123, "a")
    srv = _OneShotServer("OK")
    assert _client(srv.port).put("London", 123, 0) is None
    srv.join()
    assert srv.request == "PUT\tlocation_en\tLondon\t123\t0\n"


# ---- Server (server_spec.rb:27-50), in its own process -------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.fixture
def running_server(tmp_path):
    port = _free_port()
    proc = subprocess.Popen([sys.executable, "-m", "blurrily_amd.server", "--host", "127.0.0.1", "--port", str(port),
                             "--directory", str(tmp_path / "data")], cwd=ROOT)
    deadline = time.time() + 60
    while True:
        try:
            socket.create_connection(("127.0.0.1", port), timeout=1).close()
            break
        except OSError:
            assert proc.poll() is None and time.time() < deadline, "server did not start"
            time.sleep(0.1)
    yield port, proc, tmp_path / "data"
    if proc.poll() is None:
        proc.kill()
        proc.wait()


def test_server_responds_and_keeps_the_connection(running_server):
    port, _, _ = running_server
    f = socket.create_connection(("127.0.0.1", port)).makefile("rwb")
    f.write(b"Who is most beautiful in the world?\n")
    f.flush()
    assert f.readline().startswith(b"ERROR\tUnknown command")
    f.write(b"Bad command\n" * 3)
    f.flush()
    for _ in range(3):
        assert f.readline().startswith(b"ERROR")
    f.write(b"PUT\twords\tmerveilleux\t1\nPUT\twords\tmerveille\t2\t0\nDELETE\twords\t2\n")
    f.flush()
    assert [f.readline() for _ in range(3)] == [b"OK\n"] * 3


def test_server_saves_when_quitting(running_server):
    port, proc, directory = running_server
    f = socket.create_connection(("127.0.0.1", port)).makefile("rwb")
    f.write(b"PUT\twords\tmerveilleux\t1\n")
    f.flush()
    assert f.readline() == b"OK\n"
    f.close()
    proc.send_signal(signal.SIGTERM)
    assert proc.wait(30) == 0
    assert (directory / "words.trigrams").exists()
    assert Map.load(directory / "words.trigrams").stats()["references"] == 1


def test_server_saves_on_usr1_through_the_worker_thread(running_server):
    """server.rb:26 (USR1 saves).  Saves run on the worker thread that owns the maps, queued behind
    the commands already accepted, so the file holds every acknowledged PUT."""
    port, proc, directory = running_server
    f = socket.create_connection(("127.0.0.1", port)).makefile("rwb")
    f.write(b"".join(b"PUT\twords\tword%d\t%d\n" % (i, i) for i in range(1, 201)))
    f.flush()
    assert [f.readline() for _ in range(200)] == [b"OK\n"] * 200
    proc.send_signal(signal.SIGUSR1)
    deadline = time.time() + 30
    while not (directory / "words.trigrams").exists():
        assert time.time() < deadline
        time.sleep(0.05)
    time.sleep(0.2)                                             # (the rename is atomic; let the save finish)
    assert Map.load(directory / "words.trigrams").stats()["references"] == 200
    f.write(b"PUT\twords\tlater\t999\n")                        # still serving
    f.flush()
    assert f.readline() == b"OK\n"


def test_server_refuses_an_endless_line(running_server):
    port, _, _ = running_server
    s = socket.create_connection(("127.0.0.1", port))
    s.sendall(b"FIND\twords\t" + b"a" * (80 * 1024))              # no newline
    f = s.makefile("rb")
    assert f.readline() == b"ERROR\tline too long\n"
    assert f.readline() == b""                                  # and the connection is closed


def test_dispatcher_outlives_an_exception_that_escapes_a_batch(tmp_path):
    """Whatever escapes Server._run (it answers per line, so nothing should) becomes an ERROR reply for the lines of
    that batch; the dispatcher goes on and the next command is served -- a dead dispatcher would leave every client
    waiting on an open connection (the reference's reactor would have died: server.rb:40-46 has no rescue)."""
    import asyncio
    from blurrily_amd.server import Server

    async def scenario():
        srv = Server("127.0.0.1", 0, str(tmp_path))
        ready = asyncio.Event()
        task = asyncio.ensure_future(srv.serve(ready))
        await ready.wait()
        real_run, calls = srv._run, []

        def flaky(lines):
            calls.append(list(lines))
            if len(calls) == 1:
                raise RuntimeError("boom")
            return real_run(lines)
        srv._run = flaky
        reader, writer = await asyncio.open_connection("127.0.0.1", srv.port)
        writer.write(b"PUT\twords\tmerveilleux\t1\n")
        await writer.drain()
        first = await asyncio.wait_for(reader.readline(), 10)
        writer.write(b"PUT\twords\tmerveille\t2\n")
        await writer.drain()
        second = await asyncio.wait_for(reader.readline(), 10)
        writer.close()
        srv.stop()
        await asyncio.wait_for(task, 30)
        return first, second

    first, second = asyncio.run(scenario())
    assert first == b"ERROR\tboom\n" and second == b"OK\n"
