"""Front-end on the GPU: the FIND vectors of spec/blurrily/command_processor_spec.rb:15-23,53-55 and
the batching server (coalesced FINDs give, line for line, what one find per line gives)."""
import asyncio
import threading

import numpy as np
import pytest

import workloads as W
from blurrily_amd import Client, CommandProcessor, Map, MapGroup
from blurrily_amd.server import Server

pytestmark = pytest.mark.gpu


def test_put_and_find_finds_something(tmp_path):
    cp = CommandProcessor(MapGroup(tmp_path))
    assert cp.process_command("PUT\tlocations_en\tgreat london\t12") == "OK"
    assert cp.process_command("PUT\tlocations_en\tgreater masovian\t13") == "OK"
    assert cp.process_command("FIND\tlocations_en\tgreat") == "OK\t12\t6\t12\t13\t5\t16"
    assert cp.process_command("FIND\tlocations_en\tgreat\t1") == "OK\t12\t6\t12"


def test_find_returns_ok_if_nothing_found(tmp_path):
    cp = CommandProcessor(MapGroup(tmp_path))
    assert cp.process_command("FIND\tlocations_en\tgreat london") == "OK"
    assert cp.process_command("FIND\tdb\tWhatever string\t2") == "OK"


class _InProcessServer:
    def __init__(self, directory, **kw):
        self.server = Server("127.0.0.1", 0, directory, save_interval=3600, **kw)
        self._ready = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        assert self._ready.wait(60)
        self.port = self.server.port

    def _run(self):
        self.loop = asyncio.new_event_loop()
        asyncio.set_event_loop(self.loop)

        async def main():
            ready = asyncio.Event()
            task = asyncio.ensure_future(self.server.serve(ready))
            await ready.wait()
            self._ready.set()
            await task

        # signal handlers need the main thread: serve() without them
        self.loop.add_signal_handler = lambda *a, **k: None
        self.loop.run_until_complete(main())

    def stop(self):
        self.loop.call_soon_threadsafe(self.server.stop)
        self._thread.join(30)


async def _pipelined(port, lines):
    reader, writer = await asyncio.open_connection("127.0.0.1", port)
    writer.write(("\n".join(lines) + "\n").encode())
    await writer.drain()
    out = [(await reader.readline()).decode().rstrip("\n") for _ in lines]
    writer.close()
    return out


def test_coalesced_finds_equal_one_find_per_line(tmp_path):
    hay, off = W.geonames(40000, 6000, seed=23)
    strings = [s.decode() for s in W.unpack(hay, off)]
    reference = Map()                                     # answers one find at a time, like the reference's server
    reference.put_many(strings, list(range(1, len(strings) + 1)))
    srv = _InProcessServer(tmp_path)
    try:
        srv.server._map_group.map("places").put_many(strings, list(range(1, len(strings) + 1)))
        rng = np.random.default_rng(3)
        per_conn = []
        for c in range(16):
            lines = []
            for _ in range(200):
                s = strings[int(rng.integers(0, len(strings)))]
                limit = int(rng.choice([1, 3, 10, 25]))
                lines.append(f"FIND\tplaces\t{s.upper() if rng.random() < 0.2 else s}\t{limit}")
            lines.insert(50, "FIND\tplaces\tx\tnope")
            lines.insert(120, "bogus")
            per_conn.append(lines)

        async def run_all():
            return await asyncio.gather(*[_pipelined(srv.port, lines) for lines in per_conn])

        replies = asyncio.run(run_all())
        cp = CommandProcessor(type("G", (), {"map": lambda self, name: reference, "clear": None})())
        for lines, got in zip(per_conn, replies):
            assert got == [cp.process_command(line) for line in lines]
        stats = srv.server.stats
        assert stats["finds"] == 16 * 200
        assert stats["batches"] < stats["finds"] / 4 and stats["largest_batch"] > 16, stats
    finally:
        srv.stop()


def test_sixty_four_clients_every_reply_checked(tmp_path):
    """Sixty-four connections, each asking one FIND at a time and waiting for its reply (the reference's client,
    lib/blurrily/client.rb:35-41): what the server coalesces is then a batch of a few dozen needles -- the size that takes
    find_one_kernel's shared launch (raw needles, normalised by the library on the host) or latency mode's ranges -- and
    every reply equals what one find per line gives (lib/blurrily/command_processor.rb:41-46)."""
    hay, off = W.geonames(120000, 9000, seed=29)                # two windows
    strings = [s.decode() for s in W.unpack(hay, off)]
    reference = Map()
    reference.put_many(strings, list(range(1, len(strings) + 1)))
    srv = _InProcessServer(tmp_path)
    try:
        srv.server._map_group.map("places").put_many(strings, list(range(1, len(strings) + 1)))
        rng = np.random.default_rng(31)
        per_conn = []
        for c in range(64):
            lines = []
            for _ in range(30):
                s = strings[int(rng.integers(0, len(strings)))]
                if rng.random() < 0.3:
                    s = s[: max(1, len(s) - 2)]
                limit = int(rng.choice([1, 5, 10, 40]))
                lines.append(f"FIND\tplaces\t{s.title() if rng.random() < 0.3 else s}\t{limit}")
            per_conn.append(lines)

        async def one_at_a_time(lines):
            reader, writer = await asyncio.open_connection("127.0.0.1", srv.port)
            out = []
            for line in lines:
                writer.write((line + "\n").encode())
                await writer.drain()
                out.append((await reader.readline()).decode().rstrip("\n"))
            writer.close()
            return out

        async def run_all():
            return await asyncio.gather(*[one_at_a_time(lines) for lines in per_conn])

        replies = asyncio.run(run_all())
        cp = CommandProcessor(type("G", (), {"map": lambda self, name: reference, "clear": None})())
        for lines, got in zip(per_conn, replies):
            assert got == [cp.process_command(line) for line in lines]
        stats = srv.server.stats
        assert stats["finds"] == 64 * 30
        assert 1 < stats["largest_batch"] <= 64, stats           # coalesced, and never more than a needle per client
    finally:
        srv.stop()


def test_mutations_keep_their_place_between_finds(tmp_path):
    srv = _InProcessServer(tmp_path)
    try:
        lines = ["FIND\tdb\tlondon", "PUT\tdb\tlondon\t1", "FIND\tdb\tlondon", "PUT\tdb\tlondres\t2", "FIND\tdb\tlondon",
                 "DELETE\tdb\t1", "FIND\tdb\tlondon", "CLEAR\tdb", "FIND\tdb\tlondon", "FIND\tother\tlondon"]
        got = asyncio.run(_pipelined(srv.port, lines))
        assert got == ["OK", "OK", "OK\t1\t7\t6", "OK", "OK\t1\t7\t6\t2\t4\t7", "OK", "OK\t2\t4\t7", "OK", "OK", "OK"]
        c = Client(host="127.0.0.1", port=srv.port, db_name="db")
        c.put("Paris", 9)
        assert c.find("paris") == [[9, 6, 5]]
        c.delete(9)
        assert c.find("paris") == []
        c.clear()
        c.close()
    finally:
        srv.stop()
