#!/usr/bin/env python3
"""Throughput of the line-protocol front-end (blurrily_amd/server.py) on the GPU box: C connections
each pipelining Q FINDs against a Geonames-like haystack, once coalescing FINDs into GPU batches and
once with one find per line (the reference's server.rb:40-46 behaviour).

    python tools/server_bench.py [haystack_strings] [connections] [finds_per_connection]
"""
import asyncio
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np  # noqa: E402
import workloads as W  # noqa: E402
from blurrily_amd.server import Server  # noqa: E402


def serve(server, ready):
    loop = asyncio.new_event_loop()
    asyncio.set_event_loop(loop)
    loop.add_signal_handler = lambda *a, **k: None

    async def main():
        ev = asyncio.Event()
        task = asyncio.ensure_future(server.serve(ev))
        await ev.wait()
        ready.append(loop)
        await task

    loop.run_until_complete(main())


async def client(port, lines):
    reader, writer = await asyncio.open_connection("127.0.0.1", port)
    writer.write(("\n".join(lines) + "\n").encode())
    await writer.drain()
    for _ in lines:
        await reader.readline()
    writer.close()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    conns = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    hay, off = W.geonames(n, max(1000, n // 16), 3)
    strings = [s.decode() for s in W.unpack(hay, off)]
    rng = np.random.default_rng(1)
    work = [[f"FIND\tplaces\t{strings[int(i)]}" for i in rng.integers(0, n, size=per)] for _ in range(conns)]
    for coalesce in (True, False):
        with tempfile.TemporaryDirectory() as d:
            server = Server("127.0.0.1", 0, d, coalesce=coalesce, save_interval=3600)
            ready = []
            t = threading.Thread(target=serve, args=(server, ready), daemon=True)
            t.start()
            while not ready:
                time.sleep(0.01)
            m = server._map_group.map("places")
            m.put_many(strings, list(range(1, n + 1)))
            m.find("warm up")
            total = conns * per if coalesce else conns * max(1, per // 20)
            lines = work if coalesce else [w[:max(1, per // 20)] for w in work]

            async def run():
                await asyncio.gather(*[client(server.port, w) for w in lines])

            t0 = time.perf_counter()
            asyncio.run(run())
            dt = time.perf_counter() - t0
            print(f"coalesce={coalesce}: {total} FINDs over {conns} connections in {dt:.3f} s = {total / dt:,.0f} finds/s; "
                  f"{server.stats['batches']} GPU batches, largest {server.stats['largest_batch']}")
            ready[0].call_soon_threadsafe(server.stop)
            t.join(30)


if __name__ == "__main__":
    main()
