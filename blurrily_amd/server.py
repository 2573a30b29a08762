"""Blurrily::Server (lib/blurrily/server.rb:7-48) as a batching front-end.

Same wire protocol -- one tab-separated command per line, one reply line per command, replies in
request order per connection -- and same life cycle (save every 60 s, on USR1 and on shutdown; INT
and TERM stop it).  What differs is how FINDs reach the index: the reference runs one
`Map#find` per line inside the reactor callback (server.rb:40-46); here the FINDs that arrive
while the previous batch is on the GPU are coalesced into ONE `find_batch` per map, so throughput
follows the GPU's batch rate instead of a per-call latency.  Ordering is kept: a PUT / DELETE /
CLEAR on a map first flushes the FINDs queued before it on that map.
"""
import argparse
import asyncio
import os
import signal
from collections import deque
from concurrent.futures import ThreadPoolExecutor

from .command_processor import (COMMANDS, CommandProcessor, Find, ProtocolError, reply_error, reply_ok, reply_rows,
                                split_fields)
from .defaults import DEFAULT_PORT
from .map_group import MapGroup


MAX_LINE_BYTES = 64 * 1024      # longest command line accepted (a needle is tens of bytes)
MAX_PENDING = 4096              # reads queued for the dispatcher before connections are paused
RUBY_STRIP = " \t\r\n\f\v\0"   # what String#strip removes


class Server:
    def __init__(self, host="0.0.0.0", port=DEFAULT_PORT, directory=None, max_batch=8192, coalesce=True,
                 save_interval=60.0):
        self._host, self._port = host, port                          # server.rb:10-12
        self._map_group = MapGroup(directory or os.getcwd())
        self._processor = CommandProcessor(self._map_group)
        self._max_batch = max_batch if coalesce else 1
        self._save_interval = save_interval
        self._pending = deque()                                      # (lines, future) in arrival order
        self._wake = None
        self._gpu = ThreadPoolExecutor(max_workers=1)                # one batch at a time, off the reactor
        self._server = None
        self._stopping = None
        self.stats = {"finds": 0, "batches": 0, "largest_batch": 0, "commands": 0}

    # ---- life cycle --------------------------------------------------------------------------
    def start(self):                                                 # server.rb:18-31
        asyncio.run(self.serve())

    async def serve(self, ready=None):
        loop = asyncio.get_running_loop()
        self._wake = asyncio.Event()
        self._stopping = asyncio.Event()
        for sig in (signal.SIGINT, signal.SIGTERM):
            loop.add_signal_handler(sig, self._stopping.set)
        usr1_saves = set()                                           # (referenced until done: not collected mid-save)

        def on_usr1():                                               # server.rb:26
            task = asyncio.ensure_future(self._logged_save("USR1"))
            usr1_saves.add(task)
            task.add_done_callback(usr1_saves.discard)
        loop.add_signal_handler(signal.SIGUSR1, on_usr1)
        self._server = await asyncio.start_server(self._handle, self._host, self._port)
        self.port = self._server.sockets[0].getsockname()[1]
        dispatcher = asyncio.ensure_future(self._dispatch())
        saver = asyncio.ensure_future(self._periodic_save())
        if ready is not None:
            ready.set()
        try:
            await self._stopping.wait()
        finally:
            self._server.close()
            await self._server.wait_closed()
            saver.cancel()
            dispatcher.cancel()
            if usr1_saves:                                           # a save asked for by signal finishes first
                await asyncio.gather(*usr1_saves, return_exceptions=True)
            # shutdown hook (server.rb:25): queued behind the batch in flight on the one worker
            # thread, so it sees every mutation that was acknowledged
            try:
                await self._save()
            finally:
                self._gpu.shutdown(wait=True)

    def stop(self):
        if self._stopping is not None:
            self._stopping.set()

    async def _save(self):
        """MapGroup#save on the worker thread that runs every PUT / DELETE / FIND: the reference's
        reactor is single-threaded (server.rb:19-30), so a save never overlaps a mutation or the
        bucket sort of a find; here the one-worker executor gives the same guarantee (ctypes calls
        release the GIL, so a save on the event-loop thread would race the worker)."""
        await asyncio.get_running_loop().run_in_executor(self._gpu, self._map_group.save)

    async def _logged_save(self, what):
        """A save whose failure (a full disk ...) is reported, not left to 'exception was never retrieved'."""
        try:
            await self._save()
        except asyncio.CancelledError:
            raise
        except Exception as e:
            print(f"blurrily: {what} save failed: {e}", flush=True)

    async def _periodic_save(self):                                  # server.rb:23-24
        while True:
            await asyncio.sleep(self._save_interval)
            await self._logged_save("periodic")                      # one failed save must not end the saver

    # ---- one connection ------------------------------------------------------------------------
    async def _handle(self, reader, writer):
        """Lines in, reply lines out, in order.  Whatever arrived in one read is queued as one item
        (a pipelining client costs one future per read, not one per line)."""
        replies = asyncio.Queue()

        async def write_in_order():
            while True:
                fut = await replies.get()
                if fut is None:
                    break
                out = await fut
                writer.write(("\n".join(out) + "\n").encode("utf-8", "replace"))
                if replies.empty():
                    await writer.drain()

        sender = asyncio.ensure_future(write_in_order())
        loop = asyncio.get_running_loop()
        tail = b""
        try:
            while True:
                data = await reader.read(1 << 16)
                if not data:
                    break
                *complete, tail = (tail + data).split(b"\n")
                if len(tail) > MAX_LINE_BYTES:                       # a client that never sends a newline
                    fut = loop.create_future()
                    fut.set_result([reply_error("line too long")])
                    replies.put_nowait(fut)
                    break
                # server.rb:41-42: data.split("\n") -- a blank line yields no command -- then strip
                # (String#strip: ASCII whitespace and NUL only, not Unicode spaces)
                lines = [ln.decode("utf-8", "replace").strip(RUBY_STRIP) for ln in complete if ln]
                if not lines:
                    continue
                while len(self._pending) > MAX_PENDING:              # backpressure: stop reading this socket
                    await asyncio.sleep(0.001)
                fut = loop.create_future()
                replies.put_nowait(fut)
                self._pending.append((lines, fut))
                self._wake.set()
        except (ConnectionError, asyncio.IncompleteReadError):
            pass
        finally:
            replies.put_nowait(None)
            try:
                await sender
                writer.close()
            except (ConnectionError, asyncio.CancelledError):
                pass

    # ---- the dispatcher: arrival order in, batches out -------------------------------------------
    async def _dispatch(self):
        loop = asyncio.get_running_loop()
        while True:
            await self._wake.wait()
            self._wake.clear()
            while self._pending:
                work, n = [], 0
                while self._pending and (not work or n + len(self._pending[0][0]) <= self._max_batch):
                    item = self._pending.popleft()
                    work.append(item)
                    n += len(item[0])
                # everything that touches a map runs on the one worker thread, in arrival order
                lines = [line for chunk, _ in work for line in chunk]
                try:
                    results = await loop.run_in_executor(self._gpu, self._run, lines)
                except asyncio.CancelledError:
                    raise
                except Exception as e:                               # (_run answers per line; whatever still escapes must
                    results = [reply_error(str(e))] * len(lines)     #  not end the dispatcher and leave every client waiting)
                at = 0
                for chunk, fut in work:
                    if not fut.done():
                        fut.set_result(results[at:at + len(chunk)])
                    at += len(chunk)

    def _run(self, lines):
        """Replies for `lines`, in order.  Consecutive FINDs on one map become one batch; a mutation
        of a map flushes that map's queued FINDs first."""
        replies = [None] * len(lines)
        queued = {}                                                  # map name -> [(index, Find)]

        def flush(name):
            batch = queued.pop(name, None)
            if not batch:
                return
            try:
                m = self._map_group.map(name)
                limit = max(f.limit for _, f in batch)               # a smaller limit is a prefix of a larger one
                rows = m.find_batch([f.needle for _, f in batch], limit)
                for (i, f), r in zip(batch, rows):
                    replies[i] = reply_rows(r[:f.limit])
            except Exception as e:                                   # keep serving (the reference would die here)
                for i, _ in batch:
                    replies[i] = reply_error(str(e))
            self.stats["finds"] += len(batch)
            self.stats["batches"] += 1
            self.stats["largest_batch"] = max(self.stats["largest_batch"], len(batch))

        for i, line in enumerate(lines):
            self.stats["commands"] += 1
            fields = split_fields(line)
            command = fields[0] if fields else None
            name = fields[1] if len(fields) > 1 else None
            try:
                if command in COMMANDS and command != "FIND" and name is not None:
                    flush(name)
                parsed = self._processor.parse(line, fields)
                if isinstance(parsed, Find):
                    queued.setdefault(parsed.map_name, []).append((i, parsed))
                    if self._max_batch == 1:
                        flush(parsed.map_name)
                else:
                    replies[i] = reply_ok(parsed)
            except ProtocolError as e:
                replies[i] = reply_error(str(e))
            except Exception as e:
                replies[i] = reply_error(str(e))
        for name in list(queued):
            flush(name)
        return replies


def main(argv=None):
    ap = argparse.ArgumentParser(description="blurrily front-end: tab-separated FIND/PUT/DELETE/CLEAR over TCP")
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=DEFAULT_PORT)
    ap.add_argument("--directory", default=os.getcwd())
    ap.add_argument("--no-coalesce", action="store_true", help="one find per FIND line, as the reference does")
    args = ap.parse_args(argv)
    Server(args.host, args.port, args.directory, coalesce=not args.no_coalesce).start()


if __name__ == "__main__":
    main()
