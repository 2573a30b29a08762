"""Blurrily::Client (lib/blurrily/client.rb:8-135): blocking TCP client of the line protocol."""
import socket

from .defaults import DEFAULT_DATABASE, DEFAULT_HOST, DEFAULT_PORT, LIMIT_DEFAULT, LIMIT_RANGE, REF_RANGE, WEIGHT_RANGE


class Error(RuntimeError):
    """Blurrily::Client::Error (client.rb:9)."""


class Client:
    Error = Error

    def __init__(self, host=DEFAULT_HOST, port=DEFAULT_PORT, db_name=DEFAULT_DATABASE):    # client.rb:28-32
        self._host, self._port, self._db_name = host, port, db_name
        self._sock = None
        self._file = None

    def find(self, needle, limit=None):                              # client.rb:52-60
        limit = LIMIT_DEFAULT if limit is None else limit
        self._check_valid_needle(needle)
        if not _in(limit, LIMIT_RANGE):
            raise ValueError(f"LIMIT value must be in {LIMIT_RANGE.start}..{LIMIT_RANGE.stop - 1}")
        flat = [int(x) for x in self._send_cmd_and_get_results(["FIND", self._db_name, needle, limit])]
        return [flat[i:i + 3] for i in range(0, len(flat), 3)]

    def put(self, needle, ref, weight=0):                            # client.rb:78-86
        self._check_valid_needle(needle)
        self._check_valid_ref(ref)
        if not _in(weight, WEIGHT_RANGE):
            raise ValueError(f"WEIGHT value must be in {WEIGHT_RANGE.start}..{WEIGHT_RANGE.stop - 1}")
        self._send_cmd_and_get_results(["PUT", self._db_name, needle, ref, weight])
        return None

    def delete(self, ref):                                           # client.rb:88-93
        self._check_valid_ref(ref)
        self._send_cmd_and_get_results(["DELETE", self._db_name, ref])
        return None

    def clear(self):                                                 # client.rb:95-98
        self._send_cmd_and_get_results(["CLEAR", self._db_name])
        return None

    def close(self):
        if self._sock is not None:
            self._file.close()
            self._sock.close()
            self._sock = self._file = None

    # ---- private (client.rb:101-133) ---------------------------------------------------------
    @staticmethod
    def _check_valid_needle(needle):
        if not isinstance(needle, str) or needle == "" or "\t" in needle:
            raise ValueError("bad needle")

    @staticmethod
    def _check_valid_ref(ref):
        if not _in(ref, REF_RANGE):
            raise ValueError(f"REF value must be in {REF_RANGE.start}..{REF_RANGE.stop - 1}")

    def _connection(self):
        if self._sock is None:
            self._sock = socket.create_connection((self._host, self._port))
            self._file = self._sock.makefile("rwb")
        return self._file

    def _send_cmd_and_get_results(self, argv):
        f = self._connection()
        f.write("\t".join(str(a) for a in argv).encode("utf-8") + b"\n")
        f.flush()
        reply = f.readline().decode("utf-8", "replace")
        if reply == "OK\n":
            return []
        if reply.startswith("OK\t") and reply.endswith("\n"):
            return reply[3:-1].split("\t")
        if reply.startswith("ERROR\t") and reply.endswith("\n"):
            raise Error(reply[6:-1])
        if reply == "":
            raise Error("Server disconnected")
        raise Error("Server did not respect protocol")


def _in(value, rng):
    """Range#include? of the Ruby: a non-numeric value is simply not in the range."""
    return isinstance(value, int) and not isinstance(value, bool) and value in rng
