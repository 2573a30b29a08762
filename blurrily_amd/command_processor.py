"""Blurrily::CommandProcessor (lib/blurrily/command_processor.rb:5-52): one tab-separated
command line in, one reply line out.

    FIND   <db> <needle> [limit]        -> OK <ref> <matches> <weight> ...   (flattened rows)
    PUT    <db> <needle> <ref> [weight] -> OK
    DELETE <db> <ref>                   -> OK
    CLEAR  <db>                         -> OK
    anything wrong                      -> ERROR <message>

`parse` / `reply_*` are split out so the batching server (server.py) can validate a FIND, park it,
and answer it from one GPU batch; `process_command` is the reference's one-line-at-a-time call.
"""
import re

from .defaults import LIMIT_DEFAULT, LIMIT_RANGE, REF_RANGE, WEIGHT_RANGE


class ProtocolError(Exception):
    """CommandProcessor::ProtocolError (command_processor.rb:6)."""


COMMANDS = ("FIND", "PUT", "DELETE", "CLEAR")                       # command_processor.rb:24
_ARITY = {"FIND": (2, 3), "PUT": (3, 4), "DELETE": (2, 2), "CLEAR": (1, 1)}   # on_* arities, map_name included
_DB_NAME = re.compile(r"^[a-z_]+$", re.M)                           # Ruby's ^ $ anchor at lines
_DIGITS = re.compile(r"^\d+$", re.M)
_TO_I = re.compile(r"\s*([+-]?\d+(?:_\d+)*)")


def ruby_to_i(s):
    """String#to_i: the leading integer of the string (underscores between digits allowed), else 0."""
    m = _TO_I.match(s)
    return int(m.group(1).replace("_", "")) if m else 0


def split_fields(line):
    """String#split(/\\t/): trailing empty fields are dropped."""
    fields = line.split("\t")
    while fields and fields[-1] == "":
        fields.pop()
    return fields


class Find:
    """A validated FIND, ready to run."""
    __slots__ = ("map_name", "needle", "limit")

    def __init__(self, map_name, needle, limit):
        self.map_name, self.needle, self.limit = map_name, needle, limit


def reply_ok(result=None):
    """['OK', *result].compact.join("\\t") (command_processor.rb:17)."""
    return "\t".join(["OK"] + [str(x) for x in (result or [])])


def reply_error(message):
    return "ERROR\t" + message


def reply_rows(rows):
    """FIND's reply: the rows flattened (command_processor.rb:45)."""
    return reply_ok([x for row in rows for x in row])


class CommandProcessor:
    def __init__(self, map_group):
        self._map_group = map_group

    # ---- the reference's entry point -------------------------------------------------------
    def process_command(self, line):                                # command_processor.rb:12-20
        try:
            parsed = self.parse(line)
            if isinstance(parsed, Find):
                m = self._map_group.map(parsed.map_name)
                return reply_rows(m.find(parsed.needle, parsed.limit))
            return reply_ok(parsed)
        except ProtocolError as e:
            return reply_error(str(e))

    # ---- validation + the commands that do not touch the GPU --------------------------------
    def parse(self, line, fields=None):
        """Validate one line.  PUT / DELETE / CLEAR are executed here (returns None, i.e. a bare
        OK); a FIND comes back as a `Find` for the caller to run.  Raises ProtocolError."""
        if fields is None:
            fields = split_fields(line)
        command = fields[0] if fields else None
        map_name = fields[1] if len(fields) > 1 else None
        args = fields[2:]
        if command not in COMMANDS:
            raise ProtocolError("Unknown command")
        if map_name is None or not _DB_NAME.search(map_name):
            raise ProtocolError("Invalid database name")
        lo, hi = _ARITY[command]
        given = 1 + len(args)
        if not lo <= given <= hi:                                   # Ruby's ArgumentError from send()
            expected = str(lo) if lo == hi else f"{lo}..{hi}"
            raise ProtocolError(f"wrong number of arguments ({given} for {expected})")
        return getattr(self, "_on_" + command)(map_name, *args)

    def _on_PUT(self, map_name, needle, ref, weight=None):          # command_processor.rb:26-32
        if not (_DIGITS.search(ref) and ruby_to_i(ref) in REF_RANGE):
            raise ProtocolError("Invalid reference")
        if not (weight is None or (_DIGITS.search(weight) and ruby_to_i(weight) in WEIGHT_RANGE)):
            raise ProtocolError("Invalid weight")
        # [needle, ref.to_i, weight.to_i].compact: nil.to_i is 0, so a missing weight is 0
        self._map_group.map(map_name).put(needle, ruby_to_i(ref), ruby_to_i(weight) if weight is not None else 0)
        return None

    def _on_DELETE(self, map_name, ref):                            # command_processor.rb:34-39
        if not (_DIGITS.search(ref) and ruby_to_i(ref) in REF_RANGE):
            raise ProtocolError("Invalid reference")
        self._map_group.map(map_name).delete(ruby_to_i(ref))
        return None

    def _on_FIND(self, map_name, needle, limit=None):               # command_processor.rb:41-46
        if limit is not None and ruby_to_i(limit) not in LIMIT_RANGE:
            raise ProtocolError("Limit must be a number")
        return Find(map_name, needle, ruby_to_i(limit) if limit is not None else LIMIT_DEFAULT)

    def _on_CLEAR(self, map_name):                                  # command_processor.rb:48-51
        self._map_group.clear(map_name)
        return None
