"""Blurrily::Map#normalize_string (lib/blurrily/map.rb:40-47) restated in Python."""
import pytest

from blurrily_amd import normalize_string
from helpers import load_golden


@pytest.mark.parametrize("vec", load_golden("spec_vectors.json")["normalize"], ids=lambda v: v["raw"])
def test_spec_vectors(vec):
    """every string of the reference's specs that reaches normalize_string (map_spec.rb, command_processor_spec.rb,
    integration_spec.rb), with the normalised form the spec's own assertion implies: where the spec states a trigram
    count or a weight (= strlen, storage.c:409), the form must have exactly that many"""
    from helpers import Oracle
    got = normalize_string(vec["raw"])
    assert got == vec["normalized"]
    if "trigrams" in vec:
        assert len(Oracle.tokenise(got.encode())) == vec["trigrams"]
    if "weight" in vec:
        assert len(got.encode()) == vec["weight"]
    if all(ord(c) < 0x80 for c in vec["raw"]):                  # the independent checker agrees (ASCII input)
        assert Oracle.normalize_ascii(vec["raw"].encode()) == vec["normalized"].encode()


def test_at_least_a_dozen_reference_held_normaliser_vectors():
    assert len(load_golden("spec_vectors.json")["normalize"]) >= 12


def test_frozen_non_ascii_vectors():
    """Non-ASCII needles: expected strings worked out from a literal table of the Unicode Character Database's
    (immutable) decomposition mappings, not from unicodedata (tools/make_normalize_vectors.py) -- the host-side
    behaviour is frozen across Python / Unicode versions even though ActiveSupport's own tables stay unpinned."""
    vectors = load_golden("normalize_vectors.json")["vectors"]
    assert len(vectors) >= 200
    for v in vectors:
        assert normalize_string(v["raw"]) == v["normalized"], v


@pytest.mark.parametrize("raw,want", [
    ("London", "london"),
    ("  many   spaces\there ", "many spaces here"),
    ("Saint-Étienne", "saint tienne"),      # ASCII-only downcase: 'É' -> NFKD 'E'+accent -> 'E' is not [a-z] -> ' '
    ("saint-étienne", "saint etienne"),
    ("ﬁsh", "fish"),                         # NFKD compatibility decomposition
    ("123 main st.", "main st"),
    ("", ""),
    ("abc\n%%%", "abc %%%"),                # Ruby's ^...$ match per line: a clean first line skips the clean-up
])
def test_restated_behaviour(raw, want):
    assert normalize_string(raw) == want


# ---- the independent checker: oracle/normalize_oracle.c (written from the Ruby text) -----------
import numpy as np  # noqa: E402

from helpers import Oracle  # noqa: E402

NORMALISER_ALPHABET = (list(b"abcdefghijklmnopqrstuvwxyz") * 6 + list(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ") * 2
                       + list(b"      ") + list(b"0123456789-_'.,;:!?@#()[]/\\\"") + [9, 10, 10, 11, 12, 13, 0, 127, 1, 31])

EDGE_NEEDLES = [b"London", b"  New   York ", b"Port-au-Prince", b"", b"   ", b"\t\n", b"A", b"@#%", b"abc\ndef",
                b"abc\n@@ x", b"ab\ncd\0ef", b"\0abc", b"abc\0\0", b"Abc \0", b"x\ry", b"UPPER lower  MiXeD",
                b"tab\tsep", b"trailing  ", b"  leading", b"a--b", b"ab\n", b"\nab", b"a\x7fb", b"plain line\nJUNK!!\n",
                b"\n", b"\n\n", b" \n ", b"a\n\nb", b"x\x0by", b"x\x0cy", b"\x0b\x0cab\x0b", b"ab\0 ", b"ab \0\0 \0",
                b"%%\nabc", b"%%\n\nabc\n", b"abc\r\n%%", b"ABC\nDEF", b"a b\n%"]


def random_ascii_needles(seed, n):
    rng = np.random.default_rng(seed)
    return [bytes(rng.choice(NORMALISER_ALPHABET, size=int(rng.integers(0, 40))).tolist()) for _ in range(n)]


def seen_by_c(b):
    """What the C side reads of a needle: the bytes up to the first NUL (StringValuePtr + strlen)."""
    return b.split(b"\0", 1)[0]


def test_oracle_normaliser_on_the_spec_vectors():
    # the two ASCII vectors the reference's specs hold; '@€%é' is non-ASCII: the oracle declines
    assert Oracle.normalize_ascii(b"New York") == b"new york"                 # map_spec.rb:195-202
    assert Oracle.normalize_ascii(b"Port-au-Prince") == b"port au prince"     # map_spec.rb:364
    assert Oracle.normalize_ascii("@€%é".encode()) is None                    # map_spec.rb:55-59: NFKD, unpinned
    # hand-derived from the Ruby text (map.rb:40-47)
    assert Oracle.normalize_ascii(b"abc\n%%%") == b"abc %%%"                  # a plain LINE skips the clean-up
    assert Oracle.normalize_ascii(b"%%%\n") == b""
    assert Oracle.normalize_ascii(b"ab \0\0") == b"ab"                        # strip drops trailing NULs
    assert Oracle.normalize_ascii(b"a\0b") == b"a b"                          # NUL is not [a-z ]: whole needle cleaned
    assert Oracle.normalize_ascii(b"ab\ncd\0ef") == b"ab cd\0ef"              # plain first line: NUL kept


def test_host_mirror_matches_the_oracle_normaliser():
    """blurrily_amd.normalize_string against the C restatement of the Ruby, 20 000 random ASCII
    needles (letters, capitals, digits, punctuation, every whitespace kind, NUL, DEL) + edge cases."""
    for nd in EDGE_NEEDLES + random_ascii_needles(77, 20000):
        want = Oracle.normalize_ascii(nd)
        got = normalize_string(nd.decode("latin1")).encode("latin1")
        assert seen_by_c(got) == seen_by_c(want), (nd, got, want)
        assert got == want, (nd, got, want)
