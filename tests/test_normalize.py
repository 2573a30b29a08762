"""Blurrily::Map#normalize_string (lib/blurrily/map.rb:40-47) restated in Python."""
import pytest

from blurrily_amd import normalize_string
from helpers import load_golden


@pytest.mark.parametrize("vec", load_golden("spec_vectors.json")["normalize"], ids=lambda v: v["raw"])
def test_spec_vectors(vec):
    assert normalize_string(vec["raw"]) == vec["normalized"]


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
