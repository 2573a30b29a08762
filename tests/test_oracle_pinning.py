"""Pin the CPU oracle (oracle/blurrily_oracle.c) before anything is checked against it.

(a) every known-answer vector the reference's specs hold for the find path
    (tests/golden/spec_vectors.json);
(b) fixtures emitted by the reference's own C (tests/golden/ref_*.json, tools/make_golden.py);
(c) where oracle/_ref is present, the live reference on seeded random haystacks.
CPU only.
"""
import os
import tempfile

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import workloads as W
from blurrily_amd import RawMap
from helpers import Oracle, Reference, golden_find_files, golden_haystack, load_golden

SPEC = load_golden("spec_vectors.json")


@pytest.mark.parametrize("vec", SPEC["find"], ids=lambda v: v["source"])
def test_oracle_reproduces_spec_find_vectors(vec):
    o = Oracle()
    for needle, ref, weight in vec["puts"]:
        o.put(needle.encode(), ref, weight)
    rows = o.find(vec["needle"].encode(), vec["limit"])
    if "expect_rows" in vec:
        assert rows == vec["expect_rows"]
    if "expect_refs" in vec:
        assert [r[0] for r in rows] == vec["expect_refs"]
    if "expect_first" in vec:
        assert rows[0] == vec["expect_first"]
    if "expect_first_ref" in vec:
        assert rows[0][0] == vec["expect_first_ref"]
    if "expect_len" in vec:
        assert len(rows) == vec["expect_len"]
    if vec.get("expect_nonempty"):
        assert rows


@pytest.mark.parametrize("vec", SPEC["put_counts"], ids=lambda v: v["source"])
def test_oracle_put_counts(vec):
    o = Oracle()
    assert o.put(vec["needle"].encode(), 123, 0) == vec["trigrams"]
    assert o.put(vec["needle"].encode(), 123, 0) == 0           # map_spec.rb:38-41
    assert o.stats() == {"references": 1, "trigrams": vec["trigrams"]}


def test_oracle_tokeniser_matches_reference_fixture():
    for v in load_golden("ref_tokeniser.json")["vectors"]:
        needle = bytes.fromhex(v["needle_hex"])
        assert Oracle.tokenise(needle) == v["codes"], needle


def test_known_tokeniser_codes():
    """SURVEY.md section 8(a) A2 [probe]."""
    assert Oracle.tokenise(b"london") == [407, 3543, 9408, 11400, 11408, 11886, 12096]
    assert Oracle.tokenise(b"") == [0]
    assert len(Oracle.tokenise(b"new york")) == 9


@pytest.mark.parametrize("name", golden_find_files())
def test_oracle_matches_reference_find_fixture(name):
    g = load_golden(name)
    hay, off, refs, weights = golden_haystack(g["haystack"])
    o = Oracle()
    strings = W.unpack(hay, off)
    for i, s in enumerate(strings):
        o.put(s, int(refs[i]), 0 if weights is None else int(weights[i]))
    assert o.stats() == g["stats"]
    for hexed, want in zip(g["needles_hex"], g["expected"]):
        assert o.find(bytes.fromhex(hexed), g["limit"]) == want


needs_ref = pytest.mark.skipif(not Reference.available(), reason="oracle/_ref not built (no reference tree)")


@needs_ref
def test_oracle_tokeniser_vs_live_reference():
    rng = np.random.default_rng(3)
    alphabet = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz   *Q\xc3", dtype=np.uint8)
    for _ in range(2000):
        nd = bytes(rng.choice(alphabet, size=int(rng.integers(0, 60))).tolist())
        assert Oracle.tokenise(nd) == Reference.tokenise(nd), nd


@needs_ref
@pytest.mark.parametrize("kind,n,limit", [("words", 5000, 10), ("geonames", 30000, 10), ("skewed", 20000, 100)])
def test_oracle_vs_live_reference(kind, n, limit):
    hay, off = {"words": W.words, "skewed": W.skewed}.get(kind, lambda n, s: W.geonames(n, 2000, s))(n, 7)
    refs = np.arange(1, n + 1, dtype=np.uint32)
    m = RawMap()
    m.put_many_packed(hay, off, refs)
    o = Oracle()
    o.put_many(hay, off)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "h.trigrams")
        m.save(path)
        ref = Reference(path)
        assert ref.stats() == o.stats() == m.stats()
        q, qo = W.queries(hay, off, 150, 8)
        for nd in W.unpack(q, qo) + [b"", b"q"]:
            assert o.find(nd, limit) == ref.find(nd, limit), nd
        ref.close()


@needs_ref
@settings(max_examples=60, deadline=None)
@given(st.lists(st.tuples(st.binary(min_size=0, max_size=12).map(lambda b: bytes(97 + (x % 27) if x % 27 < 26 else 32
                                                                               for x in b)),
                          st.integers(0, 5)), min_size=1, max_size=40, unique_by=lambda t: t[0]),
       st.binary(min_size=0, max_size=10).map(lambda b: bytes(97 + (x % 26) for x in b)),
       st.integers(1, 12))
def test_oracle_vs_live_reference_hypothesis(entries, needle, limit):
    """Tiny adversarial haystacks: duplicates of strings, tiny weights, ties everywhere."""
    m = RawMap()
    o = Oracle()
    for i, (s, w) in enumerate(entries):
        assert m.put(s, i + 1, w) == o.put(s, i + 1, w)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "h.trigrams")
        m.save(path)
        ref = Reference(path)
        assert o.find(needle, limit) == ref.find(needle, limit)
        for s, _ in entries[:5]:
            assert o.find(s, limit) == ref.find(s, limit)
        ref.close()


def test_batch_driver_equals_single_calls():
    """oracle/oracle_batch.c (the threaded driver the full-size GPU checks use) is exactly one
    oracle_find / oracle_nb_entries / oracle_tokenise per element."""
    import numpy as np
    import workloads as W
    hay, off = W.geonames(20000, 3000, 7)
    o = Oracle()
    o.put_many(hay, off)
    q, qo = W.queries(hay, off, 300, 8)
    needles = W.unpack(q, qo)
    idx = np.arange(0, 300, 3, dtype=np.uint32)
    for threads in (1, 4):
        got = o.batch(q, qo, idx=idx, limit=7, nb=True, ntri=True, threads=threads)
        for k, i in enumerate(idx):
            nd = needles[int(i)]
            assert got["rows"][k, :got["counts"][k]].tolist() == o.find(nd, 7)
            assert int(got["nb"][k]) == o.nb_entries(nd)
            assert int(got["ntri"][k]) == len(Oracle.tokenise(nd))
    every = o.batch(q, qo, find=False, nb=True)
    assert [int(v) for v in every["nb"]] == [o.nb_entries(nd) for nd in needles]


@pytest.mark.parametrize("kind", ["geonames", "skewed"])
def test_an_answer_at_a_limit_is_the_head_of_the_answer_at_a_larger_one(kind):
    """helpers.Oracle.batch_upto (the GPU tests that ask one batch at several limits run the oracle once, at the
    largest): storage.c:568-573 truncates ONE total order to the limit, so every limit's rows are the head of a larger
    limit's -- held here on every needle, incl. the skewed haystack's floods of (matches, weight) ties, and against
    the live reference where it is built.  A mutation drops the kept answer."""
    hay, off = W.geonames(30000, 4000, 17) if kind == "geonames" else W.skewed(30000, 18)
    o = Oracle()
    o.put_many(hay, off)
    q, qo = W.queries(hay, off, 400, 19)
    for limit in (149, 1, 3, 10, 64, 100):
        direct = o.batch(q, qo, limit=limit)
        cut = o.batch_upto(q, qo, limit, 149, spot=0)
        assert np.array_equal(direct["counts"], cut["counts"])
        live = np.arange(limit)[None, :] < direct["counts"][:, None].astype(np.int64)
        assert np.array_equal(np.where(live[:, :, None], direct["rows"], 0), np.where(live[:, :, None], cut["rows"], 0))
    kept = o._upto_key
    o.delete(int(o.batch(q, qo, idx=np.arange(1, dtype=np.uint32), limit=1)["rows"][0, 0, 0]) or 1)
    after = o.batch_upto(q, qo, 10, 149)                        # (asked again: the first needle's best reference is gone)
    assert o._upto_key != kept
    assert np.array_equal(after["counts"], o.batch(q, qo, limit=10)["counts"])
    if Reference.available():
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "m.trigrams")
            m = RawMap()
            m.put_many_packed(hay, off, np.arange(1, len(off), dtype=np.uint32))
            m.save(path)
            ref = Reference(path)
            o2 = Oracle()                                       # (unmutated)
            o2.put_many(hay, off)
            big = o2.batch(q, qo, limit=149)
            for k, nd in enumerate(W.unpack(q, qo)[:60]):
                for limit in (1, 10, 64):
                    assert ref.find(nd, limit) == big["rows"][k, :min(int(big["counts"][k]), limit)].tolist()
            ref.close()
