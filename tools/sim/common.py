"""Shared by the CPU simulations of tools/sim/: the tokeniser (tokeniser.c:59-119) vectorised in numpy."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests"), ROOT]

import numpy as np  # noqa: E402


def tokenise_all(packed, off):
    """(string id, code) pairs, deduplicated."""
    n = len(off) - 1
    off = off.astype(np.int64)
    lens = off[1:] - off[:-1]
    sym = np.zeros(256, dtype=np.int64); sym[ord('a'):ord('z')+1] = np.arange(1, 27)
    s = sym[packed]
    # positions: for string i, k in 0..len
    tot = int((lens + 1).sum())
    sid = np.repeat(np.arange(n), lens + 1)
    start = np.repeat(off[:-1], lens + 1)
    first = np.cumsum(lens + 1) - (lens + 1)
    k = np.arange(tot) - np.repeat(first, lens + 1)
    ln = np.repeat(lens, lens + 1)
    sp = np.concatenate([s, [0, 0]])
    def at(j):  # symbol at string index j (may be <0 or >=len -> 0)
        ok = (j >= 0) & (j < ln)
        return np.where(ok, sp[np.clip(start + j, 0, len(s) - 1)], 0)
    code = at(k - 2) + 28 * at(k - 1) + 784 * at(k)
    key = sid * 21952 + code
    key = np.unique(key)
    return key // 21952, key % 21952, lens
