"""CPU simulation: how many rank windows a needle still has to visit if references of equal weight were
ordered by their string instead of by reference (so that a window holds similar strings) and a window were
skipped when fewer than `need` of the needle's trigrams occur in it at all.  Full configs[2] haystack:
102.5 -> 88.7 windows per needle (-13 %): not worth giving up rank = (weight, reference) order for.

    python tools/sim/window_prune_sim.py [haystack strings = 8423769] [needles = 150]
"""
import sys
import time

from common import tokenise_all
import numpy as np
import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8423769
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 200
K = 10
scale = n / 8423769
t0 = time.time()
hay, off = W.geonames(n, max(1000, int(500000 * min(1.0, scale * 4))), 3)
sid, code, lens = tokenise_all(hay, off)
print('tokenised', len(sid), time.time() - t0, flush=True)
off64 = off.astype(np.int64)
# 8-byte big-endian prefix key of every string (for the alphabetical order inside a length class)
pad = np.zeros(len(hay) + 16, dtype=np.uint8); pad[:len(hay)] = hay
key = np.zeros(n, dtype=np.uint64)
for b in range(8):
    ch = pad[off64[:-1] + b].astype(np.uint64)
    ch = np.where(b < lens, ch, 0)
    key = (key << np.uint64(8)) | ch
WR = 65535
nwin = (n + WR - 1) // WR
qp, qo = W.queries(hay, off, NQ, 3000)
qsid, qcode, qlens = tokenise_all(qp, qo)

def layout(order):
    rank_of = np.empty(n, dtype=np.int64); rank_of[order] = np.arange(n)
    prank = rank_of[sid]
    win = prank // WR
    P = np.zeros((nwin, 21952), dtype=bool)
    P[win, code] = True
    ntri_ref = np.bincount(prank, minlength=n)
    wmt = np.array([ntri_ref[w*WR:(w+1)*WR].max() for w in range(nwin)])
    return rank_of, P, wmt

orders = {
    'by (len, ref)  [current]': np.lexsort((np.arange(n), lens)),
    'by (len, string prefix)': np.lexsort((np.arange(n), key, lens)),
}
# exact need per needle (final threshold: 10th best match count)
o = np.lexsort((sid, code))   # postings by code
pc, ps = code[o], sid[o]
cstart = np.searchsorted(pc, np.arange(21953))
needs = []
Ts = []
for q in range(NQ):
    codes = qcode[qsid == q]
    allp = np.concatenate([ps[cstart[c]:cstart[c+1]] for c in codes])
    cnt = np.bincount(allp, minlength=n)
    top = np.partition(cnt, n - K)[n - K]
    needs.append(int(max(1, top))); Ts.append(len(codes))
print('mean T %.1f mean need %.1f' % (np.mean(Ts), np.mean(needs)), flush=True)
for name, order in orders.items():
    rank_of, P, wmt = layout(order)
    vis_wmt = vis_nz = 0
    for q in range(NQ):
        codes = qcode[qsid == q]
        nz = P[:, codes].sum(axis=1)
        a = np.minimum(Ts[q], wmt) >= needs[q]
        b = a & (nz >= needs[q])
        vis_wmt += a.sum(); vis_nz += b.sum()
    print('%-28s windows %d: visited by length bound %.1f, with presence bound %.1f per needle' % (name, nwin, vis_wmt / NQ, vis_nz / NQ), flush=True)
