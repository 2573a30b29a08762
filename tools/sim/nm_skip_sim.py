"""CPU simulation of leaving dense slices out of the NEEDLE-MAJOR sweep (sweep_coop with
left-out slices): one needle at a time, windows visited upward from the needle's own length class and round the
end, the threshold tightened after every window; with a threshold, the L largest dense slices of the needle in a
window are left out (L <= need - cmin), candidates are the ranks whose counted matches reach need - L, their exact
counts come from the left-out slices.  The result is ASSERTED equal to the full count.  Prints, per (cmin, dense
threshold): postings counted, candidates harvested and bitmap probes per needle and per visited window.

    python tools/sim/nm_skip_sim.py [haystack strings = 2000000] [needles = 200]
"""
import sys

from common import tokenise_all
import numpy as np
import workloads as W


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
    NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    K = 10
    scale = n / 8423769
    hay, off = W.geonames(n, max(1000, int(500000 * min(1.0, scale * 4))), 3)
    sid, code, lens = tokenise_all(hay, off)
    order = np.lexsort((np.arange(n), lens))
    rank_of = np.empty(n, dtype=np.int64); rank_of[order] = np.arange(n)
    wsorted = lens[order]
    prank = rank_of[sid]
    o = np.lexsort((prank, code))
    pcode, prank = code[o], prank[o]
    cstart = np.searchsorted(pcode, np.arange(21953))
    ntri_ref = np.bincount(rank_of[sid], minlength=n)
    WR = 65520
    nwin = (n + WR - 1) // WR
    win_max = np.array([ntri_ref[w * WR:(w + 1) * WR].max() for w in range(nwin)])
    start_win = np.array([min(nwin - 1, np.searchsorted(wsorted, L) // WR) for L in range(256)])
    qp, qo = W.queries(hay, off, NQ, 3000)
    qsid, qcode, qlens = tokenise_all(qp, qo)
    configs = [(0, 1 << 30)] + [(cmin, dth) for cmin in (2, 3, 4) for dth in (1024, 2048, 4096)]
    agg = {c: dict(counted=0, cand=0, probes=0, visited=0, skipped=0, left=0, cand_max=0, admitted=0) for c in configs}
    tot_post = 0
    hist_need = np.zeros(80, dtype=np.int64)
    for q in range(NQ):
        codes = qcode[qsid == q]; T = len(codes)
        lists = [prank[cstart[c]:cstart[c + 1]] for c in codes]
        tot_post += sum(len(l) for l in lists)
        bounds = [np.searchsorted(l, np.arange(nwin + 1) * WR) for l in lists]
        own = int(start_win[min(255, int(qlens[q]))]) & ~1
        allp = np.concatenate(lists)
        r, c = np.unique(allp, return_counts=True)
        exact = sorted(zip((T - c).tolist(), r.tolist()))[:K]
        for cfg in configs:
            cmin, dth = cfg
            d = agg[cfg]
            pool = []
            for i in range(nwin):
                w = own + i if own + i < nwin else own + i - nwin
                t = pool[K - 1] if len(pool) >= K else None
                if t is None:
                    need = 1
                else:
                    mk = T - t[0]
                    need = max(1, mk if t[1] >= w * WR else mk + 1)
                if min(T, win_max[w]) < need:
                    d['skipped'] += 1
                    continue
                d['visited'] += 1
                if cfg == configs[1]:
                    hist_need[min(79, need)] += 1
                wl = [l[b[w]:b[w + 1]] for l, b in zip(lists, bounds)]
                sizes = np.array([len(x) for x in wl])
                idx = np.argsort(-sizes, kind='stable')
                Lmax = max(0, need - cmin) if (t is not None and cmin > 0) else 0
                L = min(Lmax, int((sizes >= dth).sum()), 8)
                cold = [wl[j] for j in idx[L:]]
                ncold = sum(len(x) for x in cold)
                d['counted'] += ncold; d['left'] += L
                if ncold == 0:
                    continue
                rr, cc = np.unique(np.concatenate(cold), return_counts=True)
                sel = cc >= (need - L)
                ns = int(sel.sum())
                if L:
                    d['cand'] += ns; d['probes'] += ns * L; d['cand_max'] = max(d['cand_max'], ns)
                rc = rr[sel]; tc = cc[sel].copy()
                for j in idx[:L]:
                    tc += np.isin(rc, wl[j])
                keys = list(zip((T - tc).tolist(), rc.tolist()))
                if t is not None:
                    keys = [k for k in keys if k <= t]
                d['admitted'] += len(keys)
                pool = sorted(pool + keys)[:K]
            assert pool == exact, (q, cfg, pool, exact)
    print('n', n, 'windows', nwin, 'needles', NQ, 'postings/needle %.0f' % (tot_post / NQ))
    print('need histogram (visited windows, cmin 2 row):', {i: int(v) for i, v in enumerate(hist_need) if v})
    for cfg in configs:
        d = agg[cfg]
        v = max(1, d['visited'])
        print('cmin %d dense>=%-10d visited/needle %.1f (skipped %.1f)  counted/needle %.0f (%.1f%% of nb_entries)  '
              'per visited window: counted %.0f left-out %.2f cand %.1f (max %d) probes %.1f admitted %.2f' % (
                  cfg[0], cfg[1], d['visited'] / NQ, d['skipped'] / NQ, d['counted'] / NQ, 100 * d['counted'] / tot_post,
                  d['counted'] / v, d['left'] / v, d['cand'] / v, d['cand_max'], d['probes'] / v, d['admitted'] / v))


main()
