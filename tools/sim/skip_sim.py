"""CPU simulation of the window-major sweep's algorithm (wsweep_kernel) on a slice of
the bench workload: thresholds progress exactly as on the GPU (own window pair first, then the windows in
order), the L largest dense slices are left out, candidates are the ranks whose cold count reaches need - L,
their exact counts come from the left-out slices -- and the result is ASSERTED equal to the full count.
Prints, per (cmin, dense threshold): tasks, postings counted, candidates and probes per needle.

    python tools/sim/skip_sim.py [haystack strings = 1000000] [needles = 300]
"""
import sys
import time

from common import tokenise_all
import numpy as np
import workloads as W

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 300
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
    WR = 65535
    nwin = (n + WR - 1) // WR
    win_max = np.array([ntri_ref[w*WR:(w+1)*WR].max() for w in range(nwin)])
    start_win = np.array([min(nwin - 1, np.searchsorted(wsorted, L) // WR) for L in range(256)])
    qp, qo = W.queries(hay, off, NQ, 3000)
    qsid, qcode, qlens = tokenise_all(qp, qo)
    configs = [(cmin, dth) for cmin in (1, 2, 3) for dth in (512, 1024, 2048, 4096)]
    agg = {c: dict(cold=0, cand=0, probes=0, tasks=0, wide=0, skipped_tasks=0, hotbytes=0) for c in configs}
    tot_post = 0; tot_visit_old = 0
    for q in range(NQ):
        codes = qcode[qsid == q]; T = len(codes)
        lists = [prank[cstart[c]:cstart[c+1]] for c in codes]
        tot_post += sum(len(l) for l in lists)
        wl = [[l[(l >= w*WR) & (l < (w+1)*WR)] for l in lists] for w in range(nwin)]
        own = int(start_win[min(255, int(qlens[q]))]) & ~1
        for cfg in configs:
            cmin, dth = cfg
            d = agg[cfg]
            pool = []   # list of keys (T-m, rank)
            def thr():
                return pool[K-1] if len(pool) >= K else None
            def admit(w, cnt_pairs):
                nonlocal pool
                pool = sorted(pool + cnt_pairs)[:K]
            # phase 1: own pair, full count
            for w in (own, own + 1):
                if w >= nwin: continue
                allp = np.concatenate(wl[w]) if wl[w] else np.zeros(0, dtype=np.int64)
                if len(allp) == 0: continue
                r, c = np.unique(allp, return_counts=True)
                t = thr()
                keys = list(zip((T - c).tolist(), r.tolist()))
                if t is not None: keys = [k for k in keys if k <= t]
                admit(w, keys)
            for w in range(nwin):
                if w in (own, own + 1): continue
                t = thr()
                if t is None: need = 1
                else:
                    mk = T - t[0]
                    need = max(1, mk if t[1] >= w*WR else mk + 1)
                if min(T, win_max[w]) < need:
                    d['skipped_tasks'] += 1; continue
                d['tasks'] += 1
                sizes = np.array([len(x) for x in wl[w]])
                idx = np.argsort(-sizes)
                Lmax = max(0, need - cmin)
                elig = int((sizes >= dth).sum())
                L = min(Lmax, elig)
                coldl = [wl[w][i] for i in idx[L:]]
                ncold = sum(len(x) for x in coldl)
                d['cold'] += ncold; d['hotbytes'] += L * 8192
                if T - L > 15 and win_max[w] > 15: d['wide'] += 1
                if ncold == 0: continue
                r, c = np.unique(np.concatenate(coldl), return_counts=True)
                sel = c >= (need - L)
                d['cand'] += int(sel.sum()); d['probes'] += int(sel.sum()) * L
                # exact totals for candidates
                rc = r[sel]; cc = c[sel].copy()
                for i in idx[:L]:
                    cc += np.isin(rc, wl[w][i])
                keys = list(zip((T - cc).tolist(), rc.tolist()))
                if t is not None: keys = [k for k in keys if k <= t]
                admit(w, keys)
            # verify vs exact
            if cfg == configs[0]:
                allp = np.concatenate(lists)
                r, c = np.unique(allp, return_counts=True)
                exact = sorted(zip((T - c).tolist(), r.tolist()))[:K]
                ref_pool = exact
            assert pool == ref_pool, (q, cfg, pool, ref_pool)
    print('n', n, 'windows', nwin, 'needles', NQ, 'postings/needle %.0f' % (tot_post / NQ))
    for cfg in configs:
        d = agg[cfg]
        print('cmin %d dense>=%d: tasks/needle %.1f (skipped %.1f, wide %.1f) cold/needle %.0f (%.1f%%) cand/needle %.0f probes/needle %.0f  per task: cold %.0f cand %.1f probes %.1f' % (
            cfg[0], cfg[1], d['tasks']/NQ, d['skipped_tasks']/NQ, d['wide']/NQ, d['cold']/NQ, 100*d['cold']/tot_post, d['cand']/NQ, d['probes']/NQ,
            d['cold']/max(1,d['tasks']), d['cand']/max(1,d['tasks']), d['probes']/max(1,d['tasks'])))
main()
