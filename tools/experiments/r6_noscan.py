# timing experiment (rows are wrong): the workers' hot-loop scan does nothing
EDITS = [("kernels/needle_major.inc",
"""        scan_core<CT, NT>(cnt128, nd, need, scan_cap, &ctl->thr, &ctl->floor, A.tomb, pool, scan_pool_cap, &ctl->pool_n,
                          &ctl->overflow, wbase, wlen, STATS(A) && A.path_flags ? &A.path_flags[nd.q] : nullptr,
                          hy_ >> 24, pend_list(ring, s, ring_units), &ctl->pend_n[s], pend_cap, uint32_t(NT - 64));""",
"""        (void)wbase; (void)wlen; (void)need;""")]
