# timing experiment (rows are wrong): the scan reads, clears and tests, but harvests nothing
EDITS = [("kernels/counters.inc",
"""        __builtin_amdgcn_s_setprio(3);     // a wave that found something is the one the scan barrier will wait for
        harvest(v, i);""",
"""        __builtin_amdgcn_s_setprio(3);     // a wave that found something is the one the scan barrier will wait for
        (void)harvest;""")]
