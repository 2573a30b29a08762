# the scan's first test decided per WAVE (a ballot and a scalar branch) before any lane's exec mask is touched
EDITS = [("kernels/counters.inc",
"""      if (S::maybe(v, nq) && S::any_hit(v, nq)) {
        __builtin_amdgcn_s_setprio(3);     // a wave that found something is the one the scan barrier will wait for
        harvest(v, i);
      }""",
"""      const bool mb_ = S::maybe(v, nq);
      if (__ballot(mb_) != 0) {
        if (mb_ && S::any_hit(v, nq)) {
          __builtin_amdgcn_s_setprio(3);     // a wave that found something is the one the scan barrier will wait for
          harvest(v, i);
        }
      }""")]
