# the scan two trips a turn, ping-pong: the next vector's read is in flight while the current one is looked at
EDITS = [("kernels/counters.inc",
"""    for (uint32_t i = tid; i < nvec; i += stride) {
      uint4 v = cnt128[i];
      cnt128[i] = zq;
      v = S::mask_pad(v, i);
      // one AND per vector, then one SWAR test: the top bit of a field is set iff its counter >= need
      if (S::maybe(v, nq) && S::any_hit(v, nq)) {
        __builtin_amdgcn_s_setprio(3);     // a wave that found something is the one the scan barrier will wait for
        harvest(v, i);
      }
    }""",
"""    auto look = [&](const uint4 v, const uint32_t i) {
      // one AND per vector, then one SWAR test: the top bit of a field is set iff its counter >= need
      if (S::maybe(v, nq) && S::any_hit(v, nq)) {
        __builtin_amdgcn_s_setprio(3);     // a wave that found something is the one the scan barrier will wait for
        harvest(v, i);
      }
    };
    uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
    uint32_t i = tid;
    if (i < nvec) va = cnt128[i];
    while (i < nvec) {
      const uint32_t i1 = i + stride;
      vb = cnt128[i1 < nvec ? i1 : i];                 // (in flight while va is looked at; past the end: va's own, unused)
      cnt128[i] = zq;
      look(va, i);
      if (i1 >= nvec) break;
      const uint32_t i2 = i1 + stride;
      va = cnt128[i2 < nvec ? i2 : i1];                // (in flight while vb is looked at)
      cnt128[i1] = zq;
      look(vb, i1);
      i = i2;
    }""")]
