# one_select: the search for the bound three values a pass (quartiles), one barrier a pass
EDITS = [
("kernels/one.inc",
"""    while (lo < hi) {                                    // (at most seven passes with byte counters, four with 4-bit ones)
      const uint32_t mid = (lo + hi + 1) >> 1;
      if (tally(mid) >= keep) lo = mid; else hi = mid - 1;
    }""",
"""    while (lo < hi) {                                    // (three bounds a pass: at most three passes with byte counters, two with 4-bit ones)
      const uint32_t span = hi - lo;                     // lo < m1 <= m2 <= m3 <= hi
      const uint32_t m1 = lo + (span + 3) / 4, m2 = lo + (span + 1) / 2, m3 = lo + (3 * span + 3) / 4;
      const uint32_t i12 = wave_inclusive_sum(reach(m1) | (reach(m2) << 16)), i3 = wave_inclusive_sum(reach(m3));
      if (lane == 63) {
        if (i12 & 0xFFFFu) atomicAdd(&sh->tally[pass], i12 & 0xFFFFu);
        if (i12 >> 16) atomicAdd(&sh->tally[pass + 1], i12 >> 16);
        if (i3) atomicAdd(&sh->tally[pass + 2], i3);
      }
      __syncthreads();
      const uint32_t t1 = sh->tally[pass], t2 = sh->tally[pass + 1], t3 = sh->tally[pass + 2];
      pass += 3;
      if (t3 >= keep) lo = m3;
      else if (t2 >= keep) { lo = m2; hi = m3 - 1; }
      else if (t1 >= keep) { lo = m1; hi = m2 - 1; }
      else hi = m1 - 1;
    }"""),
]
