# timing experiment (rows are wrong): the workers count nothing (no loads, no atomics), and load nothing ahead
EDITS = [("kernels/needle_major.inc",
"""      } else {
        BLURRILY_COUNT_UNITS(s, n_units, true);
        PHASE_MARK(2);                                          // units counted
      }""",
"""      } else {
        pre_valid = false;
      }"""),
("kernels/needle_major.inc",
"""        if (tr_) TRACE_MARK(A, nd.q, e, 0u, 3u);
        BLURRILY_PRELOAD();""",
"""        if (tr_) TRACE_MARK(A, nd.q, e, 0u, 3u);""")]
