# the unit loaded ahead lives in the registers the count loop carries its unit in flight in (no copy, no select)
EDITS = [("kernels/needle_major.inc",
"""    uint4 pend_ = make_uint4(0, 0, 0, 0);                                        \\
    uint32_t pend_h_ = 0;                                                        \\
    bool pend_live_ = false;                                                     \\
    uint4 tl_ = tb_mine;                        /* (read a step ago, behind the count barrier) */ \\""",
"""    uint4 pend_ = pre_v;                                                         \\
    uint32_t pend_h_ = pre_h;                                                    \\
    bool pend_live_ = (have_mine_) && pre_valid && pre_live;                     \\
    uint4 tl_ = tb_mine;                        /* (read a step ago, behind the count barrier) */ \\"""),
("kernels/needle_major.inc",
"""    if ((have_mine_) && pre_valid) {            /* the first unit is on its way since the scan before */ \\
      pend_ = pre_v; pend_h_ = pre_h; pend_live_ = pre_live;                     \\
      if (STATS(A) && wid < (n_)) {                                              \\
        uint32_t x_, y_, h_;                                                     \\
        BLURRILY_UNIT_OF(tl_, bl_, even_, wid, x_, y_, h_);                      \\""",
"""    if ((have_mine_) && pre_valid) {            /* the first unit is on its way since the scan before */ \\
      if (STATS(A) && wid < (n_)) {                                              \\
        uint32_t x_, y_, h_;                                                     \\
        BLURRILY_UNIT_OF(tl_, bl_, even_, wid, x_, y_, h_);                      \\""")]
