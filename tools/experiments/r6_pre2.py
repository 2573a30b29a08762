# the worker's SECOND unit of the next step loaded ahead as well (behind the scan, with the first)
EDITS = [
("kernels/needle_major.inc",
"""    pre_valid = np_ < v1;                                                        \\
    pre_live = false;                                                            \\
    if (pre_valid && wid < nn_) {                                                \\
      uint32_t x_, y_;                                                           \\
      const uint32_t even_ = kNib ? __builtin_amdgcn_readlane(bd_mine.x, 63) : 0u; \\
      BLURRILY_UNIT_OF(tb_mine, bd_mine, even_, wid, x_, y_, pre_h);             \\
      pre_live = lane8 < y_ - x_;                                                \\
      if (pre_live) pre_v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(A.ent + x_) + lane16); \\
    }                                                                            \\""",
"""    pre_valid = np_ < v1;                                                        \\
    pre_live = false; pre2_valid = false; pre2_live = false;                     \\
    if (pre_valid && wid < nn_) {                                                \\
      uint32_t x_, y_;                                                           \\
      const uint32_t even_ = kNib ? __builtin_amdgcn_readlane(bd_mine.x, 63) : 0u; \\
      BLURRILY_UNIT_OF(tb_mine, bd_mine, even_, wid, x_, y_, pre_h);             \\
      pre_live = lane8 < y_ - x_;                                                \\
      if (pre_live) pre_v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(A.ent + x_) + lane16); \\
      if (wid + kWorkers < nn_) {                                                \\
        BLURRILY_UNIT_OF(tb_mine, bd_mine, even_, wid + kWorkers, x_, y_, pre2_h); \\
        pre2_valid = true;                                                       \\
        pre2_live = lane8 < y_ - x_;                                             \\
        if (pre2_live) pre2_v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(A.ent + x_) + lane16); \\
      }                                                                          \\
    }                                                                            \\"""),
("kernels/needle_major.inc",
"""      k_ = wid + kWorkers;                                                       \\
    }                                                                            \\
    pre_valid = false;                                                           \\""",
"""      k_ = wid + kWorkers;                                                       \\
      if (pre2_valid) {                         /* ... and so is the second */   \\
        if (pend_live_) bump_unit_loaded<CT>(cnt32, pend_, pend_h_);             \\
        pend_ = pre2_v; pend_h_ = pre2_h; pend_live_ = pre2_live;                \\
        if (STATS(A)) { uint32_t x_, y_, h_; BLURRILY_UNIT_OF(tl_, bl_, even_, k_, x_, y_, h_); st_ent += min(512u, y_ - x_); } \\
        k_ += kWorkers;                                                          \\
      }                                                                          \\
    }                                                                            \\
    pre_valid = false; pre2_valid = false;                                       \\"""),
("kernels/needle_major.inc",
"""  bool pre_live = false, pre_valid = false;
  (void)rk0;""",
"""  bool pre_live = false, pre_valid = false;
  uint4 pre2_v = make_uint4(0, 0, 0, 0);
  uint32_t pre2_h = 0;
  bool pre2_live = false, pre2_valid = false;
  (void)pre2_v; (void)pre2_h; (void)pre2_live; (void)pre2_valid;
  (void)rk0;"""),
("kernels/needle_major.inc",
"""    pre_valid = false;                                          // (a unit loaded ahead is dropped)""",
"""    pre_valid = false; pre2_valid = false;                      // (units loaded ahead are dropped)"""),
]
