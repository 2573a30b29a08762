# latency mode: four of a worker's units loaded before the first is counted (the chip is not full: a step waits for round trips)
EDITS = [
("kernels/needle_major.inc",
"""    pre_valid = false;                                                           \\
    for (; k_ < (n_); k_ += kWorkers) {                                          \\
      uint32_t x_, y_, h_;                                                       \\""",
"""    pre_valid = false;                                                           \\
    if constexpr (LEARNS) {                                                      \\
      for (; k_ < (n_); k_ += 4 * kWorkers) {                                    \\
        uint4 v4_[4]; uint32_t h4_[4]; bool l4_[4];                              \\
        _Pragma("unroll") for (uint32_t i_ = 0; i_ < 4; ++i_) {                  \\
          const uint32_t kk_ = k_ + i_ * kWorkers;                               \\
          v4_[i_] = make_uint4(0, 0, 0, 0); h4_[i_] = 0; l4_[i_] = false;        \\
          if (kk_ < (n_)) {                                                      \\
            uint32_t x_, y_;                                                     \\
            BLURRILY_UNIT_OF(tl_, bl_, even_, kk_, x_, y_, h4_[i_]);             \\
            l4_[i_] = lane8 < y_ - x_;                                           \\
            if (l4_[i_]) v4_[i_] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(A.ent + x_) + lane16); \\
            if (STATS(A)) st_ent += min(512u, y_ - x_);                          \\
          }                                                                      \\
        }                                                                        \\
        if (pend_live_) bump_unit_loaded<CT>(cnt32, pend_, pend_h_);             \\
        pend_live_ = false;                                                      \\
        _Pragma("unroll") for (uint32_t i_ = 0; i_ < 4; ++i_)                    \\
          if (l4_[i_]) bump_unit_loaded<CT>(cnt32, v4_[i_], h4_[i_]);            \\
      }                                                                          \\
    }                                                                            \\
    for (; k_ < (n_); k_ += kWorkers) {                                          \\
      uint32_t x_, y_, h_;                                                       \\"""),
]
