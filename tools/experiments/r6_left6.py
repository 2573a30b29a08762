EDITS = [("kernels/needle_major.inc", "constexpr uint32_t kNmMaxLeftOut = 8;", "constexpr uint32_t kNmMaxLeftOut = 6;")]
