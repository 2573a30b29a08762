# the zero quad by two 64-bit moves
EDITS = [("kernels/counters.inc",
"""  uint4 zq;
  asm volatile("v_mov_b32 %0, 0\\n\\tv_mov_b32 %1, 0\\n\\tv_mov_b32 %2, 0\\n\\tv_mov_b32 %3, 0"
               : "=v"(zq.x), "=v"(zq.y), "=v"(zq.z), "=v"(zq.w));""",
"""  uint4 zq;
  { unsigned long long z0_, z1_;
    asm volatile("v_mov_b64 %0, 0\\n\\tv_mov_b64 %1, 0" : "=v"(z0_), "=v"(z1_));
    zq = make_uint4(uint32_t(z0_), uint32_t(z0_ >> 32), uint32_t(z1_), uint32_t(z1_ >> 32)); }""")]
