# sensitivity: two more dependent global round trips in front of every sweep (the manager's codes and first table loaded a
# second time, each address depending on what the load before returned) -- what the start-up chain of a sweep is worth
EDITS = [
("kernels/needle_major.inc",
"""    mcode = lane < tc ? codes[lane] : 0u;
    BLURRILY_FETCH_TABLE(BLURRILY_STEP_AT(0u), mcode, ta, tb, ta1, tb1);
""",
"""    mcode = lane < tc ? codes[lane] : 0u;
    BLURRILY_FETCH_TABLE(BLURRILY_STEP_AT(0u), mcode, ta, tb, ta1, tb1);
    { uint32_t z_; asm volatile("v_and_b32 %0, 0, %1" : "=v"(z_) : "v"(ta ^ tb ^ ta1 ^ tb1));
      mcode = lane < tc ? codes[lane + z_] : 0u;
      asm volatile("v_and_b32 %0, 0, %1" : "=v"(z_) : "v"(mcode));
      BLURRILY_FETCH_TABLE(BLURRILY_STEP_AT(0u), mcode + z_, ta, tb, ta1, tb1); }
"""),
]
