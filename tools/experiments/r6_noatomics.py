# timing experiment (rows are wrong): units are loaded but not counted
EDITS = [("kernels/counters.inc",
"""template <typename CT>
__device__ __forceinline__ void bump_unit_loaded(uint32_t* cnt32, const uint4 v, uint32_t half) {
  static_assert(sizeof(CT) == 1 || std::is_same<CT, Nib>::value, "byte and 4-bit counters only");""",
"""template <typename CT>
__device__ __forceinline__ void bump_unit_loaded(uint32_t* cnt32, const uint4 v, uint32_t half) {
  asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); (void)cnt32; (void)half; return;
  static_assert(sizeof(CT) == 1 || std::is_same<CT, Nib>::value, "byte and 4-bit counters only");""")]
