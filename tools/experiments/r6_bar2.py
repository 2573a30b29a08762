# sensitivity experiment (rows stay exact): every LDS barrier of the sweep twice
EDITS = [('kernels/needle_major.inc', '__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory"); }', '__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier\\n\\ts_barrier" ::: "memory"); }')]
