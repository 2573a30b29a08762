# one_merge: limits up to 11 -- the keep-th smallest list HEAD bounds the answer; the keys not above it (at most keep x keep) are
# gathered from the registers they were loaded into and ordered by counting: no levels, no second read of the lists
EDITS = [
("kernels/one.inc",
"""#pragma unroll 1
  for (uint32_t base = tid; base < total; base += kRound * NT) {       // (base - tid: uniform)
    unsigned long long k[kRound];""",
"""  if (keep <= kOneFastKeep && total <= kRound * NT) {
    // Small limits: a list is sorted, so the keep-th smallest list HEAD is a key that at least `keep` keys do not exceed: the
    // answer lies among the keys not above it -- of at most `keep` lists (heads are distinct), at most keep x keep keys.
    // They go from the registers they were loaded into straight into the pool and are put in order by counting below:
    // no levels, no places, no second read of the lists.
    unsigned long long* const heads = reinterpret_cast<unsigned long long*>(reach);   // [G]
    unsigned long long k[kRound];
#pragma unroll
    for (uint32_t j = 0; j < kRound; ++j) {
      const uint32_t idx = tid + j * NT;
      k[j] = kKeyInf;
      if (idx < total) k[j] = slot(idx);
    }
#pragma unroll
    for (uint32_t j = 0; j < kRound; ++j) {
      const uint32_t idx = tid + j * NT;
      if (idx < total) { const uint32_t li = idx / keep; if (idx == li * keep) heads[li] = k[j]; }
    }
    __syncthreads();
    ONE_MARK(A, 10);
    if (tid < G) {
      const unsigned long long mine = heads[tid];
      uint32_t lower = 0;
      for (uint32_t j = 0; j < G; ++j) lower += heads[j] < mine ? 1u : 0u;          // (one address per read: broadcasts)
      if (lower == keep - 1) ctl->thr = mine;            // (the keep-th smallest; fewer than keep lists hold a key: every key passes)
    }
    __syncthreads();
    ONE_MARK(A, 11);
    const unsigned long long bound = ctl->thr;
#pragma unroll
    for (uint32_t j = 0; j < kRound; ++j)
      if (k[j] != kKeyInf && k[j] <= bound) pool[atomicAdd(&ctl->pool_n, 1u)] = k[j];
    __syncthreads();
  } else {
#pragma unroll 1
  for (uint32_t base = tid; base < total; base += kRound * NT) {       // (base - tid: uniform)
    unsigned long long k[kRound];"""),
("kernels/one.inc",
"""    pool[tid] = slot(lo * keep + (tid - off[lo]));
  }
  __syncthreads();
  ONE_MARK(A, 15);""",
"""    pool[tid] = slot(lo * keep + (tid - off[lo]));
  }
  __syncthreads();
  }
  const uint32_t n_out = ctl->pool_n;                    // (the fast way: up to keep x keep keys, of which the first keep matter)
  ONE_MARK(A, 15);"""),
("kernels/one.inc",
"""  __syncthreads();
  const uint32_t n_out = ctl->pool_n;
  if (tid < n_out) {                                     // the list whose prefix holds output slot tid: the last with off <= tid""",
"""  __syncthreads();
  const uint32_t n_out = ctl->pool_n;
  if (tid < n_out) {                                     // the list whose prefix holds output slot tid: the last with off <= tid"""),
("kernels/one.inc",
"""    if (tid < n_out) pool[to] = key;                     // (keys are distinct: the places are a permutation)
    __syncthreads();
  }
}""",
"""    if (tid < n_out) pool[to] = key;                     // (keys are distinct: the places are a permutation)
    if (tid == 0 && n_out > keep) ctl->pool_n = keep;
    __syncthreads();
  }
}"""),
("kernels/one.inc",
"""  constexpr uint32_t kPadLevel = 64;                                   // (a needle's levels: 0 .. T - 1 <= 63)""",
"""  constexpr uint32_t kPadLevel = 64;                                   // (a needle's levels: 0 .. T - 1 <= 63)
  constexpr uint32_t kOneFastKeep = 11;                                // (11 x 11 keys fit the 128 the ordering below takes)"""),
]
