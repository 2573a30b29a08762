// find_kernels.hip -- the hot path: batched trigram find on gfx950 (MI355X).
//
// What one query computes is the reference's blurrily_storage_find
// (ext/blurrily/storage.c:477-580): tokenise the needle; visit every posting of
// every needle trigram; count, per reference, how many needle trigrams it
// shares (`matches`); return the first `limit` references in the order
// matches descending, weight ascending, reference ascending.
//
// How it is computed here (DESIGN.md "Kernels"):
//   tokenise_kernel   one lane per needle: frame, encode base-28, sort, dedup
//                     (tokeniser.c:59-119) and sum the bucket sizes
//                     (nb_entries, storage.c:498-502).
//   find_kernel       one workgroup per needle at a time (persistent
//                     workgroups pull needles from a device queue).  The rank
//                     space is swept window by window (65 536 ranks each):
//                       count  every wave streams slices of 16-bit in-window
//                              ranks (16-byte coalesced loads) and bumps a
//                              per-rank byte counter in LDS (ds_add_u32 on
//                              packed counters) -- this replaces the
//                              reference's gather + sort-by-ref + run-length
//                              pass (storage.c:506-563);
//                       scan   the counters are read back 16 bytes per lane,
//                              compared SWAR-wise against the current
//                              admission threshold and cleared; survivors
//                              fetch their weight and enter a candidate pool
//                              in LDS;
//                       select when the pool fills, a bitonic sort in LDS
//                              keeps the best `limit` and tightens the
//                              threshold -- this replaces the reference's
//                              full qsort of all matches (storage.c:566).
//                     Because windows are visited in ascending rank order and
//                     ranks are monotone in the reference, "reference
//                     ascending" among (matches, weight) ties is an explicit
//                     third sort key, not a property of libc's qsort.
// All arithmetic is integer; results are bit-exact.
#include "find_kernels.h"

#include <hip/hip_runtime.h>

#include <cerrno>
#include <cstdio>

namespace blurrily {

namespace {

constexpr uint32_t kCodeChunk = 128;   // needle trigrams staged per count pass
constexpr uint64_t kKeyInf    = ~0ull;

// --------------------------------------------------------------- tokeniser ---

__device__ __forceinline__ uint32_t dev_symbol(unsigned char c) {
  return (c >= 'a' && c <= 'z') ? uint32_t(c - 'a' + 1) : 0u;   // tokeniser.c:21-31
}

__global__ void tokenise_kernel(const char* __restrict__ packed, const uint64_t* __restrict__ offsets,
                                uint32_t n, const uint32_t* __restrict__ code_total,
                                uint16_t* __restrict__ qcodes, uint32_t* __restrict__ q_ntri,
                                uint32_t* __restrict__ q_nb, uint32_t* __restrict__ big_list,
                                uint32_t* __restrict__ big_count) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const uint64_t beg = offsets[q], end = offsets[q + 1];
  const char* s = packed + beg;
  uint16_t* out = qcodes + beg + q;              // (end-beg)+1 slots reserved per needle
  uint64_t len = 0;
  const uint64_t cap = end - beg;
  while (len < cap && s[len] != 0) ++len;        // a needle is a C string (storage.c:480)

  // frame "**" + s + "*" and encode (tokeniser.c:62-75)
  uint32_t a = 0, b = 0;
  const uint64_t m = len + 1;
  for (uint64_t k = 0; k < m; ++k) {
    const uint32_t c = (k < len) ? dev_symbol((unsigned char)s[k]) : 0u;
    out[k] = uint16_t(a + 28u * b + 784u * c);
    a = b; b = c;
  }
  // sort ascending (tokeniser.c:93)
  if (m <= 96) {
    for (uint64_t i = 1; i < m; ++i) {
      const uint16_t v = out[i];
      uint64_t j = i;
      while (j > 0 && out[j - 1] > v) { out[j] = out[j - 1]; --j; }
      out[j] = v;
    }
  } else {
    // heap sort in place for very long needles
    auto sift = [&](uint64_t root, uint64_t lim) {
      for (;;) {
        uint64_t child = 2 * root + 1;
        if (child >= lim) return;
        if (child + 1 < lim && out[child] < out[child + 1]) ++child;
        if (out[root] >= out[child]) return;
        const uint16_t t = out[root]; out[root] = out[child]; out[child] = t;
        root = child;
      }
    };
    for (uint64_t i = m / 2; i-- > 0;) sift(i, m);
    for (uint64_t lim = m; lim-- > 1;) {
      const uint16_t t = out[0]; out[0] = out[lim]; out[lim] = t;
      sift(0, lim);
    }
  }
  // drop duplicates (tokeniser.c:96-107), sum bucket sizes (storage.c:498-502)
  uint32_t d = 0;
  uint64_t nb = 0;
  for (uint64_t k = 0; k < m; ++k) {
    const uint16_t v = out[k];
    if (d == 0 || out[d - 1] != v) { out[d++] = v; nb += code_total[v]; }
  }
  q_ntri[q] = d;
  q_nb[q] = nb > 0xFFFFFFFFull ? 0xFFFFFFFFu : uint32_t(nb);
  if (d > 127) big_list[atomicAdd(big_count, 1u)] = q;
}

// ------------------------------------------------------------- find kernel ---

// Packed LDS counters.  CT = uint8_t (needles with <= 127 distinct trigrams:
// every real-world needle) or uint16_t (anything longer; the code space has
// 19 683 reachable codes so 15 bits always suffice).
template <typename CT> struct Packing;
template <> struct Packing<uint8_t> {
  static constexpr uint32_t kPerWord = 4, kBits = 8, kLog = 2, kHi = 0x80808080u, kOnes = 0x01010101u,
                            kTop = 0x80u, kMask = 0xFFu;
};
template <> struct Packing<uint16_t> {
  static constexpr uint32_t kPerWord = 2, kBits = 16, kLog = 1, kHi = 0x80008000u, kOnes = 0x00010001u,
                            kTop = 0x8000u, kMask = 0xFFFFu;
};

struct Control {            // workgroup-shared scalars
  unsigned long long thr_hi;
  uint32_t thr_rk;
  uint32_t pool_n;
  uint32_t overflow;
  uint32_t q;
  uint32_t nonempty[3];
};

template <typename CT>
__device__ __forceinline__ void bump(uint32_t* cnt32, uint32_t r) {
  using P = Packing<CT>;
  // one relaxed LDS atomic per posting; result unused -> ds_add_u32
  __hip_atomic_fetch_add(&cnt32[r >> P::kLog], 1u << ((r & (P::kPerWord - 1)) * P::kBits),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Count one 16-byte group of eight 16-bit in-window ranks; `c` is the entry
// index of its first element, [a, b) the live range of the slice.
template <typename CT>
__device__ __forceinline__ void bump8(uint32_t* cnt32, const uint4 v, uint32_t c, uint32_t a, uint32_t b) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  if (c >= a && c + 8 <= b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bump<CT>(cnt32, w[j] & 0xFFFFu);
      bump<CT>(cnt32, w[j] >> 16);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t i0 = c + 2 * j, i1 = i0 + 1;
      if (i0 >= a && i0 < b) bump<CT>(cnt32, w[j] & 0xFFFFu);
      if (i1 >= a && i1 < b) bump<CT>(cnt32, w[j] >> 16);
    }
  }
}

// Sort the candidate pool ascending by (hi, rank), keep the best `keep`, and
// tighten the admission threshold.  Called by all threads of the workgroup.
template <int NT>
__device__ void compact_pool(unsigned long long* pool_hi, uint32_t* pool_rk, Control* ctl,
                             uint32_t cap, uint32_t keep) {
  const uint32_t tid = threadIdx.x;
  const uint32_t n = min(ctl->pool_n, cap);
  uint32_t P = 1;
  while (P < n) P <<= 1;
  for (uint32_t i = n + tid; i < P; i += NT) { pool_hi[i] = kKeyInf; pool_rk[i] = 0xFFFFFFFFu; }
  __syncthreads();
  for (uint32_t size = 2; size <= P; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t i = tid; i < (P >> 1); i += NT) {
        const uint32_t lo = 2 * i - (i & (stride - 1));
        const uint32_t hi = lo + stride;
        const bool asc = (lo & size) == 0;
        const unsigned long long ah = pool_hi[lo], bh = pool_hi[hi];
        const uint32_t ar = pool_rk[lo], br = pool_rk[hi];
        const bool gt = (ah > bh) || (ah == bh && ar > br);
        if (gt == asc) { pool_hi[lo] = bh; pool_hi[hi] = ah; pool_rk[lo] = br; pool_rk[hi] = ar; }
      }
      __syncthreads();
    }
  }
  if (tid == 0) {
    ctl->pool_n = min(n, keep);
    ctl->overflow = 0;
    if (n >= keep && keep > 0) { ctl->thr_hi = pool_hi[keep - 1]; ctl->thr_rk = pool_rk[keep - 1]; }
  }
  __syncthreads();
}

template <typename CT, int NT>
__global__ __launch_bounds__(NT) void find_kernel(const FindArgs A) {
  using P = Packing<CT>;
  constexpr uint32_t kNW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // carve: counters | pool_hi | pool_rk | slice bounds | control
  uint32_t* cnt32 = reinterpret_cast<uint32_t*>(smem);
  uint4*    cnt128 = reinterpret_cast<uint4*>(smem);
  constexpr uint32_t kCntBytes = kWindowSize * sizeof(CT);
  unsigned long long* pool_hi = reinterpret_cast<unsigned long long*>(smem + kCntBytes);
  uint32_t* pool_rk = reinterpret_cast<uint32_t*>(smem + kCntBytes + size_t(A.pool_cap) * 8);
  uint32_t* s_a = pool_rk + A.pool_cap;
  uint32_t* s_b = s_a + kCodeChunk;
  Control*  ctl = reinterpret_cast<Control*>(s_b + kCodeChunk);

  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  for (uint32_t i = tid; i < kCntBytes / 16; i += NT) cnt128[i] = make_uint4(0, 0, 0, 0);
  if (tid == 0) { ctl->nonempty[0] = ctl->nonempty[1] = ctl->nonempty[2] = 0; }
  __syncthreads();

  const uint32_t n_work = A.n_work_dev ? *A.n_work_dev : A.n_work;
  uint32_t step = 0;                                   // rotates the three `nonempty` flags

  for (;;) {
    if (tid == 0) ctl->q = atomicAdd(A.queue, 1u);
    __syncthreads();
    const uint32_t slot = ctl->q;
    __syncthreads();                                   // everyone has read q before it is rewritten
    if (slot >= n_work) break;
    const uint32_t q = A.work_list ? A.work_list[slot] : slot;
    const uint32_t T = A.q_ntri[q];
    if (!A.work_list && T > 127) continue;             // long needles go to the uint16_t launch
    const uint32_t have = A.pass_base ? A.counts[q] : 0u;
    if (A.q_nb[q] == 0 || A.keep == 0 || have < A.pass_base) {
      if (tid == 0 && A.pass_base == 0) A.counts[q] = 0;
      continue;
    }
    const uint16_t* codes = A.qcodes + A.offsets[q] + q;
    // results after this key only (later passes of a limit larger than the pool)
    const bool has_floor = A.pass_base != 0;
    const unsigned long long fl_hi = has_floor ? A.floor_hi[q] : 0ull;
    const uint32_t fl_rk = has_floor ? A.floor_rk[q] : 0u;

    if (tid == 0) { ctl->pool_n = 0; ctl->overflow = 0; ctl->thr_hi = kKeyInf; ctl->thr_rk = 0xFFFFFFFFu; }
    __syncthreads();

    for (uint32_t w = 0; w < A.n_windows; ++w) {
      const uint32_t wbase = w << kWindowBits;
      const uint32_t wlen = min(kWindowSize, A.n_refs - wbase);
      const uint32_t* soff = A.slice_off + size_t(w) * kNumCodes;
      bool redo;
      do {
        redo = false;
        // ---- count: stream the needle's slices of this window ---------------
        bool touched = false;
        for (uint32_t c0 = 0; c0 < T; c0 += kCodeChunk) {
          const uint32_t tc = min(kCodeChunk, T - c0);
          const uint32_t fl = step % 3;                         // rotating flag, see DESIGN.md
          ++step;
          if (tid == 0) ctl->nonempty[(fl + 1) % 3] = 0;
          if (tid < tc) {
            const uint32_t code = codes[c0 + tid];
            const uint32_t a = soff[code], b = soff[code + 1];
            s_a[tid] = a; s_b[tid] = b;
            if (b > a) ctl->nonempty[fl] = 1;
          }
          __syncthreads();
          if (ctl->nonempty[fl]) {
            touched = true;
            for (uint32_t t = wid; t < tc; t += kNW) {
              const uint32_t a = __builtin_amdgcn_readfirstlane(s_a[t]);
              const uint32_t b = __builtin_amdgcn_readfirstlane(s_b[t]);
              if (a == b) continue;
              uint32_t c = (a & ~7u) + lane * 8;
              for (; c + 512 < b; c += 1024) {                  // two 16-byte loads in flight
                const uint4 v0 = *reinterpret_cast<const uint4*>(A.ent + c);
                const uint4 v1 = *reinterpret_cast<const uint4*>(A.ent + c + 512);
                bump8<CT>(cnt32, v0, c, a, b);
                bump8<CT>(cnt32, v1, c + 512, a, b);
              }
              if (c < b) {
                const uint4 v0 = *reinterpret_cast<const uint4*>(A.ent + c);
                bump8<CT>(cnt32, v0, c, a, b);
              }
            }
            __syncthreads();                                    // counts visible; s_a/s_b reusable
          }
        }
        if (!touched) break;                                    // nothing of this needle in the window

        // ---- scan: admit counters that can still reach the top `keep` -------
        const unsigned long long thr_hi = ctl->thr_hi;
        const uint32_t thr_rk = ctl->thr_rk;
        // counters below `need` cannot beat the current keep-th candidate
        const uint32_t need = (thr_hi == kKeyInf) ? 1u : max(1u, T - uint32_t(thr_hi >> 32));
        const uint32_t bias = (P::kTop - need) * P::kOnes;
        const uint32_t nvec = (wlen * sizeof(CT) + 15) / 16;
        for (uint32_t i = tid; i < nvec; i += NT) {
          const uint4 v = cnt128[i];
          if ((v.x | v.y | v.z | v.w) == 0) continue;
          cnt128[i] = make_uint4(0, 0, 0, 0);
          const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            uint32_t m = (wv[j] + bias) & P::kHi;
            while (m) {
              const uint32_t bit = __ffs(m) - 1;
              m &= m - 1;
              const uint32_t pos = bit / P::kBits;
              const uint32_t cnt = (wv[j] >> (pos * P::kBits)) & P::kMask;
              const uint32_t rank = wbase + (i * 4 + j) * P::kPerWord + pos;
              const uint32_t wgt = A.weight_of_rank[rank];
              const unsigned long long hi = (static_cast<unsigned long long>(T - cnt) << 32) | wgt;
              bool pass = (hi < thr_hi) || (hi == thr_hi && rank <= thr_rk);
              if (has_floor) pass = pass && ((hi > fl_hi) || (hi == fl_hi && rank > fl_rk));
              if (pass) {
                const uint32_t at = atomicAdd(&ctl->pool_n, 1u);
                if (at < A.pool_cap) { pool_hi[at] = hi; pool_rk[at] = rank; }
                else ctl->overflow = 1;
              }
            }
          }
        }
        __syncthreads();

        // ---- select: keep the pool small and the threshold tight ------------
        const uint32_t ov = ctl->overflow;
        const uint32_t pn = ctl->pool_n;
        if (ov || pn > A.pool_cap / 2) {
          compact_pool<NT>(pool_hi, pool_rk, ctl, A.pool_cap, A.keep);
          if (ov) {
            // The pool overflowed mid-window: candidates of this window were
            // lost.  Keep the tightened threshold (it is the keep-th best of a
            // subset, hence a valid bound), forget this window's survivors and
            // sweep the window again.
            if (tid == 0) {
              uint32_t j = 0;
              const uint32_t n = ctl->pool_n;
              for (uint32_t i = 0; i < n; ++i)
                if (pool_rk[i] < wbase) { pool_hi[j] = pool_hi[i]; pool_rk[j] = pool_rk[i]; ++j; }
              ctl->pool_n = j;
            }
            __syncthreads();
            redo = true;
          }
        }
      } while (redo);
    }

    // ---- emit: best `keep` in final order -----------------------------------
    compact_pool<NT>(pool_hi, pool_rk, ctl, A.pool_cap, A.keep);
    const uint32_t nres = ctl->pool_n;
    trigram_match_t* out = A.results + size_t(q) * A.limit + A.pass_base;
    for (uint32_t i = tid; i < nres; i += NT) {
      const unsigned long long hi = pool_hi[i];
      const uint32_t rk = pool_rk[i];
      trigram_match_t r;
      r.reference = A.ref_of_rank[rk];
      r.matches = T - uint32_t(hi >> 32);
      r.weight = uint32_t(hi);
      out[i] = r;
    }
    if (tid == 0) {
      A.counts[q] = A.pass_base + nres;
      if (A.floor_hi && nres > 0) { A.floor_hi[q] = pool_hi[nres - 1]; A.floor_rk[q] = pool_rk[nres - 1]; }
    }
    __syncthreads();                                   // pool reads done before the next needle resets it
  }
}

size_t find_lds_bytes(size_t counter_bytes, uint32_t pool_cap) {
  return size_t(kWindowSize) * counter_bytes + size_t(pool_cap) * 12 + 2 * kCodeChunk * 4 + sizeof(Control) + 16;
}

}  // namespace

// ------------------------------------------------------------------ launch ---

#define BLURRILY_HIP_TRY(expr)                                                        \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      std::fprintf(stderr, "blurrily_hip: %s failed: %s\n", #expr, hipGetErrorString(e_)); \
      errno = (e_ == hipErrorOutOfMemory) ? ENOMEM : EIO;                             \
      return -1;                                                                      \
    }                                                                                 \
  } while (0)

int launch_tokenise(const TokeniseArgs& t, hipStream_t stream) {
  if (t.n == 0) return 0;
  const uint32_t block = 128;
  const uint32_t grid = (t.n + block - 1) / block;
  hipLaunchKernelGGL(tokenise_kernel, dim3(grid), dim3(block), 0, stream, t.packed, t.offsets, t.n,
                     t.code_total, t.qcodes, t.q_ntri, t.q_nb, t.big_list, t.big_count);
  BLURRILY_HIP_TRY(hipGetLastError());
  return 0;
}

uint32_t find_pool_cap(uint32_t keep) {
  uint32_t cap = 1024;
  while (cap < 4 * keep && cap < 4096) cap <<= 1;
  return cap;
}

int launch_find(const FindArgs& a, bool long_needles, uint32_t grid, hipStream_t stream) {
  constexpr int NT = 256;
  if (grid == 0) return 0;
  if (!long_needles) {
    const size_t lds = find_lds_bytes(1, a.pool_cap);
    static bool attr_done = false;
    if (!attr_done) {
      BLURRILY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&find_kernel<uint8_t, NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_done = true;
    }
    hipLaunchKernelGGL((find_kernel<uint8_t, NT>), dim3(grid), dim3(NT), lds, stream, a);
  } else {
    const size_t lds = find_lds_bytes(2, a.pool_cap);
    static bool attr_done = false;
    if (!attr_done) {
      BLURRILY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&find_kernel<uint16_t, NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_done = true;
    }
    hipLaunchKernelGGL((find_kernel<uint16_t, NT>), dim3(grid), dim3(NT), lds, stream, a);
  }
  BLURRILY_HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace blurrily
