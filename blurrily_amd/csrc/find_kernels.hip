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

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <type_traits>

// This file is compiled twice (csrc/Makefile): as it is -- the kernels that are timed, with no trace of
// the request counters -- and through find_kernels_counted.hip with BLURRILY_COUNTED defined, into
// namespace blurrily::counted: the same kernels keeping FindArgs::stats (blurrily_storage_set_stats) and, when
// FindArgs::phase_clocks is set, wave 0's shader clocks per phase of the sweep (tools/phase_profile.py).
// (Counting behind a run-time `if (A.stats)` cost the needle-major kernel 7 %: registers and branches
// in its innermost loops, measured on one box, tools/ab_probe.py.)  This is the file's only build switch.
#ifdef BLURRILY_COUNTED
#define STATS(A) ((A).stats)
// one thread marks the path a needle's find takes (FindArgs::path_flags)
#define PATH_FLAG(A, q_, bits_) do { if ((A).path_flags && (threadIdx.x & 63u) == 0) atomicOr(&(A).path_flags[q_], (bits_)); } while (0)
#define BLURRILY_KERNELS_BEGIN namespace counted {
#define BLURRILY_KERNELS_END }
#else
#define STATS(A) (static_cast<unsigned long long*>(nullptr))
#define PATH_FLAG(A, q_, bits_) do { } while (0)
#define BLURRILY_KERNELS_BEGIN
#define BLURRILY_KERNELS_END
#endif

// (temporary) trace build of the TIMED kernels: shader-clock stamps of the steps of needles [kTraceQ0, kTraceQ0 + 64),
// wave 0 (role 0) and the manager wave (role 1), eight marks per step, 64 steps per needle
#if defined(BLURRILY_TRACE) && !defined(BLURRILY_COUNTED)
#define TRACE_MARK(A, q_, e_, role_, mark_) do { if ((A).phase_clocks && (q_) - 200000u < 64u && (e_) < 64u) { \
    const unsigned long long t_ = clock64(); \
    if ((threadIdx.x & 63u) == 0) (A).phase_clocks[(((((q_) - 200000u) * 64u + (e_)) * 2u + (role_)) * 8u) + (mark_)] = t_; } } while (0)
#define TRACE_MARK_AT(A, base_, q_, e_, role_, mark_) do { if ((A).phase_clocks && (q_) - (base_) < 64u && (e_) < 64u) { \
    const unsigned long long t_ = clock64(); \
    if ((threadIdx.x & 63u) == 0) (A).phase_clocks[(((((q_) - (base_)) * 64u + (e_)) * 2u + (role_)) * 8u) + (mark_)] = t_; } } while (0)
// find_one_kernel: the device's 100 MHz wall clock at mark i of workgroup blockIdx.x, phase_clocks[blockIdx.x * 16 + i]
#define ONE_MARK(A, i_) do { if ((A).phase_clocks && threadIdx.x == 0) (A).phase_clocks[blockIdx.x * 16u + (i_)] = wall_clock64(); } while (0)
#else
#define TRACE_MARK(A, q_, e_, role_, mark_) do { } while (0)
#define TRACE_MARK_AT(A, base_, q_, e_, role_, mark_) do { } while (0)
#define ONE_MARK(A, i_) do { } while (0)
#endif

namespace blurrily {
BLURRILY_KERNELS_BEGIN

namespace {

constexpr uint32_t kCodeChunk = 128;   // needle trigrams staged per count pass

constexpr int      kFindThreads    = 1024; // find_kernel's workgroup: sixteen waves, two workgroups per CU (256 and 512 were
                                           // measured slower in round 1 and are no longer built)
constexpr uint32_t kRankSortMax    = 512;   // pools up to this size are compacted by rank counting (two keys per read), larger ones by
                                            // a bitonic sort (256 -> 512: -0.7 % at configs[2]; 1 024: +1 % at configs[4], limit 100)
constexpr int      kSerialPrio     = 2;    // wave priority in a workgroup's serial sections (between needles, compaction, cold start)

// Phase profile (counted build only): wave 0's shader-clock time per phase of the sweep,
// accumulated per workgroup into FindArgs::phase_clocks[blockIdx.x * 16 + phase].
#ifdef BLURRILY_COUNTED
#define PHASE_DECL unsigned long long ph_last = clock64(), ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_units = 0, ph_lanes = 0
// head units of wave 0 and their live lanes (lane utilisation of the LDS atomics)
// per needle, in the kernel body: slot 10 setup (queue pop .. first sweep), 11 the sweeps, 12 final compaction and rows
#define PHASE_NEEDLE_DECL unsigned long long pn_last = clock64()
#define PHASE_NEEDLE(i) do { const unsigned long long t_ = clock64(); if (threadIdx.x == 0 && A.phase_clocks) \
    atomicAdd(&A.phase_clocks[blockIdx.x * 16 + (i)], t_ - pn_last); pn_last = t_; } while (0)
#define PHASE_UNIT(v) do { const unsigned long long m_ = __ballot(group_live(v)); if (m_) { ++ph_units; ph_lanes += __popcll(m_); } } while (0)
#define PHASE_MARK(i) do { const unsigned long long t_ = clock64(); ph_acc[i] += t_ - ph_last; ph_last = t_; } while (0)
#define PHASE_FLUSH(A) do { if (threadIdx.x == 0 && (A).phase_clocks) { \
    for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&(A).phase_clocks[blockIdx.x * 16 + i_], ph_acc[i_]); \
    atomicAdd(&(A).phase_clocks[blockIdx.x * 16 + 8], ph_units); atomicAdd(&(A).phase_clocks[blockIdx.x * 16 + 9], ph_lanes); } } while (0)
// sweep_role's manager wave (the workgroup's last): its phases go to row kPhaseManagerRow + blockIdx.x
constexpr uint32_t kPhaseManagerRow = 4096;
#define PHASE_FLUSH_MANAGER(A) do { if ((threadIdx.x & 63u) == 0 && (A).phase_clocks) { \
    for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&(A).phase_clocks[(kPhaseManagerRow + blockIdx.x) * 16 + i_], ph_acc[i_]); } } while (0)
#else
#define PHASE_DECL
#define PHASE_NEEDLE_DECL
#define PHASE_NEEDLE(i)
#define PHASE_MARK(i)
#define PHASE_UNIT(v)
#define PHASE_FLUSH(A)
#define PHASE_FLUSH_MANAGER(A)
#endif
constexpr uint64_t kKeyInf    = ~0ull;
constexpr uint32_t kPadPair   = uint32_t(kPadRank) | (uint32_t(kPadRank) << 16);   // two padding sentinels

// --------------------------------------------------------------- tokeniser ---

__device__ __forceinline__ uint32_t dev_symbol(unsigned char c) {
  return (c >= 'a' && c <= 'z') ? uint32_t(c - 'a' + 1) : 0u;   // tokeniser.c:21-31
}

__global__ void tokenise_kernel(const char* __restrict__ packed, const uint64_t* __restrict__ offsets,
                                uint32_t n, const uint32_t* __restrict__ code_total,
                                uint16_t* __restrict__ qcodes, uint32_t* __restrict__ q_ntri,
                                uint32_t* __restrict__ q_nb, uint32_t* __restrict__ big_list,
                                uint32_t* __restrict__ big_count, uint32_t* __restrict__ mid_list,
                                uint32_t* __restrict__ mid_count, const uint32_t* __restrict__ start_win,
                                uint32_t* __restrict__ q_start) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const uint64_t beg = offsets[q], end = offsets[q + 1];
  const char* s = packed + beg;
  uint16_t* out = qcodes + beg + q;              // (end-beg)+1 slots reserved per needle
  uint64_t len = 0;
  const uint64_t cap = end - beg;
  while (len < cap && s[len] != 0) ++len;        // a needle is a C string (storage.c:480)

  // frame "**" + s + "*" and encode (tokeniser.c:62-75); sort ascending (tokeniser.c:93).
  // Needles of up to 64 codes (nearly all) are sorted in the lane's private LDS row -- an
  // insertion sort straight in global memory is a chain of dependent round trips.
  __shared__ uint16_t s_row[128][66];                     // 66: odd word stride, no bank conflicts
  uint32_t a = 0, b = 0;
  const uint64_t m = len + 1;
  uint16_t* work = (m <= 64) ? s_row[threadIdx.x] : out;
  for (uint64_t k = 0; k < m; ++k) {
    const uint32_t c = (k < len) ? dev_symbol((unsigned char)s[k]) : 0u;
    work[k] = uint16_t(a + 28u * b + 784u * c);
    a = b; b = c;
  }
  if (m <= 96) {
    for (uint64_t i = 1; i < m; ++i) {
      const uint16_t v = work[i];
      uint64_t j = i;
      while (j > 0 && work[j - 1] > v) { work[j] = work[j - 1]; --j; }
      work[j] = v;
    }
  } else {
    // heap sort in place for very long needles
    auto sift = [&](uint64_t root, uint64_t lim) {
      for (;;) {
        uint64_t child = 2 * root + 1;
        if (child >= lim) return;
        if (child + 1 < lim && work[child] < work[child + 1]) ++child;
        if (work[root] >= work[child]) return;
        const uint16_t t = work[root]; work[root] = work[child]; work[child] = t;
        root = child;
      }
    };
    for (uint64_t i = m / 2; i-- > 0;) sift(i, m);
    for (uint64_t lim = m; lim-- > 1;) {
      const uint16_t t = work[0]; work[0] = work[lim]; work[lim] = t;
      sift(0, lim);
    }
  }
  // drop duplicates (tokeniser.c:96-107), sum bucket sizes (storage.c:498-502)
  uint32_t d = 0;
  uint64_t nb = 0;
  uint16_t last = 0;
  for (uint64_t k = 0; k < m; ++k) {
    const uint16_t v = work[k];
    if (d == 0 || last != v) { out[d++] = v; last = v; }
  }
  // (bucket sizes in a loop of their own, four loads in flight: a single needle waits on nothing else)
  uint32_t k4 = 0;
  for (; k4 + 4 <= d; k4 += 4) {
    const uint32_t c0 = code_total[out[k4]], c1 = code_total[out[k4 + 1]], c2 = code_total[out[k4 + 2]],
                   c3 = code_total[out[k4 + 3]];
    nb += uint64_t(c0) + c1 + c2 + c3;
  }
  for (; k4 < d; ++k4) nb += code_total[out[k4]];
  q_ntri[q] = d;
  // where references as long as the needle live (the sweep's first window; one or two length classes earlier or
  // later measured in round 3: +1 ... +4 % each way)
  q_start[q] = start_win[len < 255 ? len : 255];
  q_nb[q] = nb > 0xFFFFFFFFull ? 0xFFFFFFFFu : uint32_t(nb);
  if (d > 127) big_list[atomicAdd(big_count, 1u)] = q;            // 16-bit counters
  else if (d > 64) mid_list[atomicAdd(mid_count, 1u)] = q;        // byte counters, two table slots per lane
}

// The same, one WAVE per needle (needles of at most 63 bytes; small batches, where one lane per
// needle leaves a single lane sorting and chasing bucket sizes alone): lane k encodes position k,
// a bitonic network over the 64 lanes sorts, a ballot drops the duplicates, the bucket sizes are
// loaded one per lane and summed across the wave.
__global__ __launch_bounds__(64) void tokenise_wave_kernel(const char* __restrict__ packed,
                                                           const uint64_t* __restrict__ offsets, uint32_t n,
                                                           const uint32_t* __restrict__ code_total,
                                                           uint16_t* __restrict__ qcodes, uint32_t* __restrict__ q_ntri,
                                                           uint32_t* __restrict__ q_nb,
                                                           const uint32_t* __restrict__ start_win,
                                                           uint32_t* __restrict__ q_start) {
  const uint32_t q = blockIdx.x, lane = threadIdx.x;
  if (q >= n) return;
  const uint64_t beg = offsets[q];
  const uint32_t cap = uint32_t(offsets[q + 1] - beg);                   // <= 63 (the launcher checked)
  const unsigned char ch = lane < cap ? static_cast<unsigned char>(packed[beg + lane]) : 0;
  const unsigned long long nul = __ballot(lane < cap && ch == 0);
  const uint32_t len = nul ? uint32_t(__builtin_ctzll(nul)) : cap;       // a needle is a C string (storage.c:480)
  const uint32_t m = len + 1;                                            // positions of "**" + s + "*" (tokeniser.c:62-75)
  const uint32_t c = lane < len ? dev_symbol(ch) : 0u;
  const uint32_t b1 = __shfl_up(c, 1), a2 = __shfl_up(c, 2);
  const uint32_t code = (lane >= 2 ? a2 : 0u) + 28u * (lane >= 1 ? b1 : 0u) + 784u * c;
  uint32_t key = lane < m ? code : 0xFFFFu;                              // sentinel above every code
  // bitonic sort, ascending over the lanes (tokeniser.c:93)
#pragma unroll
  for (uint32_t size = 2; size <= 64; size <<= 1) {
#pragma unroll
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      const uint32_t other = __shfl_xor(key, int(stride));
      const bool up = (lane & size) == 0;                                // direction of this lane's block
      const bool low = (lane & stride) == 0;                             // lower lane of the pair
      key = (low == up) ? min(key, other) : max(key, other);
    }
  }
  // drop duplicates (tokeniser.c:96-107), sum bucket sizes (storage.c:498-502)
  const uint32_t prev = __shfl_up(key, 1);
  const bool uniq = lane < m && (lane == 0 || key != prev);
  const unsigned long long keep = __ballot(uniq);
  uint16_t* out = qcodes + beg + q;
  if (uniq) out[__popcll(keep & ((1ull << lane) - 1))] = uint16_t(key);
  unsigned long long nb = uniq ? code_total[key] : 0u;
#pragma unroll
  for (int d = 32; d; d >>= 1) {
    const uint32_t lo = __shfl_xor(uint32_t(nb), d), hi = __shfl_xor(uint32_t(nb >> 32), d);
    nb += (static_cast<unsigned long long>(hi) << 32) | lo;
  }
  if (lane == 0) {
    q_ntri[q] = uint32_t(__popcll(keep));
    q_start[q] = start_win[len];                                         // len <= 63
    q_nb[q] = nb > 0xFFFFFFFFull ? 0xFFFFFFFFu : uint32_t(nb);
  }
}

// ------------------------------------------------------------ normaliser ---
// Blurrily::Map#normalize_string (lib/blurrily/map.rb:40-47) for ASCII needles, one lane per
// needle, byte for byte what the Ruby does:
//   downcase; then, UNLESS some line of the needle is made of [a-z ] only (map.rb:42 -- Ruby's
//   ^ and $ anchor at lines): every byte that is not a-z becomes ' ' (for ASCII input the NFKD
//   step and the "delete non-ASCII" step are the identity); then runs of whitespace
//   [ \t\n\v\f\r] are squeezed to one space and both ends stripped (trailing NULs too, as
//   String#strip does).
// The result is written at the needle's own offset -- never longer than the input,
// NUL-terminated when shorter, which is where the tokeniser stops -- so `out` may alias `in`.
// Needles holding a byte >= 0x80 are flagged: their NFKD decomposition is host work
// (ActiveSupport's tables, Gemfile.lock:11).
__device__ __forceinline__ bool dev_is_space(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

__global__ void normalise_kernel(const char* in, const uint64_t* __restrict__ offsets, uint32_t n, char* out,
                                 uint32_t* __restrict__ non_ascii) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const uint64_t beg = offsets[q], end = offsets[q + 1];
  // pass 1: is there a line of [a-z ]+ (after downcasing)?  any byte >= 0x80?
  bool plain = false, line_ok = true;
  uint64_t line_len = 0;
  uint32_t high = 0;
  for (uint64_t k = beg; k < end; ++k) {
    unsigned char c = static_cast<unsigned char>(in[k]);
    high |= c >> 7;
    if (c == '\n') {
      plain |= line_ok && line_len > 0;
      line_ok = true; line_len = 0;
      continue;
    }
    if (c >= 'A' && c <= 'Z') c += 'a' - 'A';
    line_ok &= (c >= 'a' && c <= 'z') || c == ' ';
    ++line_len;
  }
  plain |= line_ok && line_len > 0;
  // pass 2: rewrite
  uint64_t w = beg, keep = beg;        // keep: end of the text once trailing blanks / NULs are stripped
  bool gap = false;                    // whitespace seen since the last byte written
  for (uint64_t k = beg; k < end; ++k) {
    unsigned char c = static_cast<unsigned char>(in[k]);
    if (c >= 'A' && c <= 'Z') c += 'a' - 'A';
    const bool letter = c >= 'a' && c <= 'z';
    if (!plain && !letter) c = ' ';
    if (dev_is_space(c)) { gap = true; continue; }
    if (gap && w > beg) out[w++] = ' ';
    gap = false;
    out[w++] = static_cast<char>(c);
    if (c != 0) keep = w;
  }
  if (keep < end) out[keep] = 0;
  if (non_ascii) non_ascii[q] = high;
}

// ------------------------------------------------------------- find kernel ---

// Packed LDS counters.  CT = uint8_t (needles with <= 127 distinct trigrams:
// every real-world needle) or uint16_t (anything longer; the code space has
// 19 683 reachable codes so 15 bits always suffice).
template <typename CT> struct Packing;
template <> struct Packing<uint8_t> {
  static constexpr uint32_t kPerWord = 4, kBits = 8, kHi = 0x80808080u, kOnes = 0x01010101u,
                            kTop = 0x80u, kMask = 0xFFu;
};
template <> struct Packing<uint16_t> {
  static constexpr uint32_t kPerWord = 2, kBits = 16, kLog = 1, kHi = 0x80008000u, kOnes = 0x00010001u,
                            kTop = 0x8000u, kMask = 0xFFFFu;
};

// 4-bit counters for needles with at most 15 distinct trigrams (two needles in three at
// Geonames scale): the same 64 KiB of LDS then hold TWO windows, so a sweep takes half the steps.
// In-window rank r keeps its byte-counter address (word r >> 2, byte 3 - (r & 3)); the even window
// of the step counts in the low nibble of that byte, the odd one in the high nibble -- so a
// posting costs the same instructions as with byte counters, the increment being 1 or 16 shifted
// by the byte position.  Both nibbles of the padding slot 0xFFFF sit in the lowest byte of the
// last word, whose other three bytes belong to no reference (kWindowRanks): their overflow harms nobody,
// and the word is cleared every step.
struct Nib {};
template <> struct Packing<Nib> {
  static constexpr uint32_t kPerWord = 8, kBits = 4, kMask = 0xFu;
};

// What the scan needs to know about a packing: which vectors to read, which counters reached
// `need` (one bit per counter, at the top bit of its field), where the padding slots are.
static_assert(kWindowRanks + 16 <= kWindowSize, "the scan relies on the last 16 counter slots holding no reference");
template <typename CT> struct ScanTraits {
  using P = Packing<CT>;
  static constexpr uint32_t kVecs = kWindowSize * sizeof(CT) / 16;
  static constexpr uint32_t kMaxCount = P::kTop - 1;
  struct Need { uint32_t bias, pre; };
  // pre: a counter >= need >= 2^k has one of the bits k.. of its field set -- one AND per vector tells
  // whether any counter can reach `need` at all before the exact test
  static __device__ __forceinline__ Need prepare(uint32_t need) {
    const uint32_t top = 31u - __clz(max(need, 1u));
    return Need{(P::kTop - need) * P::kOnes, P::kOnes * ((P::kMask << top) & P::kMask)};
  }
  static __device__ __forceinline__ bool maybe(uint4 v, Need n) { return ((v.x | v.y | v.z | v.w) & n.pre) != 0; }
  static __device__ __forceinline__ uint32_t hits(uint32_t v, Need n) { return (v + n.bias) & P::kHi; }
  static __device__ __forceinline__ uint32_t any_hit(uint4 v, Need n) {
    return ((v.x + n.bias) | (v.y + n.bias) | (v.z + n.bias) | (v.w + n.bias)) & P::kHi;
  }
  static __device__ __forceinline__ uint32_t nvec(uint32_t wlen) { return (wlen * uint32_t(sizeof(CT)) + 15) / 16; }
  // The vector holding slot 0xFFFF (padding) holds no reference (kWindowRanks leaves the last 16 slots
  // free), so nvec() never reaches it: nothing to mask; clear_unreached_pad clears it every step.
  static __device__ __forceinline__ uint4 mask_pad(uint4 v, uint32_t) { return v; }
  // counter index = kPerWord * word + field; byte counters sit in descending rank order inside their word
  static __device__ __forceinline__ uint32_t rank_of(uint32_t wbase, uint32_t idx) {
    return wbase + (sizeof(CT) == 1 ? idx ^ 3u : idx);
  }
  // which window of the step a counter belongs to (one window per step here)
  static __device__ __forceinline__ uint32_t parity_of(uint32_t) { return 0u; }
  // a short window does not reach the vector holding the padding slot: clear it here
  static __device__ __forceinline__ void clear_unreached_pad(uint4* cnt128, uint32_t nv, uint32_t tid) {
    if (nv < kVecs && tid == 0) reinterpret_cast<uint32_t*>(cnt128)[kVecs * 4 - 1] = 0;
  }
};
template <> struct ScanTraits<Nib> {
  using P = Packing<Nib>;
  static constexpr uint32_t kVecs = kWindowSize / 16;               // the same 64 KiB
  static constexpr uint32_t kMaxCount = 15;
  // counter >= need, with c = counter, lo = c & 7:  need <= 8: c >= 8 or lo + (8 - need) >= 8;
  //                                                 need >  8: c >= 8 and lo + (16 - need) >= 8
  struct Need { uint32_t bias; bool low; uint32_t pre; };
  static __device__ __forceinline__ Need prepare(uint32_t need) {
    const uint32_t top = 31u - __clz(max(need, 1u));
    const uint32_t pre = 0x11111111u * ((0xFu << top) & 0xFu);
    return need <= 8 ? Need{(8 - need) * 0x11111111u, true, pre} : Need{(16 - need) * 0x11111111u, false, pre};
  }
  static __device__ __forceinline__ bool maybe(uint4 v, Need n) { return ((v.x | v.y | v.z | v.w) & n.pre) != 0; }
  static __device__ __forceinline__ uint32_t hits(uint32_t v, Need n) {
    const uint32_t t = (v & 0x77777777u) + n.bias;
    return (n.low ? (t | v) : (t & v)) & 0x88888888u;
  }
  static __device__ __forceinline__ uint32_t any_hit(uint4 v, Need n) {
    return hits(v.x, n) | hits(v.y, n) | hits(v.z, n) | hits(v.w, n);
  }
  // bytes in use: in-window ranks [0, min(wlen, one window)) -- the odd window is never longer than the even one
  static __device__ __forceinline__ uint32_t nvec(uint32_t wlen) { return (min(wlen, kWindowRanks) + 15) / 16; }
  static __device__ __forceinline__ uint4 mask_pad(uint4 v, uint32_t) { return v; }   // (as above: never reached)
  // counter index = 8 * word + nibble; nibble = 2 * (byte in word) + (window parity); byte = 3 - (rank & 3)
  static __device__ __forceinline__ uint32_t rank_of(uint32_t wbase, uint32_t idx) {
    return wbase + (idx & 1u) * kWindowRanks + ((idx >> 1) ^ 3u);
  }
  static __device__ __forceinline__ uint32_t parity_of(uint32_t idx) { return idx & 1u; }
  static __device__ __forceinline__ void clear_unreached_pad(uint4* cnt128, uint32_t nv, uint32_t tid) {
    if (nv < kVecs && tid == 0) reinterpret_cast<uint32_t*>(cnt128)[kVecs * 4 - 1] = 0;
  }
};

// sweep_coop's pending lists (candidates of a step that left slices out of its count, waiting for their bitmap
// words), one per ring slot, and the pool's tail that takes the ones that pass (compact_pool)
constexpr uint32_t kPendMax = 128;
// (the pool's tail of settled candidates: it has to hold the keys a glance lets pass -- select_at() -- beside a step's
// pending list; 128 slots of the 512-entry pool of limits up to 64, 256 of the 1 024-entry pool of limits up to ~150)
__host__ __device__ constexpr uint32_t adm_max(uint32_t pool_cap) { return pool_cap <= 512 ? 128u : 256u; }

// A candidate is one 64-bit key: (T - matches) in the high word, rank in the low word.  Ranks
// follow (weight, reference), so ascending keys are the reference's result order.
struct Control {            // workgroup-shared scalars
  unsigned long long thr;   // admission threshold: the keep-th best key seen (kKeyInf: none yet)
  unsigned long long floor; // keys at or before this one were delivered by earlier passes
  uint32_t pool_n;          // (pool_n, overflow, adm_n, q: ONE 16-byte read, the hot loop's glance behind a scan)
  uint32_t overflow;
  uint32_t adm_n;           // sweep_coop: keys in the pool's tail -- settled candidates, merged in by compact_pool
  uint32_t q;
  uint32_t pend_n[2];       // sweep_coop: candidates of a step waiting to be settled through bitmaps, by ring slot
  uint32_t tally;           // scratch of cold_start_need
};
static_assert(offsetof(Control, pool_n) % 16 == 0, "the glance reads pool_n .. q as one vector");

// One posting = one relaxed LDS atomic (result unused -> ds_add_u32) on the word holding the
// rank's counter.  This is the generic form (16-bit counters); byte and 4-bit counters take the
// packed-dword path below (bump_pair_bytes: 2 + 3 VALU instructions per pair of postings).
template <typename CT>
__device__ __forceinline__ void bump(uint32_t* cnt32, uint32_t r) {
  using P = Packing<CT>;
  __hip_atomic_fetch_add(&cnt32[r >> P::kLog], 1u << ((r & (P::kPerWord - 1)) * P::kBits),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// (rank in the high half of v) << SH, in one instruction
template <uint32_t SH>
__device__ __forceinline__ uint32_t hi_half_shl(uint32_t v) {
  uint32_t out;
  const uint32_t sh = SH;
  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
      : "=v"(out) : "s"(sh), "v"(v));
  return out;
}

// value >> (amount & 31), the AND being the hardware's
__device__ __forceinline__ uint32_t shr_low5(uint32_t value, uint32_t amount) {
  uint32_t out;
  asm("v_lshrrev_b32_e64 %0, %1, %2" : "=v"(out) : "v"(amount), "s"(value));   // (the constant from a scalar register: it costs no VGPR)
  return out;
}

// top >> (8 * (sel & 3)): v_alignbyte_b32 shifts {0, top} right by the bytes its third operand's low TWO bits
// name (gfx950: tools/micro/alignbyte_probe.hip) -- a byte counter's increment straight from the posting
__device__ __forceinline__ uint32_t shr_bytes_low2(uint32_t top, uint32_t sel) {
  uint32_t out;
  asm("v_alignbyte_b32 %0, 0, %1, %2" : "=v"(out) : "s"(top), "v"(sel));
  return out;
}

// byte counters (ONE = 1) and 4-bit counters (ONE = 1: even window of the step, 16: odd window):
// both ranks of a dword.  In-window rank r counts in word r >> 2, byte 3 - (r & 3) -- the bytes of a word in
// DESCENDING rank order, so that the increment is the constant ONE << 24 moved down by (r & 3) bytes.
template <uint32_t ONE>
__device__ __forceinline__ void bump_pair_bytes(uint32_t* cnt32, uint32_t v) {
  static_assert(kWindowBits == 16, "the packed-dword arithmetic below is written for 16-bit in-window ranks");
  constexpr uint32_t kTop = ONE << 24;
  // low half: address and increment are one instruction each (the shift by bytes reads bits 1:0 of v itself)
  __hip_atomic_fetch_add(&cnt32[(v & 0xFFFFu) >> 2], shr_bytes_low2(kTop, v), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
  // high half: (rank << 3) by SDWA operand select, then the shift whose amount the hardware masks to five
  // bits, which is exactly (rank & 3) * 8 -- written in C++ the amount needs an AND that the compiler cannot
  // drop behind the opaque SDWA result.  3 + 2 VALU instructions per pair of postings.
  __hip_atomic_fetch_add(&cnt32[v >> 18], shr_low5(kTop, hi_half_shl<3>(v)), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Count one 16-byte group: eight 16-bit in-window ranks, every one of them either a real
// rank or the padding sentinel 0xFFFF (whose counter slot the scan ignores) -- no range
// checks.  A lane that had nothing to load holds eight sentinels and skips the group; a loaded
// group always has a real rank in its first slot (device_index.hip arranges that).
__device__ __forceinline__ bool group_live(const uint4 v) { return (v.x & 0xFFFFu) != kPadRank; }

template <typename CT>
__device__ __forceinline__ void bump8(uint32_t* cnt32, const uint4 v) {
  if (!group_live(v)) return;
  if constexpr (sizeof(CT) == 1) {
    bump_pair_bytes<1>(cnt32, v.x); bump_pair_bytes<1>(cnt32, v.y);
    bump_pair_bytes<1>(cnt32, v.z); bump_pair_bytes<1>(cnt32, v.w);
  } else {
    bump<CT>(cnt32, v.x & 0xFFFFu); bump<CT>(cnt32, v.x >> 16);
    bump<CT>(cnt32, v.y & 0xFFFFu); bump<CT>(cnt32, v.y >> 16);
    bump<CT>(cnt32, v.z & 0xFFFFu); bump<CT>(cnt32, v.z >> 16);
    bump<CT>(cnt32, v.w & 0xFFFFu); bump<CT>(cnt32, v.w >> 16);
  }
}

// 4-bit flavour: HALF = parity of the unit's window within the step
template <uint32_t HALF>
__device__ __forceinline__ void bump8_nib(uint32_t* cnt32, const uint4 v) {
  if (!group_live(v)) return;
  constexpr uint32_t kOne = HALF ? 16u : 1u;
  bump_pair_bytes<kOne>(cnt32, v.x); bump_pair_bytes<kOne>(cnt32, v.y);
  bump_pair_bytes<kOne>(cnt32, v.z); bump_pair_bytes<kOne>(cnt32, v.w);
}
// count one unit; `half` (uniform) = parity of its window, used by the 4-bit flavour only.
// (Reading the short last unit of a slice with four or two ranks per lane -- fewer, fuller
// atomic instructions -- was measured: 17% slower.  The LDS atomics cost by lane and by bank
// conflict, not by instruction, and the narrow reads undo the bank-aware dealing.)
// the same for a lane KNOWN to hold a loaded group (sweep_coop carries the load's lane predicate instead of
// filling idle lanes with sentinels): no liveness test.  (A wave's atomics in flight bounded -- s_waitcnt lgkmcnt(2) or
// (4) behind every pair -- measured in round 3: -0.3 %, noise; every pair waited for: +1.2 % only: a wave's own
// atomics are not what its next LDS read queues behind.)
template <typename CT>
__device__ __forceinline__ void bump_unit_loaded(uint32_t* cnt32, const uint4 v, uint32_t half) {
  static_assert(sizeof(CT) == 1 || std::is_same<CT, Nib>::value, "byte and 4-bit counters only");
  if constexpr (std::is_same<CT, Nib>::value) {
    if (half) {
      bump_pair_bytes<16>(cnt32, v.x); bump_pair_bytes<16>(cnt32, v.y);
      bump_pair_bytes<16>(cnt32, v.z); bump_pair_bytes<16>(cnt32, v.w);
    } else {
      bump_pair_bytes<1>(cnt32, v.x); bump_pair_bytes<1>(cnt32, v.y);
      bump_pair_bytes<1>(cnt32, v.z); bump_pair_bytes<1>(cnt32, v.w);
    }
  } else {
    (void)half;
    bump_pair_bytes<1>(cnt32, v.x); bump_pair_bytes<1>(cnt32, v.y);
    bump_pair_bytes<1>(cnt32, v.z); bump_pair_bytes<1>(cnt32, v.w);
  }
}

template <typename CT>
__device__ __forceinline__ void bump_unit(uint32_t* cnt32, const uint4 v, uint32_t half) {
  if constexpr (std::is_same<CT, Nib>::value) {
    if (half) bump8_nib<1>(cnt32, v); else bump8_nib<0>(cnt32, v);
  } else {
    (void)half;
    bump8<CT>(cnt32, v);
  }
}

// request counters (FindArgs::stats) for the sweeps that load per lane: postings of one wave-load
__device__ __forceinline__ void stat_unit(unsigned long long* stats, const uint4 v) {
  if (!stats) return;
  const unsigned long long m = __ballot((v.x & 0xFFFFu) != kPadRank);
  if (m && (threadIdx.x & 63) == 0) atomicAdd(&stats[kStatPostingEntries], 8ull * __popcll(m));
}

__device__ __forceinline__ uint4 load_group(const uint16_t* ent, uint32_t c, uint32_t b) {
  uint4 v = make_uint4(kPadPair, kPadPair, kPadPair, kPadPair);
  if (c < b) v = *reinterpret_cast<const uint4*>(ent + c);
  return v;
}

// Sort the candidate pool ascending, keep the best `keep`, and tighten the admission
// threshold.  Called by all threads of the workgroup.
// (`scan_cap`: the slots the scans fill -- all of them, or all but the last kAdmMax where sweep_coop leaves slices out)
template <int NT>
__device__ void compact_pool(unsigned long long* pool, Control* ctl, uint32_t cap, uint32_t keep, uint32_t scan_cap = 0) {
  if (scan_cap == 0) scan_cap = cap;
  __builtin_amdgcn_s_setprio(kSerialPrio);            // barriers and LDS round trips, nothing to overlap inside the workgroup
  const uint32_t tid = threadIdx.x;
  // sweep_coop: candidates settled through bitmaps wait in the pool's last kAdmMax slots (the scans then fill only
  // the slots in front of them): moved up behind the scans' keys first
  const uint32_t an = min(ctl->adm_n, cap - scan_cap);
  uint32_t n = min(ctl->pool_n, scan_cap);
  if (an) {                                           // (uniform)
    const unsigned long long key = tid < an ? pool[scan_cap + tid] : 0ull;
    __syncthreads();
    if (tid < an) pool[n + tid] = key;
    if (tid == 0) ctl->adm_n = 0;
    __syncthreads();
    n += an;
  }
  if (n <= kRankSortMax) {
    // Small pools (most of them): every key counts the keys below it -- the keys are distinct,
    // a rank being harvested once -- and the best `keep` go straight to their places: three
    // barriers instead of the bitonic network's dozens.
    const unsigned long long mine = tid < n ? pool[tid] : kKeyInf;
    uint32_t below = 0;
    if (tid < n) {                                                   // (same address in every lane: a broadcast)
      const uint32_t n2 = n & ~1u;
      for (uint32_t j = 0; j < n2; j += 2) {                         // two keys per read
        const ulonglong2 two = *reinterpret_cast<const ulonglong2*>(pool + j);
        below += uint32_t(two.x < mine) + uint32_t(two.y < mine);
      }
      if (n2 < n) below += pool[n2] < mine;
    }
    __syncthreads();
    if (tid < n && below < keep) pool[below] = mine;
    __syncthreads();
    if (tid == 0) {
      ctl->pool_n = min(n, keep);
      ctl->overflow = 0;
      if (n >= keep && keep > 0) ctl->thr = pool[keep - 1];
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);
    return;
  }
  uint32_t P = 1;
  while (P < n) P <<= 1;
  for (uint32_t i = n + tid; i < P; i += NT) pool[i] = kKeyInf;
  __syncthreads();
  for (uint32_t size = 2; size <= P; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t i = tid; i < (P >> 1); i += NT) {
        const uint32_t lo = 2 * i - (i & (stride - 1));
        const uint32_t hi = lo + stride;
        const bool asc = (lo & size) == 0;
        const unsigned long long x = pool[lo], y = pool[hi];
        if ((x > y) == asc) { pool[lo] = y; pool[hi] = x; }
      }
      __syncthreads();
    }
  }
  if (tid == 0) {
    ctl->pool_n = min(n, keep);
    ctl->overflow = 0;
    if (n >= keep && keep > 0) ctl->thr = pool[keep - 1];
  }
  __syncthreads();
  __builtin_amdgcn_s_setprio(0);
}

// State of one needle's sweep that every phase needs.
struct Needle {
  uint32_t q;                       // its number in the batch (path flags of the counted build only)
  uint32_t T;                       // distinct trigrams
  bool has_floor;                   // later pass of a limit larger than the pool (floor key in Control)
};

// A counter of the window starting at rank `wbase` must reach this many matches to beat the
// current keep-th candidate `thr`: its match count, or one more when the whole window lies
// behind that candidate's rank (a tie loses to the lower rank).  Valid for any sweep order.
__device__ __forceinline__ uint32_t matches_needed(unsigned long long thr, uint32_t T, uint32_t wbase) {
  if (thr == kKeyInf) return 1u;
  const uint32_t matches_k = T - uint32_t(thr >> 32);
  return max(1u, uint32_t(thr) >= wbase ? matches_k : matches_k + 1);
}

// ---- scan: admit counters that can still reach the top `keep`, clear them ----------------
// `need`: counters below it cannot enter the pool (the caller's bound: from the threshold, or the cold start's);
// `cap`: no counter of this sweep exceeds it.  The threshold itself (*thr_p, in LDS: it does not change while a
// scan runs) is read only where a counter is harvested.
// `ls`: slices LEFT OUT of this step's count (sweep_coop: L of the even window in bits 3:0, of the odd one in bits
// 7:4; 0: none) -- a harvested counter of such a window holds the matches among the counted slices only, `need` was
// lowered by the publishing wave accordingly, and the candidate does NOT enter the pool here: if its best case
// (every left-out slice a match) beats the threshold it goes to the step's PENDING list as in-window rank | parity
// << 16 | counted matches << 20, and sweep_coop's manager wave settles it through those slices' bitmaps during the next
// step.  `stride`: the threads that scan (sweep_coop's worker waves: NT - 64).
template <typename CT, int NT>
__device__ __forceinline__ void scan_core(uint4* cnt128, const Needle& nd, const uint32_t need, const uint32_t cap,
                                          const unsigned long long* thr_p, const unsigned long long* floor,
                                          const uint32_t* tomb, unsigned long long* pool, const uint32_t pool_cap,
                                          uint32_t* pool_n, uint32_t* overflow, uint32_t wbase, uint32_t wlen,
                                          uint32_t* path_flag = nullptr, const uint32_t ls = 0, uint32_t* pend = nullptr,
                                          uint32_t* pend_n = nullptr, const uint32_t pend_cap = 0, const uint32_t stride = NT) {
  using P = Packing<CT>;
  using S = ScanTraits<CT>;
  const uint32_t tid = threadIdx.x;
  const uint32_t nvec = S::nvec(wlen);
  __builtin_amdgcn_s_setprio(1);                         // the scan is a chain of LDS round trips: it goes ahead of the other workgroup's counting
  // the zero quad the scanned vectors are cleared with, materialised once per scan (opaque: a plain zero vector
  // would be hoisted out of the sweep and held -- or spilled -- for the whole needle)
  uint4 zq;
  asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0"
               : "=v"(zq.x), "=v"(zq.y), "=v"(zq.z), "=v"(zq.w));
  if (need <= cap) {                                     // (a 4-bit window holds no counter above 15)
    const typename S::Need nq = S::prepare(need);
    // slow path of one vector: some counter reached `need`
    auto harvest = [&](const uint4 v, const uint32_t i) {
      const unsigned long long thr = *thr_p;
      auto word = [&](const uint32_t wv, const uint32_t j) {
        uint32_t m = S::hits(wv, nq);
        while (m) {
          const uint32_t bit = __ffs(m) - 1;
          m &= m - 1;
          const uint32_t pos = bit / P::kBits;
          const uint32_t cnt = (wv >> (pos * P::kBits)) & P::kMask;
          const uint32_t idx = (i * 4 + j) * P::kPerWord + pos;
          const uint32_t rank = S::rank_of(wbase, idx);
          if (ls) {
            const uint32_t h = S::parity_of(idx);
            const uint32_t L = (ls >> (4u * h)) & 15u;
            if (L) {                                   // slices of this window were left out of the count
              const unsigned long long best = (static_cast<unsigned long long>(nd.T - min(nd.T, cnt + L)) << 32) | rank;
              bool maybe = best <= thr;
              if (tomb) maybe = maybe && ((tomb[rank >> 5] >> (rank & 31)) & 1u) == 0;
              if (maybe) {
                const uint32_t at = atomicAdd(pend_n, 1u);
                if (at < pend_cap) pend[at] = (rank - wbase - h * kWindowRanks) | (h << 16) | (cnt << 20);
                else *overflow = 1;                    // (the step is swept again, every slice counted)
                if (path_flag) atomicOr(path_flag, kPathNmLeftOut);
              }
              continue;
            }
          }
          const unsigned long long key = (static_cast<unsigned long long>(nd.T - cnt) << 32) | rank;
          bool pass = key <= thr;
          if (nd.has_floor) pass = pass && key > *floor;
          if (path_flag && tomb && pass && ((tomb[rank >> 5] >> (rank & 31)) & 1u) != 0) atomicOr(path_flag, kPathTombstone);
          if (tomb) pass = pass && ((tomb[rank >> 5] >> (rank & 31)) & 1u) == 0;   // deleted since the build
          if (pass) {
            const uint32_t at = atomicAdd(pool_n, 1u);
            if (at < pool_cap) pool[at] = key;
            else *overflow = 1;
          }
        }
      };
      word(v.x, 0); word(v.y, 1); word(v.z, 2); word(v.w, 3);
    };
    // (Measured alternatives: a thread's vectors two at a time -- reads in flight together, one zero quad -- 3.5 %
    // slower in round 1; all four of a thread in flight, round 3: 7.5 % slower (330.2 vs 307.4 ms per 500 k needles,
    // same box): the LDS pipe is what the two resident workgroups share, a burst of reads lengthens the queue the
    // other one's atomics wait in; read-and-clear in one ds_wrxchg_rtn_b64 per 8 bytes: +0.5 %, noise.)
    for (uint32_t i = tid; i < nvec; i += stride) {
      uint4 v = cnt128[i];
      cnt128[i] = zq;
      v = S::mask_pad(v, i);
      // one AND per vector, then one SWAR test: the top bit of a field is set iff its counter >= need
      if (S::maybe(v, nq) && S::any_hit(v, nq)) {
        __builtin_amdgcn_s_setprio(3);     // a wave that found something is the one the scan barrier will wait for
        harvest(v, i);
      }
    }
  } else {
    // nothing in this window can enter the pool any more: just clear the counters
    for (uint32_t i = tid; i < nvec; i += stride) cnt128[i] = zq;
  }
  S::clear_unreached_pad(cnt128, nvec, tid);
  __builtin_amdgcn_s_setprio(0);
}

// Cold start: with no threshold yet, every non-zero counter of a window would flood the pool
// (overflow, sort, sweep again -- several times).  Instead find, by bisection over the counter
// value, the largest c such that at least `keep` counters of THIS window reach c: those alone
// already fill the answer, so counters below c cannot be part of it and the first scan admits
// only counters >= c.  A pass reads the counters (no clearing) and tallies the SWAR hits.
template <typename CT, int NT>
__device__ __forceinline__ uint32_t cold_start_need(const uint4* cnt128, uint32_t T, uint32_t keep, Control* ctl,
                                                    uint32_t wlen) {
  using S = ScanTraits<CT>;
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t nvec = S::nvec(wlen);
  uint32_t lo = 1, hi = min(T, S::kMaxCount);            // answer in [lo, hi]; lo = 1 means "no restriction"
  __builtin_amdgcn_s_setprio(kSerialPrio);
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (tid == 0) ctl->tally = 0;
    __syncthreads();
    const typename S::Need nq = S::prepare(mid);
    uint32_t mine = 0;
    for (uint32_t i = tid; i < nvec; i += NT) {
      const uint4 v = S::mask_pad(cnt128[i], i);
      mine += __popc(S::hits(v.x, nq)) + __popc(S::hits(v.y, nq)) + __popc(S::hits(v.z, nq)) +
              __popc(S::hits(v.w, nq));
    }
#pragma unroll
    for (uint32_t d = 32; d; d >>= 1) mine += __shfl_xor(mine, int(d));
    if (lane == 0 && mine) atomicAdd(&ctl->tally, mine);
    __syncthreads();
    if (ctl->tally >= keep) lo = mid; else hi = mid - 1;
    __syncthreads();                                     // tally read by everyone before it is reset
  }
  return lo;
}

template <typename CT, int NT>
__device__ __forceinline__ void scan_window(const FindArgs& A, const Needle& nd, uint4* cnt128,
                                            unsigned long long* pool, Control* ctl, uint32_t wbase,
                                            uint32_t wlen, uint32_t pool_cap = 0) {
  if (pool_cap == 0) pool_cap = A.pool_cap;
  const unsigned long long thr = ctl->thr;
  uint32_t need_floor = 0;
  // (not with a floor key or tombstones: candidates they reject would be counted as present)
  if (thr == kKeyInf && !nd.has_floor && !A.tomb && nd.T > 1) {
    need_floor = cold_start_need<CT, NT>(cnt128, nd.T, A.keep, ctl, wlen);
    PATH_FLAG(A, nd.q, kPathColdStart);
  }
  scan_core<CT, NT>(cnt128, nd, max(matches_needed(thr, nd.T, wbase), need_floor), min(nd.T, ScanTraits<CT>::kMaxCount),
                    &ctl->thr, &ctl->floor, A.tomb, pool, pool_cap, &ctl->pool_n, &ctl->overflow, wbase, wlen,
                    STATS(A) && A.path_flags ? &A.path_flags[nd.q] : nullptr);
}

// The pool is compacted -- and the threshold tightened: it moves only there -- when it holds more than this many keys.
// Capacity is for the rare flood (the first window's ties: an overflow costs a second sweep of the step); the
// TRIGGER is what keeps the threshold tight, and with it the bounds the steps scan with and the keys they harvest.
// It sat at half the capacity (256 keys at limit 10) through round 3; at `keep` and a half (16 keys at limit 10)
// configs[2] takes 253 ms per 500 k needles instead of 279.5 -- and 73.5 instead of 93.7 ms per 200 k at limit 1,
// 127.6 instead of 136.9 at limit 100 (tools/limit_probe.py).  12 to 20 keys measure alike at limit 10, 10 (a
// compaction behind every admission) 2 % worse, 32 / 48 keys 1.7 / 3 % worse.  The same through a smaller POOL (64 keys)
// cost configs[1] a hundred times the second sweeps.  Phase 1 of the window-major sweep keeps the old trigger: its one
// step per needle floods by design (configs[4]: 68.7 vs 69.3 ms).
__device__ __forceinline__ uint32_t select_at(const FindArgs& A) {
  return A.own_only ? A.pool_cap / 2 : min(A.keep + max(6u, A.keep / 2), A.pool_cap / 2);
}

// ---- select: keep the pool small and the threshold tight.  Returns true when the pool
// overflowed during the scan of this window, i.e. the window has to be swept again.
template <int NT>
__device__ __forceinline__ bool select_after_scan(const FindArgs& A, unsigned long long* pool, Control* ctl,
                                                  uint32_t wbase, uint32_t wlen, uint32_t q_flag, uint32_t scan_cap = 0) {
  (void)q_flag;
  const uint32_t ov = ctl->overflow;
  const uint32_t pn = ctl->pool_n + ctl->adm_n;       // (settled candidates in the pool's tail count)
  // compact when the pool holds more than select_at() keys -- or as soon as it holds `keep` candidates for the first
  // time, so that a threshold exists from then on
  if (!(ov || pn > select_at(A) || (ctl->thr == kKeyInf && pn >= A.keep))) return false;
  if (STATS(A) && threadIdx.x == 0) atomicAdd(&STATS(A)[kStatCompactions], 1ull);
  compact_pool<NT>(pool, ctl, A.pool_cap, A.keep, scan_cap);
  PATH_FLAG(A, q_flag, ov ? kPathCompaction | kPathResweep : kPathCompaction);
  if (!ov) return false;
  // The pool overflowed mid-window: candidates of this window were lost.  Keep the
  // tightened threshold (the keep-th best of a subset is a valid bound), forget this
  // window's survivors and sweep the window again.
  if (threadIdx.x == 0) {
    uint32_t j = 0;
    const uint32_t n = ctl->pool_n;
    for (uint32_t i = 0; i < n; ++i)
      if (uint32_t(pool[i]) - wbase >= wlen) pool[j++] = pool[i];     // not of this window (any sweep order)
    ctl->pool_n = j;
  }
  __syncthreads();
  return true;
}

// Every slice (already padded to whole 16-byte groups) is cut into units of 64 groups
// (512 postings, 1 KiB per wave-load).  Unit j of slice t belongs to wave (t + j) mod kNW: one
// hot trigram is streamed by the whole workgroup and a needle's many small slices spread over
// the waves.
__device__ __forceinline__ uint32_t slice_units(uint32_t a, uint32_t b) { return (((b - a) >> 3) + 63) >> 6; }

// ---- LDS-table flavour (needles with more than 128 distinct trigrams) ---------------------
// Streams every unit of this wave for the slices staged in s_a/s_b.  Returns true if the
// window holds any posting of those slices.
template <typename CT, int kNW>
__device__ __forceinline__ bool count_window(const FindArgs& A, uint32_t* cnt32, const uint32_t* s_a,
                                             const uint32_t* s_b, uint32_t tc, uint32_t wid, uint32_t lane) {
  bool any = false;
  for (uint32_t t = 0; t < tc; ++t) {
    const uint32_t a = __builtin_amdgcn_readfirstlane(s_a[t]);
    const uint32_t b = __builtin_amdgcn_readfirstlane(s_b[t]);
    if (a == b) continue;
    any = true;
    const uint32_t su = slice_units(a, b);
    for (uint32_t j = (wid - t) & (kNW - 1); j < su; j += kNW) {
      const uint4 v = load_group(A.ent, a + (j * 64 + lane) * 8, b);
      stat_unit(STATS(A), v);
      bump8<CT>(cnt32, v);
    }
  }
  return any;
}

// Sweep for needles whose trigrams need several staging chunks (> 128 distinct trigrams).
template <typename CT, int NT>
__device__ void sweep_chunked(const FindArgs& A, const Needle& nd, const uint16_t* codes, uint32_t* cnt32,
                              unsigned long long* pool, uint32_t* s_tab, Control* ctl, const uint32_t w0,
                              const uint32_t w1) {
  constexpr uint32_t kNW = NT / 64;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  uint32_t* s_a = s_tab;
  uint32_t* s_b = s_tab + kCodeChunk;
  PATH_FLAG(A, nd.q, kPathChunked);
  for (uint32_t w = w0; w < w1; ++w) {
    const uint32_t wbase = w * kWindowRanks;
    const uint32_t wlen = min(kWindowRanks, A.n_refs - wbase);
    const uint2* soff = A.slice_se + size_t(w) * kNumCodes;
    bool redo;
    do {
      redo = false;
      bool touched = false;
      for (uint32_t c0 = 0; c0 < nd.T; c0 += kCodeChunk) {
        const uint32_t tc = min(kCodeChunk, nd.T - c0);
        if (tid < tc) {
          const uint32_t code = codes[c0 + tid];
          const uint2 se_ = soff[code];
          s_a[tid] = se_.x; s_b[tid] = se_.y;
        }
        __syncthreads();
        touched |= count_window<CT, kNW>(A, cnt32, s_a, s_b, tc, wid, lane);
        __syncthreads();                                        // counts visible; s_a/s_b reusable
      }
      if (!touched) break;                                      // nothing of this needle in the window
      scan_window<CT, NT>(A, nd, reinterpret_cast<uint4*>(cnt32), pool, ctl, wbase, wlen);
      __syncthreads();
      redo = select_after_scan<NT>(A, pool, ctl, wbase, wlen, nd.q);
    } while (redo);
  }
}

// ---- register-table flavour (needles with <= 128 distinct trigrams: every real needle) -----
// A needle's slice table for one window, held in registers: lane t owns slice t (slot 0) and
// slice t+64 (slot 1); every wave keeps its own copy, so walking the table needs neither LDS
// nor a barrier, and a wave finds its units lane-parallel (one ballot per slot): the cost
// follows the units it owns, not T.
// The units of this wave inside one slot of the table (a, b: the slot's per-lane slice bounds).
// UNIT_BODY sees: k (ordinal of the unit within this wave), c (entry index of the lane's
// 16-byte group), sb (end of the slice).  Plain macro rather than a callback: closures that
// capture registers by reference end up in scratch memory.
#define BLURRILY_FOR_SLOT_UNITS(kNW, a, b, wid, lane, k, UNIT_BODY)                              \
  do {                                                                                           \
    unsigned long long mask_ = __ballot((((wid) - (lane)) & ((kNW) - 1)) < slice_units((a), (b))); \
    while (mask_) {                                                                              \
      const uint32_t t_ = __builtin_ctzll(mask_);                                                \
      mask_ &= mask_ - 1;                                                                        \
      const uint32_t sa = __builtin_amdgcn_readlane((a), t_);                                    \
      const uint32_t sb = __builtin_amdgcn_readlane((b), t_);                                    \
      const uint32_t su_ = slice_units(sa, sb);                                                  \
      for (uint32_t j_ = ((wid) - t_) & ((kNW) - 1); j_ < su_; j_ += (kNW)) {                     \
        const uint32_t c = sa + (j_ * 64 + (lane)) * 8;                                          \
        UNIT_BODY;                                                                               \
        ++(k);                                                                                   \
      }                                                                                          \
    }                                                                                            \
  } while (0)

// Head of a window: the first three units of this wave (c == b == 0: no such unit).
// `more` = the wave owns further units; returns true if the window holds any posting.
template <int kNW>
__device__ __forceinline__ bool head_units(uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, bool two_slots,
                                           uint32_t wid, uint32_t lane, bool& more, uint32_t& c0, uint32_t& e0,
                                           uint32_t& c1, uint32_t& e1, uint32_t& c2, uint32_t& e2) {
  uint32_t k = 0, x0 = 0, y0 = 0, x1 = 0, y1 = 0, x2 = 0, y2 = 0;
#define BLURRILY_KEEP_HEAD                                   \
  {                                                          \
    x0 = k == 0 ? c : x0; y0 = k == 0 ? sb : y0;             \
    x1 = k == 1 ? c : x1; y1 = k == 1 ? sb : y1;             \
    x2 = k == 2 ? c : x2; y2 = k == 2 ? sb : y2;             \
  }
  BLURRILY_FOR_SLOT_UNITS(kNW, a0, b0, wid, lane, k, BLURRILY_KEEP_HEAD);
  if (two_slots) BLURRILY_FOR_SLOT_UNITS(kNW, a1, b1, wid, lane, k, BLURRILY_KEEP_HEAD);
#undef BLURRILY_KEEP_HEAD
  c0 = x0; e0 = y0; c1 = x1; e1 = y1; c2 = x2; e2 = y2;
  more = k > 3u;
  return (__ballot(b0 > a0) | __ballot(b1 > a1)) != 0;
}

// Units of this wave beyond the first `skip`: loaded and counted in place, one unit's LDS
// atomics running while the next unit's load is in flight.
template <typename CT, int kNW>
__device__ __forceinline__ void count_rest(const uint16_t* ent, uint32_t* cnt32, uint32_t a0, uint32_t b0,
                                           uint32_t a1, uint32_t b1, bool two_slots, uint32_t wid,
                                           uint32_t lane, uint32_t skip, unsigned long long* stats) {
  uint32_t k = 0;
  uint4 pend = make_uint4(kPadPair, kPadPair, kPadPair, kPadPair);
  BLURRILY_FOR_SLOT_UNITS(kNW, a0, b0, wid, lane, k, {
    if (k >= skip) { const uint4 v = load_group(ent, c, sb); stat_unit(stats, v); bump8<CT>(cnt32, pend); pend = v; }
  });
  if (two_slots)
    BLURRILY_FOR_SLOT_UNITS(kNW, a1, b1, wid, lane, k, {
      if (k >= skip) { const uint4 v = load_group(ent, c, sb); stat_unit(stats, v); bump8<CT>(cnt32, pend); pend = v; }
    });
  bump8<CT>(cnt32, pend);
}

// Software-pipelined sweep: every wave keeps the needle's slice tables of windows w and w+1 in
// registers and, while window w is scanned, already has its first three units of window w+1
// in flight.  A window costs two barriers and, in steady state, no exposed global-memory
// round trip.
// Serves needles with 65..128 distinct trigrams (a second table slot per lane) and the wide-counter launches;
// needles with <= 64 trigrams -- nearly all -- take sweep_coop below.
template <typename CT, int NT>
__device__ void sweep_pipelined(const FindArgs& A, const Needle& nd, const uint16_t* codes, uint32_t* cnt32,
                                unsigned long long* pool, Control* ctl, const uint32_t w0, const uint32_t w1,
                                const uint32_t ws) {
  // Windows [w0, w1) are visited starting at `ws` and wrapping around: the needle's own length
  // class first (its best matches, so the threshold tightens early), the rest after.  A window
  // whose references cannot reach the threshold (win_max_tri) is skipped without being loaded.
  constexpr uint32_t kNW = NT / 64;
  constexpr uint32_t kPre = 3;                                  // units loaded one window ahead
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t tc = nd.T;                                     // <= 128
  const uint32_t nwin = w1;                                     // windows [w0, w1) of the rank space
  const bool two_slots = tc > 64;

  const bool own0 = lane < tc, own1 = lane + 64 < tc;
  const uint32_t code0 = own0 ? codes[lane] : 0u;
  const uint32_t code1 = own1 ? codes[lane + 64] : 0u;
  // slice tables of window w (ca*/cb*) and w+1 (na*/nb*), plain registers
  uint32_t ca0 = 0, cb0 = 0, ca1 = 0, cb1 = 0, na0 = 0, nb0 = 0, na1 = 0, nb1 = 0;
#define BLURRILY_FETCH_TABLE(w_, A0, B0, A1, B1)                                 \
  do {                                                                           \
    A0 = B0 = A1 = B1 = 0;                                                       \
    if ((w_) < nwin) {                                                           \
      const uint2* soff_ = A.slice_se + size_t(w_) * kNumCodes;                  \
      if (own0) { const uint2 se_ = soff_[code0]; A0 = se_.x; B0 = se_.y; }      \
      if (own1) { const uint2 se_ = soff_[code1]; A1 = se_.x; B1 = se_.y; }      \
    }                                                                            \
  } while (0)

  // head of a window: its first kPre units of this wave, loaded ahead of time
  uint4 u0, u1, u2;
  bool head_any = false, head_more = false;                     // of the window the head belongs to
  uint32_t hc0, hb0, hc1, hb1, hc2, hb2;
#define BLURRILY_LOAD_HEAD(A0, B0, A1, B1)                                                          \
  do {                                                                                              \
    head_any = head_units<kNW>(A0, B0, A1, B1, two_slots, wid, lane, head_more, hc0, hb0, hc1,      \
                               hb1, hc2, hb2);                                                      \
    u0 = load_group(A.ent, hc0, hb0);                                                               \
    u1 = load_group(A.ent, hc1, hb1);                                                               \
    u2 = load_group(A.ent, hc2, hb2);                                                               \
  } while (0)

  const uint32_t n_visit = w1 - w0;
  // i-th window of the sweep; past the end: w1 (an empty table)
#define BLURRILY_WIN_AT(i_) ((i_) < n_visit ? (ws + (i_) < w1 ? ws + (i_) : ws + (i_) - n_visit) : w1)
  // the window cannot contain a candidate: no reference of it has enough trigrams
#define BLURRILY_SKIPPABLE(w_) \
  ((w_) < w1 && min(tc, A.win_max_tri[w_]) < matches_needed(ctl->thr, tc, (w_) * kWindowRanks))

  PATH_FLAG(A, nd.q, kPathPipelined);
  BLURRILY_FETCH_TABLE(BLURRILY_WIN_AT(0u), ca0, cb0, ca1, cb1);
  BLURRILY_FETCH_TABLE(BLURRILY_WIN_AT(1u), na0, nb0, na1, nb1);
  BLURRILY_LOAD_HEAD(ca0, cb0, ca1, cb1);

  PHASE_DECL;
  for (uint32_t i = 0; i < n_visit; ++i) {
    const uint32_t w = BLURRILY_WIN_AT(i);
    const uint32_t wbase = w * kWindowRanks;
    const uint32_t wlen = min(kWindowRanks, A.n_refs - wbase);
    const bool any = head_any, more = head_more;
    PHASE_MARK(0);                                              // loop overhead
    if (any) {
      // ---- count window w: its head was loaded one window ago ---------------------------
      PHASE_UNIT(u0); PHASE_UNIT(u1); PHASE_UNIT(u2);
      stat_unit(STATS(A), u0); stat_unit(STATS(A), u1); stat_unit(STATS(A), u2);
      if (STATS(A) && tid == 0) atomicAdd(&STATS(A)[kStatSteps], 1ull);
      bump8<CT>(cnt32, u0);
      bump8<CT>(cnt32, u1);
      bump8<CT>(cnt32, u2);
      PHASE_MARK(1);                                            // head counted
      if (more) count_rest<CT, kNW>(A.ent, cnt32, ca0, cb0, ca1, cb1, two_slots, wid, lane, kPre, STATS(A));
      PHASE_MARK(2);                                            // rest counted
      __syncthreads();                                          // counts visible
      PHASE_MARK(3);                                            // barrier after count
    }
    // keep the memory pipe busy during the scan: head of the next window (all-empty past the
    // end); none at all if that window cannot hold a candidate (uniform: thr only changes
    // behind the barriers of select)
    if (BLURRILY_SKIPPABLE(BLURRILY_WIN_AT(i + 1))) {
      head_any = false; head_more = false;
      PATH_FLAG(A, nd.q, kPathSkipped);
    } else {
      BLURRILY_LOAD_HEAD(na0, nb0, na1, nb1);
    }
    PHASE_MARK(4);                                              // next head issued
    if (any) {
      for (;;) {
        scan_window<CT, NT>(A, nd, reinterpret_cast<uint4*>(cnt32), pool, ctl, wbase, wlen);
        PHASE_MARK(5);                                          // scan
        __syncthreads();                                        // counters are zero again
        PHASE_MARK(6);                                          // barrier after scan
        if (!select_after_scan<NT>(A, pool, ctl, wbase, wlen, nd.q)) break;
        count_rest<CT, kNW>(A.ent, cnt32, ca0, cb0, ca1, cb1, two_slots, wid, lane, 0u, STATS(A));   // overflow: again
        __syncthreads();
      }
      PHASE_MARK(7);                                            // select / compaction
    }
    ca0 = na0; cb0 = nb0; ca1 = na1; cb1 = nb1;
    BLURRILY_FETCH_TABLE(BLURRILY_WIN_AT(i + 2), na0, nb0, na1, nb1);
  }
  PHASE_FLUSH(A);
#undef BLURRILY_SKIPPABLE
#undef BLURRILY_WIN_AT
#undef BLURRILY_LOAD_HEAD
#undef BLURRILY_FETCH_TABLE
}

// ---- cooperative flavour (needles with <= 64 distinct trigrams: nearly all of them) --------
// In sweep_pipelined every wave walks the needle's slice table itself to find its units: sixteen
// times the same work.  Here ONE wave per step (rotating) cuts the slices of the next visited
// window into units and publishes them as descriptors {first entry, slice end} in an LDS ring;
// in the next step every wave just reads the descriptors of its units (unit k belongs to wave
// k mod kNW) and streams them, one unit's LDS atomics running under the next unit's load.
// (Loading units a step ahead, during the scan, was measured and bought nothing: 0, 1 or 2
// units in flight are equal within noise, 3 or 4 cost 7-11% -- the registers are worth more.)
// The next visited window is decided one step ahead (windows that cannot hold a candidate are
// stepped over without a barrier), so the producing wave has the table in registers before it
// needs it.
// Descriptors per ring slot: the dynamic LDS behind the counters is one budget (two workgroups per CU), shared by
// the candidate pool and the ring -- a small pool (limit <= 64) leaves room for 512 units a step, one of 1024
// entries for 384, a larger one for 256.  A step with more units than that is walked by every wave from the table
// itself (BLURRILY_COUNT_WALK).
constexpr uint32_t kRingUnitsMax = 512;
__host__ __device__ constexpr uint32_t ring_units_for(uint32_t pool_cap) {
  return pool_cap <= 512 ? kRingUnitsMax : 256u;                 // (multiples of the sixteen waves; 384 beside the 1 024-entry
                                                                 //  pool made no difference at configs[4], and the pending lists need the room)
}
// Inclusive prefix sum over the 64 lanes of a wave with DPP moves (row shifts inside the rows of 16, then the two
// row broadcasts of gfx9): ten VALU instructions, against six dependent ds_bpermute round trips for __shfl_up.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
#define BLURRILY_DPP_ADD(ctrl_, rows_) x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), ctrl_, rows_, 0xF, false))
  BLURRILY_DPP_ADD(0x111, 0xF);                                 // row_shr:1
  BLURRILY_DPP_ADD(0x112, 0xF);                                 // row_shr:2
  BLURRILY_DPP_ADD(0x114, 0xF);                                 // row_shr:4
  BLURRILY_DPP_ADD(0x118, 0xF);                                 // row_shr:8
  BLURRILY_DPP_ADD(0x142, 0xA);                                 // row_bcast:15 -> rows 1 and 3
  BLURRILY_DPP_ADD(0x143, 0xC);                                 // row_bcast:31 -> rows 2 and 3
#undef BLURRILY_DPP_ADD
  return x;
}

struct UnitRing {
  // what a step starts with, ONE 8-byte read: .x the step the slot's units belong to (past the end: none left);
  // .y = units (bits 15:0; kRingWalk: too many, walk the table) | the scan's admission bound << 16 (bits 23:16;
  // 0: the slow scan, which works it out itself -- cold start) | slices left out of the even window's count << 24
  // (bits 27:24) | of the odd window's << 28
  uint2    hdr[2];
  uint32_t visit[2];                                            // (sweep_coop_plain: the visit index chosen most recently, by turns)
  uint32_t pad_[2];
  // per step (e & 3: a step's candidates are settled while the step after the next is being published) and window
  // parity: where the postings of the slices LEFT OUT of the count start in `ent` (their bitmaps sit in front of them)
  uint32_t hot[4][2][8];
  // behind it: desc[2][ring_units_for(pool_cap)], .x first entry of the unit, .y end of its slice; pend[2][kPendMax]
};
__device__ __forceinline__ uint2* ring_slot(UnitRing* ring, uint32_t slot, uint32_t ring_units) {
  return reinterpret_cast<uint2*>(ring + 1) + (slot ? ring_units : 0u);
}
constexpr uint32_t kRingWalk = 0xFFFFu;
constexpr uint32_t kNmMaxLeftOut = 8;                           // slices left out of one window's count at most (UnitRing::hot)
// the pending lists: behind the descriptors in the dynamic LDS
__device__ __forceinline__ uint32_t* pend_list(UnitRing* ring, uint32_t slot, uint32_t ring_units) {
  return reinterpret_cast<uint32_t*>(reinterpret_cast<uint2*>(ring + 1) + 2 * ring_units) + slot * kPendMax;
}

// sweep_coop leaves slices out of a step's count only where the candidates that leaves pending are sure of their
// place in the pool (sweep_role); the pool's last kAdmMax slots are then theirs
__device__ __forceinline__ bool coop_can_leave(const FindArgs& A) {
  return A.nm_cmin != 0 && A.pool_cap <= 1024 && !A.own_only && select_at(A) + 32 <= adm_max(A.pool_cap);
}

// A workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global load in flight
// (s_waitcnt vmcnt(0)): behind a step's count that is the manager's table of the step after the next, just requested,
// behind its scan a worker's first unit of the next step -- loads that are meant to travel across the barriers.  The
// waves of a sweep exchange data through LDS only.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// why sweep_coop's hot loop was left
enum : uint32_t { kLeftDone = 0, kLeftWalk = 1, kLeftSlowScan = 2, kLeftSelect = 3 };

// ---- cooperative flavour, every wave doing everything (rounds 2-3; since round 4 the sweep of launches that leave
// nothing out of a step's count: limits above 64, phase 1 of the window-major sweep, "nm_cmin" 0) ---------------------
// ONE wave per step (rotating) cuts the slices of the next visited window into units and publishes them as
// descriptors {first entry, slice end} in an LDS ring; in the next step every wave reads the descriptors of its units
// (unit k belongs to wave k mod kNW) and streams them, one unit's LDS atomics running under the next unit's load.
template <typename CT, int NT>
__device__ void sweep_coop_plain(const FindArgs& A, const Needle& nd, const uint16_t* codes, uint32_t* cnt32,
                           unsigned long long* pool, Control* ctl, UnitRing* ring, const uint8_t* wmt,
                           const uint32_t w0, const uint32_t w1, const uint32_t ws) {
  // A step covers kWPS windows: one with byte counters, two with 4-bit counters (CT = Nib).  Lane
  // t of the table holds trigram t's slice of the step's window -- of both windows with 4-bit
  // counters (second slot) -- and a unit's descriptor carries its window's parity in bit 0.
  constexpr bool kNib = std::is_same<CT, Nib>::value;
  constexpr uint32_t kWPS = kNib ? 2 : 1;
  constexpr uint32_t kNW = NT / 64;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t tc = nd.T;                                     // <= 64
  const bool own = lane < tc;
  const uint32_t code = own ? codes[lane] : 0u;
  const uint32_t v0 = w0 / kWPS, v1 = (w1 + kWPS - 1) / kWPS, vs = ws / kWPS;   // steps [v0, v1), first one vs
  const uint32_t n_visit = v1 - v0;
  uint4* const cnt128 = reinterpret_cast<uint4*>(cnt32);
  const uint32_t ring_units = ring_units_for(A.pool_cap), ring_rows = ring_units / kNW;   // rows: units of a wave
  // (the i-th step visited: upward from the needle's own length class, round the end.  Outward from it instead --
  // one step above, one below, by turns -- measured in round 3: 270.0 vs 253.2 ms per 500 k needles, 6.7 % slower;
  // downward from it, round the start: 292.7 ms, 16 % slower -- every window visited later lies BEHIND the threshold's
  // rank when the sweep goes upward, and needs one match more)
#define BLURRILY_STEP_AT(i_) ((i_) < n_visit ? (vs + (i_) < v1 ? vs + (i_) : vs + (i_) - n_visit) : v1)
  // most trigrams of the needle a reference of the step's window(s) can hold
#define BLURRILY_WMT_AT(i_, out_)                                                \
  do {                                                                           \
    const uint32_t p_ = min(BLURRILY_STEP_AT(i_), v1 - 1) * kWPS;                \
    if (wmt) {                       /* the workgroup's LDS copy (clamped to 255 >= tc): no global round trip */ \
      out_ = wmt[p_];                                                            \
      if (kNib && p_ + 1 < w1) out_ = max(out_, uint32_t(wmt[p_ + 1]));          \
    } else {                                                                     \
      out_ = A.win_max_tri[p_];                                                  \
      if (kNib && p_ + 1 < w1) out_ = max(out_, A.win_max_tri[p_ + 1]);          \
    }                                                                            \
  } while (0)
  // first visit index >= from_ whose step can hold a candidate (n_visit: none)
#define BLURRILY_NEXT_VISIT(from_, out_)                                         \
  do {                                                                           \
    out_ = (from_);                                                              \
    while (out_ < n_visit) {                                                     \
      uint32_t m_;                                                               \
      BLURRILY_WMT_AT(out_, m_);                                                 \
      if (min(tc, m_) >= matches_needed(ctl->thr, tc, BLURRILY_STEP_AT(out_) * kWPS * kWindowRanks)) break; \
      ++out_;                                                                    \
    }                                                                            \
  } while (0)
  // slice table of step p_: (A0, B0) the (even) window, (A1, B1) the odd one with 4-bit counters
#define BLURRILY_FETCH_TABLE(p_, A0, B0, A1, B1)                                 \
  do {                                                                           \
    A0 = B0 = A1 = B1 = 0;                                                       \
    const uint32_t w_ = (p_) * kWPS;                                             \
    if (STATS(A) && w_ < w1) st_tab += 2u * tc * (kNib && w_ + 1 < w1 ? 2u : 1u); \
    if (w_ < w1 && own) {                                                        \
      const uint32_t idx_ = w_ * kNumCodes + code;                               \
      const uint2 se_ = A.slice_se[idx_]; A0 = se_.x; B0 = se_.y;                \
    }                                                                            \
    if (kNib && w_ + 1 < w1 && own) {                                            \
      const uint32_t idx_ = (w_ + 1) * kNumCodes + code;                         \
      const uint2 se_ = A.slice_se[idx_]; A1 = se_.x; B1 = se_.y;                \
    }                                                                            \
  } while (0)
  // The header of step step_ in ring slot s_, by lane 0 of the publishing wave: the step, its unit count and the
  // bound its scan admits counters from -- worked out HERE, once, from the threshold as it is now, instead of by
  // every wave behind the count barrier.  The threshold can only tighten until that scan runs (in the select of
  // the step in between), so the published bound is at most too low: the scan then looks at a few counters more,
  // and every counter it looks at is tested against the threshold of the moment before it enters the pool.
  // 0 = no threshold yet and a cold start due: the slow scan.
#define BLURRILY_PUBLISH_HDR(s_, step_, nu_)                                     \
  do {                                                                           \
    const unsigned long long thr_ = ctl->thr;                                    \
    uint32_t need_ = min(matches_needed(thr_, tc, (step_) * kWPS * kWindowRanks), 0xFFFFu); \
    if (thr_ == kKeyInf && !nd.has_floor && !A.tomb && tc > 1) need_ = 0;        \
    if (lane == 0) ring->hdr[s_] = make_uint2((step_), (nu_) | (need_ << 16));   \
  } while (0)
  // (A window in which fewer of the needle's trigrams occur AT ALL than a candidate needs, left out by the publishing
  // wave -- two ballots over the table it holds -- measured in round 3: 255.5 vs 252.9 ms, 1 % slower: at Geonames
  // scale the bound hardly ever bites, and the wave that publishes is the one the count barrier waits for.)
  // this wave publishes the units of the table into ring slot s_: a lane's even-window units,
  // then its odd-window units.  Unit k belongs to wave k mod kNW and is that wave's (k / kNW)-th: a slot is laid out
  // wave by wave, so that ONE read -- lane j the wave's j-th unit -- hands a wave all its descriptors of a step.
#define BLURRILY_UNIT_AT(k_) (((k_) & (kNW - 1)) * ring_rows + (k_) / kNW)
#define BLURRILY_MY_UNITS(s_) (ring_slot(ring, s_, ring_units)[wid * ring_rows + min(lane, ring_rows - 1)])
#define BLURRILY_PRODUCE(s_, step_, A0, B0, A1, B1)                              \
  do {                                                                           \
    const uint32_t units0_ = slice_units(A0, B0), units1_ = kNib ? slice_units(A1, B1) : 0u; \
    const uint32_t incl_ = wave_inclusive_sum(units0_ + units1_);                \
    const uint32_t total_ = __builtin_amdgcn_readlane(incl_, 63);                \
    if (total_ > ring_units) {                                                   \
      BLURRILY_PUBLISH_HDR(s_, step_, kRingWalk);                                \
    } else {                                                                     \
      uint2* const slot_ = ring_slot(ring, s_, ring_units);                      \
      uint32_t at_ = incl_ - units0_ - units1_;                                  \
      for (uint32_t j_ = 0; j_ < units0_; ++j_, ++at_)                           \
        slot_[BLURRILY_UNIT_AT(at_)] = make_uint2(A0 + j_ * 512, B0);            \
      for (uint32_t j_ = 0; j_ < units1_; ++j_, ++at_)                           \
        slot_[BLURRILY_UNIT_AT(at_)] = make_uint2((A1 + j_ * 512) | 1u, B1);     \
      BLURRILY_PUBLISH_HDR(s_, step_, total_);                                   \
    }                                                                            \
  } while (0)
  // The units of ring slot s_ that belong to this wave (k = wid, wid + kNW, ...): one unit's LDS
  // atomics run while the next unit's load is in flight.  Which lanes loaded a group travels as a lane
  // predicate -- an SGPR pair -- beside the unit in flight: no sentinels to fill idle lanes with, no liveness
  // test before the atomics.  A unit's address is a scalar base (its first entry) plus the lane's 16 bytes:
  // nothing per unit and lane but the load itself and one compare.  (Up to four of a wave's units loaded before
  // the first is counted -- four loads in flight -- measured in round 3: 3.2 % SLOWER, 317.2 vs 307.4 ms.  Starting
  // the deal behind the two waves with a turn, so that they are the last to get one unit more: 0.7 % slower.)
#define BLURRILY_COUNT_UNITS(s_, n_, have_mine_)                                 \
  do {                                                                           \
    uint4 pend_ = make_uint4(0, 0, 0, 0);                                        \
    uint32_t pend_h_ = 0;                                                        \
    bool pend_live_ = false;                                                     \
    uint2 dl_ = d_mine;                         /* (read a step ago, behind the count barrier) */ \
    if (!(have_mine_)) dl_ = BLURRILY_MY_UNITS(s_);                              \
    uint32_t j_ = 0, k_ = wid;                                                   \
    if ((have_mine_) && pre_valid) {            /* the first unit is on its way since the scan before */ \
      pend_ = pre_v; pend_h_ = pre_h; pend_live_ = pre_live;                     \
      if (STATS(A) && wid < (n_))                                                \
        st_ent += min(512u, __builtin_amdgcn_readlane(dl_.y, 0) - (__builtin_amdgcn_readlane(dl_.x, 0) & ~7u)); \
      j_ = 1; k_ = wid + kNW;                                                    \
    }                                                                            \
    pre_valid = false;                                                           \
    for (; k_ < (n_); k_ += kNW, ++j_) {                                         \
      const uint32_t x_ = __builtin_amdgcn_readlane(dl_.x, j_);                  \
      const uint32_t y_ = __builtin_amdgcn_readlane(dl_.y, j_);                  \
      const uint32_t x0_ = x_ & ~7u;                                             \
      const bool live_ = lane8 < y_ - x0_;                                       \
      uint4 v_ = pend_;                                                          \
      if (live_) v_ = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(A.ent + x0_) + lane16); \
      if (STATS(A)) st_ent += min(512u, y_ - x0_);                               \
      if (pend_live_) bump_unit_loaded<CT>(cnt32, pend_, pend_h_);               \
      pend_ = v_; pend_h_ = x_ & 1u; pend_live_ = live_;                         \
    }                                                                            \
    if (pend_live_) bump_unit_loaded<CT>(cnt32, pend_, pend_h_);                 \
  } while (0)
  // Behind a scan, in front of its barrier: the header of the next step and this wave's units of it have arrived
  // (requested before the scan), the scan's registers are free -- the load of the wave's first unit of the next step
  // goes out here and travels under the barrier and the glance at the pool: 288.0 -> 281.3 ms per 500 k needles.
  // (The first TWO units loaded here and two loads kept in flight through the count: 313.0 ms, 11 % slower -- as
  // with every other attempt at more loads in flight per wave, rounds 2 and 3.  Two units loaded here and ONE load
  // in flight through the count: 330 ms while the compiler kept the second unit in scratch -- a value carried through
  // the rare paths is spilled where it is loaded --, 289.4 vs 281.9 ms, 2.7 % slower, once it was declared dead there.
  // A read-only kernel with this shape of access reaches 7.5 TB/s on the chip with one load in flight per wave
  // (tools/micro/read_bw.hip): what the loads wait for is not more of them.  The units' loads marked
  // non-temporal, global_load_dwordx4 ... nt: 330.6 ms, 17 % slower -- the postings of the Geonames-scale image
  // are 253 MB, and the 256 MiB Infinity Cache holds most of them as long as they are allowed in.)
#define BLURRILY_PRELOAD()                                                       \
  do {                                                                           \
    const uint32_t np_ = __builtin_amdgcn_readfirstlane(h_next.x);               \
    const uint32_t nn_ = __builtin_amdgcn_readfirstlane(h_next.y) & 0xFFFFu;     \
    pre_valid = np_ < v1 && nn_ != kRingWalk;                                    \
    pre_live = false;                                                            \
    if (pre_valid && wid < nn_) {                                                \
      const uint32_t x_ = __builtin_amdgcn_readlane(d_mine.x, 0);                \
      const uint32_t y_ = __builtin_amdgcn_readlane(d_mine.y, 0);                \
      const uint32_t x0_ = x_ & ~7u;                                             \
      pre_live = lane8 < y_ - x0_;                                               \
      pre_h = x_ & 1u;                                                           \
      if (pre_live) pre_v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(A.ent + x0_) + lane16); \
    }                                                                            \
  } while (0)
  // more units than the ring holds: every wave walks the table of step p_ itself
#define BLURRILY_COUNT_WALK(p_)                                                  \
  do {                                                                           \
    uint32_t fa0_, fb0_, fa1_, fb1_, k_ = 0;                                     \
    BLURRILY_FETCH_TABLE(p_, fa0_, fb0_, fa1_, fb1_);                            \
    BLURRILY_FOR_SLOT_UNITS(kNW, fa0_, fb0_, wid, lane, k_,                      \
                            { bump_unit<CT>(cnt32, load_group(A.ent, c, sb), 0u);   \
                              if (STATS(A)) st_ent += min(512u, sb - (c - lane * 8)); }); \
    if (kNib)                                                                    \
      BLURRILY_FOR_SLOT_UNITS(kNW, fa1_, fb1_, wid, lane, k_,                    \
                              { bump_unit<CT>(cnt32, load_group(A.ent, c, sb), 1u); \
                                if (STATS(A)) st_ent += min(512u, sb - (c - lane * 8)); }); \
  } while (0)
#define BLURRILY_PRODUCER(e_) ((e_) & (kNW - 1))              /* the publishing turn goes round the waves */
  // The two turns a step has behind its units.  The wave whose turn it is publishes the next visited step (its
  // table arrived a step ago); the wave after it chooses the step after the next (the threshold only changes
  // behind select's barriers) and fetches its table, which travels during the barrier and the scan.
#define BLURRILY_TAKE_TURNS(e_, s_)                                              \
  do {                                                                           \
    if (wid == BLURRILY_PRODUCER((e_) + 1)) {                                    \
      __builtin_amdgcn_s_setprio(3);                /* the wave the count barrier waits for goes first */ \
      if (my_i < n_visit) {                                                      \
        BLURRILY_PRODUCE((s_) ^ 1u, BLURRILY_STEP_AT(my_i), ta, tb, ta1, tb1);   \
      } else if (lane == 0) {                                                    \
        ring->hdr[(s_) ^ 1u] = make_uint2(v1, 0u);                               \
      }                                                                          \
      __builtin_amdgcn_s_setprio(0);                                             \
    }                                                                            \
    PHASE_MARK(7);                                  /* (producer turn) next step's units published */ \
    if (wid == BLURRILY_PRODUCER((e_) + 2)) {                                    \
      __builtin_amdgcn_s_setprio(3);                                             \
      const uint32_t chosen_ = __builtin_amdgcn_readfirstlane(ring->visit[((e_) + 1) & 1]); \
      BLURRILY_NEXT_VISIT(chosen_ + 1, my_i);                                    \
      if (my_i != chosen_ + 1) PATH_FLAG(A, nd.q, kPathSkipped);                 \
      if (lane == 0) ring->visit[(e_) & 1] = my_i;                               \
      BLURRILY_FETCH_TABLE(BLURRILY_STEP_AT(my_i), ta, tb, ta1, tb1);            \
      __builtin_amdgcn_s_setprio(0);                                             \
    }                                                                            \
  } while (0)

  const uint32_t lane8 = lane * 8, lane16 = lane * 16;
  uint32_t ta = 0, tb = 0, ta1 = 0, tb1 = 0;                    // table this wave will publish next
  PHASE_DECL;
  uint32_t st_ent = 0, st_tab = 0, st_steps = 0, st_redo = 0, st_walk = 0;    // request counters (FindArgs::stats), wave-uniform
  // Which step comes next is decided by ONE wave per step -- the one that then fetches that step's table --
  // and travels through LDS with the units (`hdr[slot]`; the visit index chosen last in `visit[]`).  The other
  // fifteen waves read one header per step instead of each running the window-bound loop, the 64-bit threshold
  // arithmetic and the step bookkeeping themselves.
  uint32_t my_i = 0;                                            // visit index of the table this wave holds
  PATH_FLAG(A, nd.q, kNib ? kPathNibble : kPathByte);
  if (wid == BLURRILY_PRODUCER(0u)) {
    BLURRILY_FETCH_TABLE(BLURRILY_STEP_AT(0u), ta, tb, ta1, tb1);
    BLURRILY_PRODUCE(0u, BLURRILY_STEP_AT(0u), ta, tb, ta1, tb1);
  }
  if (wid == BLURRILY_PRODUCER(1u)) {
    BLURRILY_NEXT_VISIT(1u, my_i);
    if (lane == 0) ring->visit[1] = my_i;
    BLURRILY_FETCH_TABLE(BLURRILY_STEP_AT(my_i), ta, tb, ta1, tb1);
  }
  // a threshold exists (it is set by compact_pool only, i.e. outside the hot loop below: re-read behind every exit)
  bool have_thr = __builtin_amdgcn_readfirstlane(uint32_t(ctl->thr != kKeyInf)) != 0;
  const uint32_t scan_cap = min(tc, ScanTraits<CT>::kMaxCount);  // a counter of this sweep cannot exceed it
  const uint32_t sel_at = select_at(A);
  __syncthreads();
  uint2 h_next = ring->hdr[0];                                   // header of the step about to start ...
  uint4 pre_v = make_uint4(0, 0, 0, 0);                          // the wave's first unit of the next step, loaded ahead
  uint32_t pre_h = 0;
  bool pre_live = false, pre_valid = false;
  uint2 d_mine = BLURRILY_MY_UNITS(0u);  // ... and this wave's units of it (lane j: its j-th)

  // The sweep is a HOT LOOP of steps that need nothing special -- header, units, turns, barrier, scan with the
  // published bound, barrier, a glance at the pool -- and is left for everything else (more units than the ring
  // holds; no threshold yet: the cold start's bisection; the pool to be compacted, perhaps the step swept again),
  // which is dealt with behind it before the loop is entered again.  Kept apart so that what only the rare paths
  // need is not held in registers, nor worked out, step after step.
  uint32_t e = 0;
  for (;;) {
    uint32_t left, s, p, n_units;
    for (;; ++e) {
      s = e & 1;
      p = __builtin_amdgcn_readfirstlane(h_next.x);
      const uint32_t hy_ = __builtin_amdgcn_readfirstlane(h_next.y);
      n_units = hy_ & 0xFFFFu;
      if (p >= v1) { left = kLeftDone; break; }                 // no step left
      ++st_steps;
      PHASE_MARK(0);                                            // loop overhead
      // (the two waves with a turn to take behind their units are the ones the count barrier waits for: they
      // issue ahead of the others from the start of the step)
      if (wid == BLURRILY_PRODUCER(e + 1) || wid == BLURRILY_PRODUCER(e + 2)) __builtin_amdgcn_s_setprio(2);
      if (n_units == kRingWalk) { left = kLeftWalk; break; }
      BLURRILY_COUNT_UNITS(s, n_units, true);
      PHASE_MARK(2);                                            // units counted
      BLURRILY_TAKE_TURNS(e, s);
      __syncthreads();                                          // counts and next descriptors visible
      PHASE_MARK(3);                                            // barrier after count
      // The next step's header and this wave's units of it were published before that barrier: requested now,
      // they arrive under the scan instead of standing, one LDS round trip each (several hundred clocks behind the
      // other workgroup's atomics), between the scan barrier and the first load of the next step.
      h_next = ring->hdr[s ^ 1u];
      d_mine = BLURRILY_MY_UNITS(s ^ 1u);
      if (n_units == 0) continue;                               // nothing of the needle in this step's windows
      const uint32_t need = hy_ >> 16;
      if (need == 0) { left = kLeftSlowScan; break; }
      const uint32_t wbase = p * kWPS * kWindowRanks;
      const uint32_t wlen = min(kWPS * kWindowRanks, A.n_refs - wbase);
      scan_core<CT, NT>(cnt128, nd, need, scan_cap, &ctl->thr, &ctl->floor, A.tomb, pool, A.pool_cap, &ctl->pool_n,
                        &ctl->overflow, wbase, wlen, STATS(A) && A.path_flags ? &A.path_flags[nd.q] : nullptr);
      BLURRILY_PRELOAD();
      PHASE_MARK(5);                                            // scan
      __syncthreads();                                          // counters are zero again
      PHASE_MARK(6);                                            // barrier after scan
      // (Looking at the pool a count phase later -- the read requested here, used behind the next count barrier, a
      // compaction then running with the next step counted and not yet scanned -- was built and measured in round 3:
      // 2 % slower, 293.0 vs 287.2 ms per 500 k needles; the thresholds the headers carry are a step staler.)
      const uint2 c_ = *reinterpret_cast<const uint2*>(&ctl->pool_n);          // pool_n, overflow
      const uint32_t pn_ = __builtin_amdgcn_readfirstlane(c_.x), ov_ = __builtin_amdgcn_readfirstlane(c_.y);
      if (ov_ != 0 || pn_ > sel_at || (!have_thr && pn_ >= A.keep)) { left = kLeftSelect; break; }
    }
    if (left == kLeftDone) break;
    // ---- the rare paths of step p ----------------------------------------------------------------
    pre_valid = false;                                          // (a unit loaded ahead is dropped)
    const uint32_t wbase = p * kWPS * kWindowRanks;
    const uint32_t wlen = min(kWPS * kWindowRanks, A.n_refs - wbase);
    if (left == kLeftWalk) {
      PATH_FLAG(A, nd.q, kPathRingOverflow);
      ++st_walk;
      BLURRILY_COUNT_WALK(p);
      BLURRILY_TAKE_TURNS(e, s);
      __syncthreads();
    }
    // kLeftSelect: the step is scanned, select_after_scan finds the pool as the hot loop saw it; else: scan first
    bool scanned = left == kLeftSelect;
    for (;;) {
      if (!scanned) {
        scan_window<CT, NT>(A, nd, cnt128, pool, ctl, wbase, wlen);
        __syncthreads();
      }
      scanned = false;
      if (!select_after_scan<NT>(A, pool, ctl, wbase, wlen, nd.q)) break;
      ++st_redo;                                                // pool overflow: sweep step p again
      if (n_units == kRingWalk) BLURRILY_COUNT_WALK(p);
      else BLURRILY_COUNT_UNITS(s, n_units, false);
      __syncthreads();
    }
    have_thr = __builtin_amdgcn_readfirstlane(uint32_t(ctl->thr != kKeyInf)) != 0;
    ++e;
    h_next = ring->hdr[e & 1];                                  // (published behind step p's count barrier)
    d_mine = BLURRILY_MY_UNITS(e & 1);
  }
  PHASE_FLUSH(A);
  if (STATS(A) && lane == 0) {
    atomicAdd(&STATS(A)[kStatPostingEntries], static_cast<unsigned long long>(st_ent));
    atomicAdd(&STATS(A)[kStatTableWords], static_cast<unsigned long long>(st_tab));
    if (wid == 0) {
      atomicAdd(&STATS(A)[kStatSteps], static_cast<unsigned long long>(st_steps));
      atomicAdd(&STATS(A)[kStatResweeps], static_cast<unsigned long long>(st_redo));
      atomicAdd(&STATS(A)[kStatUnits], static_cast<unsigned long long>(st_walk));   // (needle-major: steps that walked the table)
    }
  }
  (void)st_walk;
  __syncthreads();                                              // ring and ctl quiet before the needle ends
#undef BLURRILY_TAKE_TURNS
#undef BLURRILY_PRODUCER
#undef BLURRILY_COUNT_WALK
#undef BLURRILY_COUNT_UNITS
#undef BLURRILY_PRELOAD
#undef BLURRILY_PRODUCE
#undef BLURRILY_MY_UNITS
#undef BLURRILY_UNIT_AT
#undef BLURRILY_PUBLISH_HDR
#undef BLURRILY_FETCH_TABLE
#undef BLURRILY_NEXT_VISIT
#undef BLURRILY_WMT_AT
#undef BLURRILY_STEP_AT
}


// inclusive sums of a published slice table's units (sweep_role: BLURRILY_TAB), lane t: the units up to and with slice t
__device__ __forceinline__ uint32_t tab_incl(const uint4 tb) {
  return (tb.w & 0xFFFFu) + (tb.w >> 16) + ((((tb.z >> 16) >> 3) + 63u) >> 6);
}

// One needle's sweep as ONE of two roles (round 4).  Through round 3 every wave of the workgroup did everything: its
// share of a step's units, its share of the scan -- and, by turns, the choosing of the next step, the fetching of
// its slice table and the publishing of its units, so that every wave held the table's registers and the turns' code,
// and the wave with a turn, which took it BEHIND its own units, was the one the count barrier waited for.  Now the
// workgroup's last wave is the MANAGER: it chooses, fetches and publishes (and, since this round, decides which dense
// slices a step leaves out of its count and settles the candidates that leaves pending) while the fifteen WORKERS
// count -- and counts and scans nothing itself; the workers count, scan and glance at the pool, and hold none of the
// manager's state.  Both roles run the same sequence of barriers and leave their hot loops on the same, uniform
// conditions (the step's header, the pool's state behind a scan); the rare paths behind the loops are common code.
// The two are separate instantiations called from disjoint branches, so that what one role keeps in registers is not
// live in the other's loop (every attempt of this round to add the left-out slices' bookkeeping to the shared loop cost
// 6-17 % with NOTHING left out: 64 VGPRs, two workgroups a CU).
template <typename CT, int NT, bool MANAGER>
__device__ __forceinline__ void sweep_role(const FindArgs& A, const Needle& nd, const uint16_t* codes, uint32_t* cnt32,
                                           unsigned long long* pool, Control* ctl, UnitRing* ring, const uint8_t* wmt,
                                           const uint32_t w0, const uint32_t w1, const uint32_t ws) {
  // A step covers kWPS windows: one with byte counters, two with 4-bit counters (CT = Nib).  Lane
  // t of the table holds trigram t's slice of the step's window -- of both windows with 4-bit
  // counters (second slot) -- and a unit's descriptor carries its window's parity in bit 0.
  constexpr bool kNib = std::is_same<CT, Nib>::value;
  constexpr uint32_t kWPS = kNib ? 2 : 1;
  constexpr uint32_t kNW = NT / 64, kWorkers = kNW - 1;
  const uint32_t tid = threadIdx.x;
  // (the lane number taken HERE, opaquely: derived from threadIdx.x it is a kernel-lifetime value, and with it every
  // address and mask computed from it -- hoisted to the kernel's entry for both roles and all four sweeps, spilled there
  // and reloaded from scratch inside the hot loops, each reload a wait for every global load in flight)
  uint32_t lane;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  const uint32_t wid = MANAGER ? kWorkers : __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t tc = nd.T;                                     // <= 64
  const uint32_t v0 = w0 / kWPS, v1 = (w1 + kWPS - 1) / kWPS, vs = ws / kWPS;   // steps [v0, v1), first one vs
  const uint32_t n_visit = v1 - v0;
  uint4* const cnt128 = reinterpret_cast<uint4*>(cnt32);
  const uint32_t ring_units = ring_units_for(A.pool_cap);      // (where the pending lists start behind the ring: pend_list)
  // Slices are left out of a step's count only where the candidates that leaves pending are sure of their place in the
  // pool: limits up to 64 (the 512-entry pool), its last kAdmMax slots theirs alone, a step's pending list no longer than
  // what those slots hold beside the keys a glance lets pass (select_at()); not in phase 1 of the window-major sweep.
  const uint32_t sel_at = select_at(A);
  const bool can_leave = coop_can_leave(A);
  const uint32_t scan_pool_cap = can_leave ? A.pool_cap - adm_max(A.pool_cap) : A.pool_cap;
  const uint32_t pend_cap = can_leave ? min(kPendMax, adm_max(A.pool_cap) - sel_at) : 0u;
  // (the i-th step visited: upward from the needle's own length class, round the end.  Outward from it instead --
  // one step above, one below, by turns -- measured in round 3: 270.0 vs 253.2 ms per 500 k needles, 6.7 % slower;
  // downward from it, round the start: 292.7 ms, 16 % slower -- every window visited later lies BEHIND the threshold's
  // rank when the sweep goes upward, and needs one match more)
#define BLURRILY_STEP_AT(i_) ((i_) < n_visit ? (vs + (i_) < v1 ? vs + (i_) : vs + (i_) - n_visit) : v1)
  // Which step comes next: the first visit index >= from_ whose windows can hold a candidate (n_visit: none) -- a window
  // is stepped over when no reference of it has as many trigrams as a candidate needs (win_max_tri).  The manager keeps
  // the bounds of 64 visits in a register (lane l: visit wm_base + l; reloaded every 64 visits) and the threshold in
  // scalar registers (thr_c: it moves only in compact_pool, i.e. outside the hot loop), so that choosing is one compare
  // and a ballot -- no LDS round trip, which behind the workers' atomics takes 500 to 1 000 clocks each (through the first
  // version of this round two dependent ones per step looked at: the bound's byte, the threshold).
#define BLURRILY_LOAD_WMT(base_)                                                 \
  do {                                                                           \
    const uint32_t i_ = (base_) + lane;                                          \
    uint32_t m_ = 0;                                                             \
    if (i_ < n_visit) {                                                          \
      const uint32_t p_ = BLURRILY_STEP_AT(i_) * kWPS;                           \
      if (wmt) {                     /* the workgroup's LDS copy (clamped to 255 >= tc) */ \
        m_ = wmt[p_];                                                            \
        if (kNib && p_ + 1 < w1) m_ = max(m_, uint32_t(wmt[p_ + 1]));            \
      } else {                                                                   \
        m_ = A.win_max_tri[p_];                                                  \
        if (kNib && p_ + 1 < w1) m_ = max(m_, A.win_max_tri[p_ + 1]);            \
      }                                                                          \
    }                                                                            \
    wm_l = m_; wm_base = (base_);                                                \
  } while (0)
#define BLURRILY_NEXT_VISIT(from_, out_)                                         \
  do {                                                                           \
    uint32_t f_ = (from_);                                                       \
    out_ = n_visit;                                                              \
    while (f_ < n_visit) {                                                       \
      const uint32_t b_ = f_ & ~63u;                                             \
      if (b_ != wm_base) BLURRILY_LOAD_WMT(b_);                                  \
      const uint32_t i_ = b_ + lane;                                             \
      const uint32_t need_l_ = matches_needed(thr_c, tc, BLURRILY_STEP_AT(i_) * kWPS * kWindowRanks); \
      const unsigned long long ok_ = __ballot(i_ < n_visit && min(tc, wm_l) >= need_l_) >> (f_ - b_); \
      if (ok_) { out_ = f_ + uint32_t(__builtin_ctzll(ok_)); break; }           \
      f_ = b_ + 64u;                                                             \
    }                                                                            \
  } while (0)
  // slice table of step p_ (lane t: the needle's trigram code_): (A0, B0) the (even) window, (A1, B1) the odd one
  // with 4-bit counters; a dense slice's postings start behind its bitmap (postings_start)
#define BLURRILY_FETCH_TABLE(p_, code_, A0, B0, A1, B1)                          \
  do {                                                                           \
    A0 = B0 = A1 = B1 = 0;                                                       \
    const uint32_t w_ = (p_) * kWPS;                                             \
    if (STATS(A) && w_ < w1) st_tab += 2u * tc * (kNib && w_ + 1 < w1 ? 2u : 1u); \
    if (w_ < w1 && lane < tc) {                                                  \
      const uint32_t idx_ = w_ * kNumCodes + (code_);                            \
      const uint2 se_ = A.slice_se[idx_]; A0 = se_.x; B0 = se_.y;                \
    }                                                                            \
    if (kNib && w_ + 1 < w1 && lane < tc) {                                      \
      const uint32_t idx_ = (w_ + 1) * kNumCodes + (code_);                      \
      const uint2 se_ = A.slice_se[idx_]; A1 = se_.x; B1 = se_.y;                \
    }                                                                            \
  } while (0)
  // The header of step step_ in ring slot s_, by lane 0 of the manager: the step, its unit count, the bound its scan
  // admits counters from -- worked out HERE, once, from the threshold as it is now, instead of by every wave behind
  // the count barrier -- and how many slices each of its windows leaves out of the count.  The threshold can only
  // tighten until that scan runs (in the select of the step in between), so the published bound is at most too low:
  // the scan then looks at a few counters more, and every counter it looks at is tested against the threshold of the
  // moment before it enters the pool.  A bound of 0 = no threshold yet and a cold start due: the slow scan.
#define BLURRILY_PUBLISH_HDR(s_, step_, nu_, need_, ls_)                         \
  do {                                                                           \
    if (lane == 0) ring->hdr[s_] = make_uint2((step_), (nu_) | ((need_) << 16) | ((ls_) << 24)); \
  } while (0)
  // (A window in which fewer of the needle's trigrams occur AT ALL than a candidate needs, left out by the publishing
  // wave -- two ballots over the table it holds -- measured in round 3: 255.5 vs 252.9 ms, 1 % slower: at Geonames
  // scale the bound hardly ever bites.)
  // What the manager publishes of a step is its SLICE TABLE, one 16-byte store for the wave: lane t = the needle's
  // trigram t -- .x / .y where the slice's postings start in `ent` (even / odd window of the step), .z their lengths in
  // entries (16 bits each; 0: empty, or left out of the count), .w the units in front of the lane's (exclusive prefix
  // sum, bits 15:0) and the units of its even slice (bits 23:16).  Unit k of the step belongs to worker k mod 15; a
  // worker reads the table once per step (one 16-byte read, where it read its descriptors before) and finds unit k's
  // slice with one compare and ballot over the inclusive sums (tab_incl), its bounds with four v_readlane.  (Through
  // round-4's first version the manager listed every UNIT -- a store per unit from lanes looping over their slices,
  // queued behind the workers' atomics: 4 300 clocks per step, the wave the count barrier waited for in nine steps of ten.)
#define BLURRILY_TAB(s_) (reinterpret_cast<uint4*>(ring + 1) + (s_) * 64u)
  // scalar bounds of the step's unit k_ from the table in tb_ / its inclusive sums incl_
#define BLURRILY_UNIT_OF(tb_, incl_, k_, start_, end_, half_)                    \
  do {                                                                           \
    const uint32_t t_ = uint32_t(__builtin_ctzll(__ballot((incl_) > (k_)) | (1ull << 63))); \
    const uint32_t a0_ = __builtin_amdgcn_readlane((tb_).x, t_), a1_ = __builtin_amdgcn_readlane((tb_).y, t_); \
    const uint32_t z_ = __builtin_amdgcn_readlane((tb_).z, t_), w_ = __builtin_amdgcn_readlane((tb_).w, t_); \
    const uint32_t ji_ = (k_) - (w_ & 0xFFFFu), u0_ = w_ >> 16;                  \
    half_ = ji_ >= u0_ ? 1u : 0u;                                                \
    start_ = half_ ? a1_ + (ji_ - u0_) * 512u : a0_ + ji_ * 512u;                \
    end_ = half_ ? a1_ + (z_ >> 16) : a0_ + (z_ & 0xFFFFu);                      \
  } while (0)
  // LEFT OUT of the count (round 4; the MaxScore argument of wsweep_kernel, inside the needle-major step): once the
  // needle has a threshold -- `need_` matches to enter its top `keep` in this step's windows -- the l_max_ = need_ -
  // nm_cmin LARGEST slices of at least nm_dense postings of a window need not be counted.  A reference with need_
  // matches has at least need_ - L of them among the counted slices, so the scan finds it with the bound lowered by
  // L; what it finds goes to the step's pending list, and the manager settles its exact count from the left-out
  // slices' bitmaps during the next step (PEND_SETTLE).  Lane t ranks its slice among the window's dense ones by size
  // (one readlane per dense slice); the chosen ones note where their postings start (UnitRing::hot) and list no units.
#define BLURRILY_LEAVE_OUT(hs_, h_, A_, rk_, units_)                             \
  do {                                                                           \
    const bool skip_ = (rk_) < l_max_;                                           \
    const unsigned long long sm_ = __ballot(skip_);                              \
    if (skip_) {                                                                 \
      ring->hot[hs_][h_][__builtin_amdgcn_mbcnt_hi(uint32_t(sm_ >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(sm_), 0u))] = (A_); \
      units_ = 0;                                                                \
    }                                                                            \
    ls_ |= uint32_t(__popcll(sm_)) << (4u * (h_));                               \
  } while (0)
  // A slice's rank by size among the dense slices of its window (lane t: trigram t's; 0xFF: not dense) -- what
  // BLURRILY_LEAVE_OUT holds against l_max_.  It does not depend on the threshold, so the manager works it out for the
  // table it will publish NEXT while the workers scan (it has the time there: the count barrier is the one that waits
  // for it), one v_readlane per dense slice of the window.
#define BLURRILY_RANK_ONE(A_, B_, rk_)                                           \
  do {                                                                           \
    const uint32_t size_ = (B_) - (A_);                                          \
    const bool dense_ = size_ >= A.nm_dense;                                     \
    uint32_t bigger_ = 0;                                                        \
    for (unsigned long long m_ = __ballot(dense_); m_; m_ &= m_ - 1) {           \
      const uint32_t u_ = __builtin_ctzll(m_);                                   \
      const uint32_t su_ = __builtin_amdgcn_readlane(size_, u_);                 \
      bigger_ += (su_ > size_ || (su_ == size_ && u_ < lane)) ? 1u : 0u;         \
    }                                                                            \
    rk_ = dense_ ? bigger_ : 0xFFu;                                              \
  } while (0)
#define BLURRILY_RANK()                                                          \
  do {                                                                           \
    BLURRILY_RANK_ONE(ta, tb, rk0);                                              \
    if (kNib) BLURRILY_RANK_ONE(ta1, tb1, rk1);                                  \
    ranked = true;                                                               \
  } while (0)
  // the manager publishes the step's slice table into ring slot s_ (a lane's even-window units come first, then its
  // odd-window units)
#define BLURRILY_PRODUCE(s_, hs_, step_, A0, B0, A1, B1)                         \
  do {                                                                           \
    const unsigned long long thr_ = thr_c;                                       \
    uint32_t need_ = min(matches_needed(thr_, tc, (step_) * kWPS * kWindowRanks), 0xFFu); \
    uint32_t units0_ = slice_units(A0, B0), units1_ = kNib ? slice_units(A1, B1) : 0u; \
    uint32_t ls_ = 0;                                                            \
    const uint32_t l_max_ = (can_leave && thr_ != kKeyInf && need_ > A.nm_cmin) ? min(need_ - A.nm_cmin, kNmMaxLeftOut) : 0u; \
    if (l_max_) {                                                                \
      if (!ranked) BLURRILY_RANK();            /* (the first step of a sweep, the one behind a rare path) */ \
      BLURRILY_LEAVE_OUT(hs_, 0u, A0, rk0, units0_);                             \
      if (kNib) BLURRILY_LEAVE_OUT(hs_, 1u, A1, rk1, units1_);                   \
    }                                                                            \
    ranked = false;                            /* (the table is about to be replaced) */ \
    if (thr_ == kKeyInf && !nd.has_floor && !A.tomb && tc > 1) need_ = 0;        \
    const uint32_t incl_ = wave_inclusive_sum(units0_ + units1_);                \
    const uint32_t total_ = __builtin_amdgcn_readlane(incl_, 63);                \
    /* (a slice left out of the count lists no units: its length is published as 0) */ \
    const uint32_t len0_ = units0_ ? (B0) - (A0) : 0u, len1_ = units1_ ? (B1) - (A1) : 0u; \
    BLURRILY_TAB(s_)[lane] = make_uint4((A0), (A1), len0_ | (len1_ << 16), (incl_ - units0_ - units1_) | (units0_ << 16)); \
    /* the scan's bound, lowered by the most slices either window leaves out (a harvested counter is settled    \
       exactly, so a bound that is too low for the other window costs a look, nothing else) */ \
    const uint32_t lmost_ = max(ls_ & 15u, ls_ >> 4);                            \
    BLURRILY_PUBLISH_HDR(s_, step_, total_, need_ - min(need_, lmost_), ls_);    \
  } while (0)
  // The units of ring slot s_ that belong to this worker (k = wid, wid + 15, ...): one unit's LDS
  // atomics run while the next unit's load is in flight.  Which lanes loaded a group travels as a lane
  // predicate -- an SGPR pair -- beside the unit in flight: no sentinels to fill idle lanes with, no liveness
  // test before the atomics.  A unit's address is a scalar base (its first entry) plus the lane's 16 bytes:
  // nothing per unit and lane but the load itself and one compare.  (Up to four of a wave's units loaded before
  // the first is counted -- four loads in flight -- measured in round 3: 3.2 % SLOWER, 317.2 vs 307.4 ms.)
#define BLURRILY_COUNT_UNITS(s_, n_, have_mine_)                                 \
  do {                                                                           \
    uint4 pend_ = make_uint4(0, 0, 0, 0);                                        \
    uint32_t pend_h_ = 0;                                                        \
    bool pend_live_ = false;                                                     \
    uint4 tl_ = tb_mine;                        /* (read a step ago, behind the count barrier) */ \
    uint32_t il_ = incl_mine;                                                    \
    if (!(have_mine_)) { tl_ = BLURRILY_TAB(s_)[lane]; il_ = tab_incl(tl_); }    \
    uint32_t k_ = wid;                                                           \
    if ((have_mine_) && pre_valid) {            /* the first unit is on its way since the scan before */ \
      pend_ = pre_v; pend_h_ = pre_h; pend_live_ = pre_live;                     \
      if (STATS(A) && wid < (n_)) {                                              \
        uint32_t x_, y_, h_;                                                     \
        BLURRILY_UNIT_OF(tl_, il_, wid, x_, y_, h_);                             \
        st_ent += min(512u, y_ - x_);                                            \
      }                                                                          \
      k_ = wid + kWorkers;                                                       \
    }                                                                            \
    pre_valid = false;                                                           \
    for (; k_ < (n_); k_ += kWorkers) {                                          \
      uint32_t x_, y_, h_;                                                       \
      BLURRILY_UNIT_OF(tl_, il_, k_, x_, y_, h_);                                \
      const bool live_ = lane8 < y_ - x_;                                        \
      uint4 v_ = pend_;                                                          \
      if (live_) v_ = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(A.ent + x_) + lane16); \
      if (STATS(A)) st_ent += min(512u, y_ - x_);                                \
      if (pend_live_) bump_unit_loaded<CT>(cnt32, pend_, pend_h_);               \
      pend_ = v_; pend_h_ = h_; pend_live_ = live_;                              \
    }                                                                            \
    if (pend_live_) bump_unit_loaded<CT>(cnt32, pend_, pend_h_);                 \
  } while (0)
  // Behind a scan, in front of its barrier: the header of the next step and this wave's units of it have arrived
  // (requested before the scan), the scan's registers are free -- the load of the wave's first unit of the next step
  // goes out here and travels under the barrier and the glance at the pool: 288.0 -> 281.3 ms per 500 k needles.
  // (The first TWO units loaded here and two loads kept in flight through the count: 313.0 ms, 11 % slower -- as
  // with every other attempt at more loads in flight per wave, rounds 2 and 3.  Two units loaded here and ONE load
  // in flight through the count: 330 ms while the compiler kept the second unit in scratch -- a value carried through
  // the rare paths is spilled where it is loaded --, 289.4 vs 281.9 ms, 2.7 % slower, once it was declared dead there.
  // A read-only kernel with this shape of access reaches 7.5 TB/s on the chip with one load in flight per wave
  // (tools/micro/read_bw.hip): what the loads wait for is not more of them.  The units' loads marked
  // non-temporal, global_load_dwordx4 ... nt: 330.6 ms, 17 % slower -- the postings of the Geonames-scale image
  // are 253 MB, and the 256 MiB Infinity Cache holds most of them as long as they are allowed in.)
#define BLURRILY_PRELOAD()                                                       \
  do {                                                                           \
    const uint32_t np_ = __builtin_amdgcn_readfirstlane(h_next.x);               \
    const uint32_t nn_ = __builtin_amdgcn_readfirstlane(h_next.y) & 0xFFFFu;     \
    incl_mine = tab_incl(tb_mine);                                               \
    pre_valid = np_ < v1;                                                        \
    pre_live = false;                                                            \
    if (pre_valid && wid < nn_) {                                                \
      uint32_t x_, y_;                                                           \
      BLURRILY_UNIT_OF(tb_mine, incl_mine, wid, x_, y_, pre_h);                  \
      pre_live = lane8 < y_ - x_;                                                \
      if (pre_live) pre_v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(A.ent + x_) + lane16); \
    }                                                                            \
  } while (0)
  // a step swept AGAIN with every slice counted (pool overflow where slices were left out): every wave -- the manager
  // too -- fetches the table of step p_ itself and walks it
#define BLURRILY_COUNT_WALK(p_)                                                  \
  do {                                                                           \
    uint32_t fa0_, fb0_, fa1_, fb1_, k_ = 0;                                     \
    const uint32_t wcode_ = lane < tc ? codes[lane] : 0u;                        \
    BLURRILY_FETCH_TABLE(p_, wcode_, fa0_, fb0_, fa1_, fb1_);                    \
    BLURRILY_FOR_SLOT_UNITS(kNW, fa0_, fb0_, wid, lane, k_,                      \
                            { bump_unit<CT>(cnt32, load_group(A.ent, c, sb), 0u);   \
                              if (STATS(A)) st_ent += min(512u, sb - (c - lane * 8)); }); \
    if (kNib)                                                                    \
      BLURRILY_FOR_SLOT_UNITS(kNW, fa1_, fb1_, wid, lane, k_,                    \
                              { bump_unit<CT>(cnt32, load_group(A.ent, c, sb), 1u); \
                                if (STATS(A)) st_ent += min(512u, sb - (c - lane * 8)); }); \
  } while (0)
  // The manager's turn of a step: it publishes the next visited step (whose table arrived a step ago), then chooses the
  // step after the next (the threshold only changes behind select's barriers) and fetches its table, which travels
  // during the barrier and the scan.
#define BLURRILY_MANAGER_TURN(e_, s_)                                            \
  do {                                                                           \
    if (my_i < n_visit) {                                                        \
      BLURRILY_PRODUCE((s_) ^ 1u, ((e_) + 1u) & 3u, BLURRILY_STEP_AT(my_i), ta, tb, ta1, tb1); \
    } else if (lane == 0) {                                                      \
      ring->hdr[(s_) ^ 1u] = make_uint2(v1, 0u);                                 \
    }                                                                            \
    PHASE_MARK(7);                                  /* next step's units published */ \
    TRACE_MARK(A, nd.q, (e_), 1u, 7u);                                           \
    const uint32_t chosen_ = my_i;                                               \
    BLURRILY_NEXT_VISIT(chosen_ + 1, my_i);                                      \
    if (my_i != chosen_ + 1) PATH_FLAG(A, nd.q, kPathSkipped);                   \
    BLURRILY_FETCH_TABLE(BLURRILY_STEP_AT(my_i), mcode, ta, tb, ta1, tb1);       \
    PHASE_MARK(2);                                  /* step after the next chosen, its table requested */ \
  } while (0)
  // The manager settles the PENDING candidates of ring slot slot_ (step pstep_, whose windows left ls_ slices out,
  // noted in hot[hs_]): lane c takes candidate c -- in-window rank | parity << 16 | counted matches << 20 --, loads the
  // word of every left-out slice's bitmap that holds the candidate's bit (up to eight loads in flight), adds the bits
  // to the counted matches and admits the candidate to the pool, or not, like any harvested counter: into the pool's
  // tail, which is the manager's alone (compact_pool moves it up), so that nothing a flooding scan does to the pool
  // meanwhile can cost a settled candidate its place.  Done while the workers scan the next step.
#define BLURRILY_PEND_SETTLE(slot_, hs_, pstep_, ls_)                            \
  do {                                                                           \
    const uint32_t np_ = min(uint32_t(__builtin_amdgcn_readfirstlane(ctl->pend_n[slot_])), pend_cap); \
    if (np_) {                                                                   \
      const uint32_t* const pl_ = pend_list(ring, slot_, ring_units);            \
      for (uint32_t c_ = lane; c_ < np_; c_ += 64) {                             \
        const uint32_t e_ = pl_[c_];                                             \
        const uint32_t h_ = (e_ >> 16) & 1u, r16_ = e_ & 0xFFFFu;                \
        const uint32_t L_ = ((ls_) >> (4u * h_)) & 15u;                          \
        /* where the left-out slices' postings start: two 16-byte reads, then the loads together */ \
        const uint4 ha_ = *reinterpret_cast<const uint4*>(&ring->hot[hs_][h_][0]); \
        const uint4 hb_ = *reinterpret_cast<const uint4*>(&ring->hot[hs_][h_][4]); \
        const uint32_t hs8_[kNmMaxLeftOut] = {ha_.x, ha_.y, ha_.z, ha_.w, hb_.x, hb_.y, hb_.z, hb_.w}; \
        uint32_t w_[kNmMaxLeftOut];                                              \
        _Pragma("unroll") for (uint32_t k_ = 0; k_ < kNmMaxLeftOut; ++k_) {      \
          w_[k_] = 0;                                                            \
          if (k_ < L_)                                                           \
            w_[k_] = reinterpret_cast<const uint32_t*>(A.ent + (hs8_[k_] - kBitmapSlots))[r16_ >> 5]; \
        }                                                                        \
        uint32_t cnt_ = e_ >> 20;                                                \
        _Pragma("unroll") for (uint32_t k_ = 0; k_ < kNmMaxLeftOut; ++k_) cnt_ += (w_[k_] >> (r16_ & 31u)) & 1u; \
        if (STATS(A)) atomicAdd(&STATS(A)[kStatProbes], static_cast<unsigned long long>(L_)); \
        const uint32_t rank_ = (pstep_) * kWPS * kWindowRanks + h_ * kWindowRanks + r16_; \
        const unsigned long long key_ = (static_cast<unsigned long long>(tc - min(tc, cnt_)) << 32) | rank_; \
        bool pass_ = key_ <= thr_c;                                              \
        if (nd.has_floor) pass_ = pass_ && key_ > ctl->floor;                    \
        if (pass_) {                                     /* (tombstones were looked at where it was harvested) */ \
          const uint32_t at_ = atomicAdd(&ctl->adm_n, 1u);                       \
          if (at_ < adm_max(A.pool_cap)) pool[A.pool_cap - adm_max(A.pool_cap) + at_] = key_; \
        }                                                                        \
      }                                                                          \
      if (lane == 0) ctl->pend_n[slot_] = 0;                                     \
    }                                                                            \
  } while (0)

  const uint32_t lane8 = lane * 8, lane16 = lane * 16;
  PHASE_DECL;
  uint32_t st_ent = 0, st_tab = 0, st_steps = 0, st_redo = 0, st_walk = 0;    // request counters (FindArgs::stats), wave-uniform
  // ---- the manager's state: the needle's codes, the table it will publish next and its visit index, the step before
  uint32_t mcode = 0, ta = 0, tb = 0, ta1 = 0, tb1 = 0, my_i = 0, p_prev = 0, ls_prev = 0;
  uint32_t rk0 = 0xFFu, rk1 = 0xFFu;                            // ranks of the held table's slices (BLURRILY_RANK)
  bool ranked = false;
  uint32_t wm_l = 0, wm_base = 0xFFFFFFFFu;                     // win_max_tri of 64 visits (BLURRILY_LOAD_WMT)
  // the threshold as scalars: set by compact_pool only, i.e. outside the hot loop -- read again behind every exit
  unsigned long long thr_c = ctl->thr;
  thr_c = (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(uint32_t(thr_c >> 32))) << 32) |
          __builtin_amdgcn_readfirstlane(uint32_t(thr_c));
  // ---- a worker's: its units of the step about to start (lane j: its j-th), the first of them loaded ahead
  uint4 tb_mine = make_uint4(0, 0, 0, 0);
  uint32_t incl_mine = 0;
  uint4 pre_v = make_uint4(0, 0, 0, 0);
  uint32_t pre_h = 0;
  bool pre_live = false, pre_valid = false;
  (void)rk0; (void)rk1; (void)ranked; (void)wm_l; (void)wm_base; (void)thr_c; (void)mcode; (void)ta; (void)tb; (void)ta1; (void)tb1; (void)my_i; (void)p_prev; (void)ls_prev;
  (void)tb_mine; (void)incl_mine; (void)pre_v; (void)pre_h; (void)pre_live; (void)pre_valid; (void)lane8; (void)lane16; (void)pend_cap;
  if (MANAGER) PATH_FLAG(A, nd.q, kNib ? kPathNibble : kPathByte);
  if constexpr (MANAGER) {
    __builtin_amdgcn_s_setprio(3);                              // the wave everybody's next step waits for
    mcode = lane < tc ? codes[lane] : 0u;
    BLURRILY_FETCH_TABLE(BLURRILY_STEP_AT(0u), mcode, ta, tb, ta1, tb1);
    BLURRILY_PRODUCE(0u, 0u, BLURRILY_STEP_AT(0u), ta, tb, ta1, tb1);
    BLURRILY_NEXT_VISIT(1u, my_i);
    BLURRILY_FETCH_TABLE(BLURRILY_STEP_AT(my_i), mcode, ta, tb, ta1, tb1);
  }
  // a threshold exists (it is set by compact_pool only, i.e. outside the hot loop below: re-read behind every exit)
  bool have_thr = thr_c != kKeyInf;
  const uint32_t scan_cap = min(tc, ScanTraits<CT>::kMaxCount);  // a counter of this sweep cannot exceed it
  __syncthreads();
  uint2 h_next = ring->hdr[0];                                   // header of the step about to start
  if constexpr (!MANAGER) { tb_mine = BLURRILY_TAB(0u)[lane]; incl_mine = tab_incl(tb_mine); }

  // The sweep is a HOT LOOP of steps that need nothing special -- header, units (the manager: its turn), barrier,
  // scan with the published bound (the manager: the step before's pending candidates), barrier, a glance at the pool --
  // and is left for everything else (more units than the ring holds; no threshold yet: the cold start's bisection;
  // the pool to be compacted, perhaps the step swept again), which is dealt with behind it before the loop is entered
  // again.  Kept apart so that what only the rare paths need is not held in registers, nor worked out, step after step.
  uint32_t e = 0;
  for (;;) {
    uint32_t left, s, p, n_units, hy_;
    for (;; ++e) {
      s = e & 1;
      p = __builtin_amdgcn_readfirstlane(h_next.x);
      hy_ = __builtin_amdgcn_readfirstlane(h_next.y);
      n_units = hy_ & 0xFFFFu;
      if (p >= v1) { left = kLeftDone; break; }                 // no step left
      ++st_steps;
      if constexpr (MANAGER) __builtin_amdgcn_s_setprio(3);     // (the rare paths leave it at 0)
      PHASE_MARK(0);                                            // loop overhead
      const bool tr_ = MANAGER || wid == 0;
      if (tr_) TRACE_MARK(A, nd.q, e, MANAGER ? 1u : 0u, 0u);
      if constexpr (MANAGER) {
        BLURRILY_MANAGER_TURN(e, s);
      } else {
        BLURRILY_COUNT_UNITS(s, n_units, true);
        PHASE_MARK(2);                                          // units counted
      }
      if (tr_) TRACE_MARK(A, nd.q, e, MANAGER ? 1u : 0u, 1u);
      lds_barrier();                                            // counts and next descriptors visible
      if (tr_) TRACE_MARK(A, nd.q, e, MANAGER ? 1u : 0u, 2u);
      PHASE_MARK(3);                                            // barrier after count
      // The next step's header and a worker's units of it were published before that barrier: requested now,
      // they arrive under the scan instead of standing, one LDS round trip each (several hundred clocks behind the
      // other workgroup's atomics), between the scan barrier and the first load of the next step.
      h_next = ring->hdr[s ^ 1u];
      if constexpr (!MANAGER) tb_mine = BLURRILY_TAB(s ^ 1u)[lane];   // (its inclusive sums: BLURRILY_PRELOAD)
      if constexpr (MANAGER) {
        if (can_leave) {                                        // while the workers scan
          BLURRILY_PEND_SETTLE(s ^ 1u, (e - 1u) & 3u, p_prev, ls_prev);
          if (have_thr && n_units != 0 && ((hy_ >> 16) & 0xFFu) != 0) BLURRILY_RANK();   // (the table of the step after the next: fetched before the barrier)
        }
        p_prev = p; ls_prev = n_units ? hy_ >> 24 : 0u;
      }
      if (n_units == 0) {                                       // nothing of the needle in this step's windows
        if constexpr (!MANAGER) incl_mine = tab_incl(tb_mine);
        continue;
      }
      const uint32_t need = (hy_ >> 16) & 0xFFu;
      if (need == 0) { left = kLeftSlowScan; break; }
      if constexpr (!MANAGER) {
        const uint32_t wbase = p * kWPS * kWindowRanks;
        const uint32_t wlen = min(kWPS * kWindowRanks, A.n_refs - wbase);
        scan_core<CT, NT>(cnt128, nd, need, scan_cap, &ctl->thr, &ctl->floor, A.tomb, pool, scan_pool_cap, &ctl->pool_n,
                          &ctl->overflow, wbase, wlen, STATS(A) && A.path_flags ? &A.path_flags[nd.q] : nullptr,
                          hy_ >> 24, pend_list(ring, s, ring_units), &ctl->pend_n[s], pend_cap, uint32_t(NT - 64));
        if (tr_) TRACE_MARK(A, nd.q, e, 0u, 3u);
        BLURRILY_PRELOAD();
      }
      if (tr_) TRACE_MARK(A, nd.q, e, MANAGER ? 1u : 0u, 4u);
      PHASE_MARK(5);                                            // scan
      lds_barrier();                                            // counters are zero again
      if (tr_) TRACE_MARK(A, nd.q, e, MANAGER ? 1u : 0u, 5u);
      PHASE_MARK(6);                                            // barrier after scan
      // (Looking at the pool a count phase later -- the read requested here, used behind the next count barrier, a
      // compaction then running with the next step counted and not yet scanned -- was built and measured in round 3:
      // 2 % slower, 293.0 vs 287.2 ms per 500 k needles; the thresholds the headers carry are a step staler.)
      const uint4 c_ = *reinterpret_cast<const uint4*>(&ctl->pool_n);          // pool_n, overflow, adm_n, (q)
      const uint32_t pn_ = __builtin_amdgcn_readfirstlane(c_.x) + __builtin_amdgcn_readfirstlane(c_.z);
      const uint32_t ov_ = __builtin_amdgcn_readfirstlane(c_.y);
      if (tr_) TRACE_MARK(A, nd.q, e, MANAGER ? 1u : 0u, 6u);
      if (ov_ != 0 || pn_ > sel_at || (!have_thr && pn_ >= A.keep)) { left = kLeftSelect; break; }
    }
    // ---- behind the hot loop: pending candidates first -- the manager's business, all waves wait ----------------
    // Left in front of a count barrier: the step before's have not been looked at yet.  Left behind a scan with the
    // pool or the step's own pending list overflowed: the step is swept again with every slice counted, its pending
    // candidates would come twice and are dropped; not overflowed: they are settled behind the compaction below, which
    // makes room in the pool's tail first.
    const bool own_pending = can_leave && left == kLeftSelect && ctl->overflow == 0;   // (uniform: written behind barriers only)
    if (can_leave) {
      if constexpr (MANAGER) {
        if (left == kLeftDone) BLURRILY_PEND_SETTLE((e & 1u) ^ 1u, (e - 1u) & 3u, p_prev, ls_prev);
        if (left == kLeftSelect && !own_pending && lane == 0) ctl->pend_n[s] = 0;
        // (the step's own pending candidates, where it was not swept again, stay on their list: the next step's scan phase
        // settles them like any step's -- the compaction below empties the pool's tail first, so they have their room.
        // Settling them here, synchronously, was a global round trip with sixteen waves waiting, at every other compaction.)
        p_prev = p; ls_prev = own_pending ? hy_ >> 24 : 0u;
        if (!ranked && have_thr && left != kLeftDone) BLURRILY_RANK();   // (the others are at the barrier: off the next step's count phase)
      }
      __syncthreads();
    }
    if (left == kLeftDone) break;
    // ---- the rare paths of step p ----------------------------------------------------------------
    pre_valid = false;                                          // (a unit loaded ahead is dropped)
    const uint32_t wbase = p * kWPS * kWindowRanks;
    const uint32_t wlen = min(kWPS * kWindowRanks, A.n_refs - wbase);
    // kLeftSelect: the step is scanned, select_after_scan finds the pool as the hot loop saw it; else: scan first
    bool scanned = left == kLeftSelect;
    for (;;) {
      if (!scanned) {
        scan_window<CT, NT>(A, nd, cnt128, pool, ctl, wbase, wlen, scan_pool_cap);
        __syncthreads();
      }
      scanned = false;
      if (!select_after_scan<NT>(A, pool, ctl, wbase, wlen, nd.q, scan_pool_cap)) break;
      ++st_redo;                                                // pool overflow: sweep step p again -- every slice of it
      if ((hy_ >> 24) != 0) BLURRILY_COUNT_WALK(p);             // (the published table lists no units of left-out slices)
      else if constexpr (!MANAGER) BLURRILY_COUNT_UNITS(s, n_units, false);
      __syncthreads();
    }
    thr_c = ctl->thr;
    thr_c = (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(uint32_t(thr_c >> 32))) << 32) |
            __builtin_amdgcn_readfirstlane(uint32_t(thr_c));
    have_thr = thr_c != kKeyInf;
    ++e;
    h_next = ring->hdr[e & 1];                                  // (published behind step p's count barrier)
    if constexpr (!MANAGER) { tb_mine = BLURRILY_TAB(e & 1)[lane]; incl_mine = tab_incl(tb_mine); }
  }
  if constexpr (MANAGER) PHASE_FLUSH_MANAGER(A); else PHASE_FLUSH(A);
  if (STATS(A) && lane == 0) {
    atomicAdd(&STATS(A)[kStatPostingEntries], static_cast<unsigned long long>(st_ent));
    atomicAdd(&STATS(A)[kStatTableWords], static_cast<unsigned long long>(st_tab));
    if (wid == 0) {
      atomicAdd(&STATS(A)[kStatSteps], static_cast<unsigned long long>(st_steps));
      atomicAdd(&STATS(A)[kStatResweeps], static_cast<unsigned long long>(st_redo));
      atomicAdd(&STATS(A)[kStatUnits], static_cast<unsigned long long>(st_walk));   // (needle-major: steps that walked the table)
    }
  }
  (void)st_walk;
  if constexpr (MANAGER) __builtin_amdgcn_s_setprio(0);
  __syncthreads();                                              // ring and ctl quiet before the needle ends
#undef BLURRILY_PEND_SETTLE
#undef BLURRILY_MANAGER_TURN
#undef BLURRILY_COUNT_WALK
#undef BLURRILY_COUNT_UNITS
#undef BLURRILY_PRELOAD
#undef BLURRILY_PRODUCE
#undef BLURRILY_RANK
#undef BLURRILY_RANK_ONE
#undef BLURRILY_LEAVE_OUT
#undef BLURRILY_UNIT_OF
#undef BLURRILY_TAB
#undef BLURRILY_PUBLISH_HDR
#undef BLURRILY_FETCH_TABLE
#undef BLURRILY_NEXT_VISIT
#undef BLURRILY_LOAD_WMT
#undef BLURRILY_STEP_AT
}

// ---- cooperative flavour (needles with <= 64 distinct trigrams: nearly all of them): the workgroup's last wave
// manages the sweep, the others work through it (sweep_role)
template <typename CT, int NT>
__device__ __forceinline__ void sweep_coop(const FindArgs& A, const Needle& nd, const uint16_t* codes, uint32_t* cnt32,
                                           unsigned long long* pool, Control* ctl, UnitRing* ring, const uint8_t* wmt,
                                           const uint32_t w0, const uint32_t w1, const uint32_t ws) {
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == NT / 64 - 1)
    sweep_role<CT, NT, true>(A, nd, codes, cnt32, pool, ctl, ring, wmt, w0, w1, ws);
  else
    sweep_role<CT, NT, false>(A, nd, codes, cnt32, pool, ctl, ring, wmt, w0, w1, ws);
}

// RANGED = latency mode (a needle's windows cut into ranges); a separate instantiation so the
// throughput kernel does not carry the extra live registers.  SHORT = the launch owns only
// needles with <= 64 distinct trigrams (one table slot per lane, which frees the registers for a
// fourth unit in flight); 65..127 follow in a launch over the tokeniser's mid list.
// Residency is set by LDS: two workgroups per CU with byte counters, one with 16-bit counters;
// the second launch bound (waves per SIMD) asks the register allocator for exactly that.
// LEAVE (SHORT launches only) = the sweep may leave dense slices out of a step's count (sweep_coop: a manager wave
// and fifteen workers); without it the round-3 sweep, every wave doing everything (sweep_coop_plain).
template <typename CT, int NT, bool RANGED, bool SHORT, bool LEAVE>
__global__ __launch_bounds__(NT, (sizeof(CT) == 1 ? NT / 128 : NT / 256)) void find_kernel(const FindArgs A) {
  // The counters are a static array: their LDS address is a compile-time constant, so the
  // per-posting ds_add needs no base add.  Dynamic LDS behind them: candidate pool | slice table
  // (long needles) | control.
  constexpr uint32_t kCntBytes = kWindowSize * sizeof(CT);
  __shared__ __attribute__((aligned(16))) uint32_t s_counters[kCntBytes / 4];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* cnt32 = s_counters;
  uint4*    cnt128 = reinterpret_cast<uint4*>(s_counters);
  unsigned long long* pool = reinterpret_cast<unsigned long long*>(smem);
  uint32_t* s_tab = reinterpret_cast<uint32_t*>(pool + A.pool_cap);   // [2][kCodeChunk], long needles only
  Control*  ctl = reinterpret_cast<Control*>(s_tab + 2 * kCodeChunk);
  UnitRing* ring = reinterpret_cast<UnitRing*>(reinterpret_cast<unsigned char*>(ctl) + 64);
  (void)ring;

  const uint32_t tid = threadIdx.x;

  for (uint32_t i = tid; i < kCntBytes / 16; i += NT) cnt128[i] = make_uint4(0, 0, 0, 0);
  // The per-window bound the sweep steps over windows with (win_max_tri) as bytes in LDS, once per workgroup:
  // the wave that chooses the next step reads it in a dependent loop just before the count barrier, and a global
  // round trip there is a wait for all sixteen waves (barrier after count: 1 392 of a step's 6 619 clocks).  It
  // lives in the slice table of the long-needle sweep, which the SHORT instantiations never use; clamped to 255,
  // exact for the <= 64 trigrams of a needle that gets here.
  const uint8_t* wmt = nullptr;
  if constexpr (SHORT) {
    if (A.n_windows <= 2 * kCodeChunk * 4) {
      uint8_t* w8 = reinterpret_cast<uint8_t*>(s_tab);
      for (uint32_t i = tid; i < A.n_windows; i += NT) w8[i] = uint8_t(min(A.win_max_tri[i], 255u));
      wmt = w8;
    }
  }
  (void)wmt;
  __syncthreads();

  const uint32_t n_work = A.n_work_dev ? *A.n_work_dev : A.n_work;

  PHASE_NEEDLE_DECL;
  for (;;) {
    __builtin_amdgcn_s_setprio(kSerialPrio);        // between two needles: one chain of round trips
    if (tid == 0) ctl->q = atomicAdd(A.queue, 1u);
    __syncthreads();
    const uint32_t slot = ctl->q;
    __syncthreads();                                   // everyone has read q before it is rewritten
    if (slot >= n_work) break;
    // latency mode: a needle's windows are cut into `ranges` tasks swept by different workgroups
    const uint32_t R = RANGED ? A.ranges : 1u;
    const uint32_t item = RANGED ? slot / R : slot, range = RANGED ? slot - item * R : 0u;
    const uint32_t q = A.work_list ? A.work_list[item] : item;
    // (ranges start on even windows: the 4-bit sweep takes windows in pairs)
    const uint32_t n_pairs = (A.n_windows + 1) / 2;
    // (own_only: the window pair of the needle's own length class -- phase 1 of the window-major sweep)
    const bool own_only = !RANGED && A.own_only;
    const uint32_t own0 = own_only ? (A.q_start[q] & ~1u) : 0u;
    const uint32_t w0 = RANGED ? 2 * uint32_t(uint64_t(n_pairs) * range / R) : own0;
    const uint32_t w1 = RANGED ? min(A.n_windows, 2 * uint32_t(uint64_t(n_pairs) * (range + 1) / R))
                               : own_only ? min(A.n_windows, own0 + 2) : A.n_windows;
    Needle nd;
    nd.q = q;
    nd.T = A.q_ntri[q];
    if (!A.work_list && nd.T > (SHORT ? 64u : 127u)) { // longer needles: the mid / wide-counter launches
      if (RANGED && tid == 0) A.part_count[slot] = 0;
      continue;
    }
    const uint32_t have = A.pass_base ? A.counts[q] : 0u;
    if (A.q_nb[q] == 0 || A.keep == 0 || have < A.pass_base) {
      if (tid == 0 && A.pass_base == 0) {
        if (RANGED) A.part_count[slot] = 0; else A.counts[q] = 0;
      }
      continue;
    }
    const uint16_t* codes = A.qcodes + A.offsets[q] + q;
    // the sweep starts at the window of the needle's own length class (inside this task's range)
    // -- worth it only when there are enough windows for the early threshold to pay back the
    // ties it forfeits (a window behind the threshold's rank needs one match more)
    const uint32_t qs = A.q_start[q];
    const uint32_t ws = ((w1 - w0 >= 8 || own_only) && qs >= w0 && qs < w1) ? qs : w0;
    // results after this key only (later passes of a limit larger than the pool)
    nd.has_floor = A.pass_base != 0;

    if (tid == 0) {
      ctl->pool_n = 0; ctl->overflow = 0; ctl->thr = kKeyInf;
      ctl->adm_n = 0; ctl->pend_n[0] = 0; ctl->pend_n[1] = 0;
      if (nd.has_floor) ctl->floor = A.floor[q];
    }
    __syncthreads();

    PHASE_NEEDLE(10);
    PATH_FLAG(A, q, (sizeof(CT) == 2 ? kPathWide : 0u) | (RANGED ? kPathRanged : 0u) | (nd.has_floor ? kPathMultiPass : 0u) |
                    (own_only ? kPathOwnOnly : 0u));
    __builtin_amdgcn_s_setprio(0);
    if (STATS(A) && tid == 0) atomicAdd(&STATS(A)[kStatTasks], 1ull);
    // (a macro, not a closure: closures capturing the kernel arguments end up in scratch memory)
#define BLURRILY_SWEEP(a_, b_, start_)                                                                  \
  do {                                                                                                  \
    const uint32_t sa = (a_), sb = (b_), st = (start_);                                                 \
    if constexpr (SHORT) {                                                                              \
      /* 4-bit counters, two windows per step: every window for a needle with <= 15 trigrams, and for   \
         ANY needle the leading windows whose references have <= 15 trigrams (ranks follow weight, i.e. \
         length: about half the windows at Geonames scale) -- a counter there cannot exceed 15 whatever \
         the needle.  The byte-counter part goes first: it holds the needle's own length class. */      \
      const uint32_t nib_end = nd.T <= 15 ? sb : min(sb, max(sa, A.nib_windows));                       \
      if constexpr (LEAVE) {                                                                            \
        if (nib_end < sb)                                                                               \
          sweep_coop<CT, NT>(A, nd, codes, cnt32, pool, ctl, ring, wmt, nib_end, sb, max(st, nib_end));      \
        if (sa < nib_end)                                                                               \
          sweep_coop<Nib, NT>(A, nd, codes, cnt32, pool, ctl, ring, wmt, sa, nib_end, min(st, nib_end - 1)); \
      } else {                                                                                          \
        if (nib_end < sb)                                                                               \
          sweep_coop_plain<CT, NT>(A, nd, codes, cnt32, pool, ctl, ring, wmt, nib_end, sb, max(st, nib_end));      \
        if (sa < nib_end)                                                                               \
          sweep_coop_plain<Nib, NT>(A, nd, codes, cnt32, pool, ctl, ring, wmt, sa, nib_end, min(st, nib_end - 1)); \
      }                                                                                                 \
    } else if constexpr (sizeof(CT) == 1) {      /* byte counters: T <= 127 by construction */          \
      sweep_pipelined<CT, NT>(A, nd, codes, cnt32, pool, ctl, sa, sb, st);                              \
    } else {                                                                                            \
      if (nd.T <= kCodeChunk) sweep_pipelined<CT, NT>(A, nd, codes, cnt32, pool, ctl, sa, sb, st);    \
      else                    sweep_chunked<CT, NT>(A, nd, codes, cnt32, pool, s_tab, ctl, sa, sb);     \
    }                                                                                                   \
  } while (0)
    uint32_t sw_a = 0, sw_b = 0, sw_st = 0;
    (void)sw_a; (void)sw_b; (void)sw_st;
#define BLURRILY_SWEEP_ARGS(a_, b_, st_) do { sw_a = (a_); sw_b = (b_); sw_st = (st_); } while (0)
    if constexpr (RANGED) {
      // A range that does not contain the needle's own length class first sweeps that window
      // only to learn a threshold (the keep-th best of real candidates bounds the answer), then
      // forgets those candidates -- the range that owns the window reports them -- and sweeps
      // its own windows with few admissions instead of a cold start.  (It pays even for a range
      // of one step: without it a single needle takes 88 us instead of 81.)
      // (ONE expansion of the sweep serves both: two of them -- four inlined sweeps -- cost the latency-mode
      // kernel 380 spilled registers)
      const bool learn = qs < A.n_windows && !(qs >= w0 && qs < w1);
      for (uint32_t pass = learn ? 0u : 1u; pass < 2u; ++pass) {
        if (pass == 0) {
          BLURRILY_SWEEP_ARGS(qs, qs + 1, qs);
        } else {
          BLURRILY_SWEEP_ARGS(w0, w1, ws);
        }
        BLURRILY_SWEEP(sw_a, sw_b, sw_st);
        if (pass == 0) {
          compact_pool<NT>(pool, ctl, A.pool_cap, A.keep, SHORT && LEAVE && coop_can_leave(A) ? A.pool_cap - adm_max(A.pool_cap) : A.pool_cap);
          if (tid == 0) ctl->pool_n = 0;
          __syncthreads();
        }
      }
    } else {
      BLURRILY_SWEEP(w0, w1, ws);
    }
#undef BLURRILY_SWEEP_ARGS
#undef BLURRILY_SWEEP
    PHASE_NEEDLE(11);

    // ---- emit: best `keep` in final order; weights are looked up only here ---------------
    compact_pool<NT>(pool, ctl, A.pool_cap, A.keep, SHORT && LEAVE && coop_can_leave(A) ? A.pool_cap - adm_max(A.pool_cap) : A.pool_cap);
    __builtin_amdgcn_s_setprio(kSerialPrio);
    const uint32_t nres = ctl->pool_n;
    if (RANGED) {
      // latency mode: leave this range's best keys for merge_parts_kernel
      for (uint32_t i = tid; i < nres; i += NT) A.part_keys[size_t(slot) * A.keep + i] = pool[i];
      if (tid == 0) A.part_count[slot] = nres;
      __syncthreads();
      continue;
    }
    if (own_only) {
      // the needle's state for wsweep_kernel: its best keys so far, two words each, where its rows will be
      uint32_t* st = reinterpret_cast<uint32_t*>(A.results + size_t(q) * A.limit);
      for (uint32_t i = tid; i < nres; i += NT) {
        const unsigned long long key = pool[i];
        st[2 * i] = uint32_t(key); st[2 * i + 1] = uint32_t(key >> 32);
      }
      if (tid == 0) A.counts[q] = nres;
      __syncthreads();
      continue;
    }
    trigram_match_t* out = A.results + size_t(q) * A.limit + A.pass_base;
    for (uint32_t i = tid; i < nres; i += NT) {
      const unsigned long long key = pool[i];
      const uint32_t rk = uint32_t(key);
      trigram_match_t r;
      r.reference = A.ref_of_rank[rk];
      r.matches = nd.T - uint32_t(key >> 32);
      r.weight = A.weight_of_rank[rk];
      out[i] = r;
    }
    if (tid == 0) {
      A.counts[q] = A.pass_base + nres;
      if (A.floor && nres > 0) A.floor[q] = pool[nres - 1];
    }
    __syncthreads();                                   // pool reads done before the next needle resets it
    PHASE_NEEDLE(12);
  }
}

// ---- latency mode, second half: merge the per-range candidates of every needle -------------
// One workgroup per needle: the best `keep` of the union of the per-range best `keep` keys is
// the needle's result (every range keeps its own top `keep`, so nothing can be lost).
template <int NT>
__global__ __launch_bounds__(NT) void merge_parts_kernel(const FindArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* pool = reinterpret_cast<unsigned long long*>(smem);
  Control* ctl = reinterpret_cast<Control*>(pool + A.pool_cap);
  const uint32_t tid = threadIdx.x;
  const uint32_t item = blockIdx.x, R = A.ranges;
  const uint32_t q = A.work_list ? A.work_list[item] : item;
  if (tid == 0) { ctl->pool_n = 0; ctl->overflow = 0; ctl->thr = kKeyInf; ctl->adm_n = 0; }
  __syncthreads();
  for (uint32_t r = 0; r < R; ++r) {
    const uint32_t slot = item * R + r;
    const uint32_t n = A.part_count[slot];
    if (tid < n) pool[atomicAdd(&ctl->pool_n, 1u)] = A.part_keys[size_t(slot) * A.keep + tid];
  }
  __syncthreads();
  compact_pool<NT>(pool, ctl, A.pool_cap, A.keep);
  const uint32_t nres = ctl->pool_n;
  const uint32_t T = A.q_ntri[q];
  trigram_match_t* out = A.results + size_t(q) * A.limit;
  for (uint32_t i = tid; i < nres; i += NT) {
    const unsigned long long key = pool[i];
    const uint32_t rk = uint32_t(key);
    trigram_match_t row;
    row.reference = A.ref_of_rank[rk];
    row.matches = T - uint32_t(key >> 32);
    row.weight = A.weight_of_rank[rk];
    out[i] = row;
  }
  if (tid == 0) A.counts[q] = nres;
}

// Small limits (<= 64 rows, <= 256 ranges): the per-range lists are already sorted, so one wave
// merges them by repeated minimum extraction -- every lane watches the heads of up to four
// lists, a wave-wide minimum picks the winner, the winning lane advances -- instead of sorting
// all ranges x keep keys (1 290 keys for a single needle at Geonames scale).
__global__ __launch_bounds__(64) void merge_parts_small_kernel(const FindArgs A) {
  __shared__ unsigned long long s_win[64];
  const uint32_t lane = threadIdx.x;
  const uint32_t item = blockIdx.x, R = A.ranges, keep = A.keep;
  const uint32_t q = A.work_list ? A.work_list[item] : item;
  const uint32_t s0 = item * R + lane, s1 = s0 + 64, s2 = s0 + 128, s3 = s0 + 192;
  const uint32_t n0 = lane < R ? A.part_count[s0] : 0u, n1 = lane + 64 < R ? A.part_count[s1] : 0u;
  const uint32_t n2 = lane + 128 < R ? A.part_count[s2] : 0u, n3 = lane + 192 < R ? A.part_count[s3] : 0u;
  uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
  auto head = [&](uint32_t slot, uint32_t p, uint32_t n) {
    return p < n ? A.part_keys[size_t(slot) * keep + p] : kKeyInf;
  };
  unsigned long long h0 = head(s0, 0, n0), h1 = head(s1, 0, n1), h2 = head(s2, 0, n2), h3 = head(s3, 0, n3);
  uint32_t nres = 0;
  for (uint32_t k = 0; k < keep; ++k) {
    const unsigned long long mine = min(min(h0, h1), min(h2, h3));
    unsigned long long best = mine;
#pragma unroll
    for (int d = 32; d; d >>= 1) {
      const uint32_t lo = __shfl_xor(uint32_t(best), d), hi = __shfl_xor(uint32_t(best >> 32), d);
      const unsigned long long other = (static_cast<unsigned long long>(hi) << 32) | lo;
      best = min(best, other);
    }
    if (best == kKeyInf) break;
    if (mine == best) {                      // keys are distinct (distinct ranks): exactly one lane, one list
      if (h0 == best)      h0 = head(s0, ++p0, n0);
      else if (h1 == best) h1 = head(s1, ++p1, n1);
      else if (h2 == best) h2 = head(s2, ++p2, n2);
      else                 h3 = head(s3, ++p3, n3);
    }
    if (lane == 0) s_win[k] = best;
    ++nres;
  }
  __syncthreads();
  const uint32_t T = A.q_ntri[q];
  trigram_match_t* out = A.results + size_t(q) * A.limit;
  if (lane < nres) {
    const unsigned long long key = s_win[lane];
    const uint32_t rk = uint32_t(key);
    trigram_match_t row;
    row.reference = A.ref_of_rank[rk];
    row.matches = T - uint32_t(key >> 32);
    row.weight = A.weight_of_rank[rk];
    out[lane] = row;
  }
  if (lane == 0) A.counts[q] = nres;
}

// ---- base + delta: merge two per-needle result lists in result order ----------------------
// (matches descending, weight ascending, reference ascending; the two images hold disjoint
// references, so the first `limit` rows of the merge are the answer).  One lane per needle.
__global__ void merge_rows_kernel(const trigram_match_t* __restrict__ a_rows, const uint32_t* __restrict__ a_counts,
                                  const trigram_match_t* __restrict__ b_rows, const uint32_t* __restrict__ b_counts,
                                  uint32_t n, uint32_t limit, trigram_match_t* __restrict__ out,
                                  uint32_t* __restrict__ out_counts) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const trigram_match_t* a = a_rows + size_t(q) * limit;
  const trigram_match_t* b = b_rows + size_t(q) * limit;
  trigram_match_t* o = out + size_t(q) * limit;
  const uint32_t na = a_counts[q], nb = b_counts[q];
  uint32_t i = 0, j = 0, k = 0;
  while (k < limit && (i < na || j < nb)) {
    bool take_a;
    if (i >= na) take_a = false;
    else if (j >= nb) take_a = true;
    else {
      const trigram_match_t x = a[i], y = b[j];
      if (x.matches != y.matches) take_a = x.matches > y.matches;
      else if (x.weight != y.weight) take_a = x.weight < y.weight;
      else take_a = x.reference < y.reference;
    }
    o[k++] = take_a ? a[i++] : b[j++];
  }
  out_counts[q] = k;
}

// ============================================================ window-major sweep ============
// The needle-major sweep above streams, for every needle, every posting of every one of its
// trigrams: at Geonames scale 1.8 M postings per needle, 3.6 MB from HBM, most of them from the few
// DENSE slices of the needle (a common trigram holds several per cent of a window's references).
// Once a needle has a threshold -- `need` matches to enter its top `keep` -- those slices need not
// be counted at all (the classic MaxScore argument, exact here):
//
//   leave out the L largest dense slices of the needle in this window, L <= need - cmin.  A
//   reference with m >= need matches has at least need - L >= cmin >= 1 of them among the slices
//   that ARE counted ("cold" count), so the scan finds it with the lower threshold need - L; for
//   each such candidate the L left-out slices are consulted through their BITMAPS (one bit each,
//   device_index.h), which gives its exact match count, and the usual key test admits it or not.
//
// That touches 10-25 % of the postings.  The bitmap probes are random 4-byte reads, affordable
// only from L2, so the sweep runs WINDOW-MAJOR: one launch per window, every workgroup of the chip
// working on the same window for different needles, the window's slice table, postings and bitmaps
// (a few MB) resident in every XCD's L2 and read from HBM once per launch.  A needle's state -- its
// best keys so far, hence its threshold -- lives in global memory between launches (in the needle's
// own result rows, two words per key); phase 1 (find_kernel with own_only) seeds it from the
// window pair of the needle's own length class, where its best matches live; finalize_rows_kernel
// turns keys into rows.
//
// One task = (needle, window), run by a workgroup of 4 waves with 32 KiB of counters -- four
// workgroups per CU: 4-bit counters for the whole window when a cold count cannot exceed 15 (nearly
// always: the cold slices are few), else byte counters over the two halves of the window in turn.
constexpr int      kWsNT    = 256;             // (512 -- eight waves per task and SIMD, 64 VGPRs, 100 B of scratch -- measured:
                                               //  configs[2] 322 -> 346 ms per 300 k needles, configs[4] 82.3 -> 85.6 ms)
constexpr uint32_t kWsNW    = kWsNT / 64;
constexpr uint32_t kWsChunk = 64;              // most needles per queue pop (the task arrays in LDS; with 256 of them
                                               // the per-task publication slots below would cost the fourth workgroup per CU)
constexpr uint32_t kWsAhead = 4;               // units a wave loads before it counts the first of them
constexpr uint32_t kWsCand  = 512;             // candidate list (rank | cold << 16): two per thread
constexpr uint32_t kWsPool  = 256;             // candidate pool, >= 2 * kWsMaxKeep
constexpr uint32_t kWsCntWords = kWindowSize / 8;   // 8192 words = 32 KiB: 65 536 nibbles or 32 768 bytes
static_assert(kWsPool >= 2 * kWsMaxKeep && kWsPool <= uint32_t(kWsNT), "compact_pool's rank sort needs pool <= threads");
static_assert(kWsCand <= 2 * kWsNT, "the probe phase gives every thread at most two candidates");

struct WsControl {
  Control  c;
  uint32_t n_cand;
  uint32_t cand_ov;           // the candidate list overflowed in this pass
  uint32_t n_tasks;
  uint32_t chunk;
  // published by a task's owner wave with s_units[slot] / s_hot[slot], slot = the task's place in its group
  uint32_t pub_units[kWsNW], pub_L[kWsNW], pub_wide[kWsNW];
};

__device__ __forceinline__ unsigned long long ws_load_key(const FindArgs& A, uint32_t q, uint32_t i) {
  const uint32_t* st = reinterpret_cast<const uint32_t*>(A.results + size_t(q) * A.limit);
  return (static_cast<unsigned long long>(st[2 * i + 1]) << 32) | st[2 * i];
}

// The single-window 4-bit layout.  The index deals the postings of a unit so that one
// LDS instruction sees each BYTE-counter bank once (bank = (r >> 2) & 31, device_index.hip); the 4-bit
// layout keeps that bank: rank r -> word (r >> 2) & 0x1FFF, i.e. byte address r & 0x7FFC, and nibble
// (r & 3) | (r >> 15) << 2 -- a word holds ranks 4w..4w+3 of the window's lower half in its low nibbles
// and ranks 32768 + 4w.. of the upper half in its high ones.  (With word = r >> 3, 61 % of the LDS-active
// cycles were bank conflicts, profiles/r02_ws_pmc.txt.)
// Both ranks of a packed dword into that layout, 5 + 6 VALU instead of 8 + 8: the shift amount is built so that
// its junk sits above bit 4, which the shifter does not read (v_lshlrev_b32 takes five bits of the amount).
//   low half  r = v & 0xFFFF:  x = v & 0x8003 keeps r's bits 1:0 and 15;  (x << 2) + (x >> 11) = 4 (r & 3) + 16 (r >> 15) + 2^17 (r >> 15)
//   high half r = v >> 16:     y = v & 0x80030000;  (y >> 14) | (y >> 27) = 4 (r & 3) + 16 (r >> 15) + 2^17 (r >> 15)
__device__ __forceinline__ uint32_t one_shl_low5(uint32_t amount) {
  uint32_t out;
  asm("v_lshlrev_b32 %0, %1, 1" : "=v"(out) : "v"(amount));
  return out;
}
__device__ __forceinline__ void ws_bump_pair_nib(uint32_t* cnt32, uint32_t v) {
  unsigned char* const base = reinterpret_cast<unsigned char*>(cnt32);
  const uint32_t x = v & 0x8003u;
  __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(base + (v & 0x7FFCu)), one_shl_low5((x << 2) + (x >> 11)),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const uint32_t y = v & 0x80030000u;
  __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(base + ((v >> 16) & 0x7FFCu)), one_shl_low5((y >> 14) | (y >> 27)),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// The same with the counter's value BEFORE the increment coming back: a counter that was at need - 1 has just
// reached `need` -- its rank goes to the candidate list at once (a counter crosses a value exactly once: a slice
// holds a reference once, counts only grow), so that the window's counters need not be SCANNED for candidates
// afterwards, only cleared.  (v_bfe_u32 reads five bits of its offset, like the shifter: the same junk-tolerant
// amounts serve.)  Padding (rank 0xFFFF) and ranks past the window's end never become candidates.
__device__ __forceinline__ uint32_t bfe4_low5(uint32_t value, uint32_t offset) {
  uint32_t out;
  asm("v_bfe_u32 %0, %1, %2, 4" : "=v"(out) : "v"(value), "v"(offset));
  return out;
}
__device__ __forceinline__ void ws_push_cand(uint32_t* cand, uint32_t* n_cand, uint32_t* cand_ov, uint32_t r16) {
  const uint32_t at = atomicAdd(n_cand, 1u);
  if (at < kWsCand) cand[at] = r16; else *cand_ov = 1;
}
__device__ __forceinline__ void ws_bump8_cross(uint32_t* cnt32, const uint4 v, uint32_t need, uint32_t wlen,
                                               uint32_t* cand, uint32_t* n_cand, uint32_t* cand_ov) {
  if (!group_live(v)) return;
  unsigned char* const base = reinterpret_cast<unsigned char*>(cnt32);
  const uint32_t d[4] = {v.x, v.y, v.z, v.w};
  uint32_t old_lo[4], old_hi[4], sh_lo[4], sh_hi[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {                              // eight returning atomics in flight
    const uint32_t x = d[j] & 0x8003u, y = d[j] & 0x80030000u;
    sh_lo[j] = (x << 2) + (x >> 11);
    sh_hi[j] = (y >> 14) | (y >> 27);
    old_lo[j] = __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(base + (d[j] & 0x7FFCu)), one_shl_low5(sh_lo[j]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old_hi[j] = __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(base + ((d[j] >> 16) & 0x7FFCu)), one_shl_low5(sh_hi[j]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  const uint32_t was = need - 1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (bfe4_low5(old_lo[j], sh_lo[j]) == was && (d[j] & 0xFFFFu) < wlen) ws_push_cand(cand, n_cand, cand_ov, d[j] & 0xFFFFu);
    if (bfe4_low5(old_hi[j], sh_hi[j]) == was && (d[j] >> 16) < wlen) ws_push_cand(cand, n_cand, cand_ov, d[j] >> 16);
  }
}
// The same returning atomics for a COLD window (no threshold yet): a counter passes every value once, so the number of
// increments that produce the value m IS the number of counters that reach m -- counted here, in hist[m] (m >= 2: one
// LDS atomic for the few postings that find their counter already at work), the cold start needs no pass over the
// counters at all to know the largest bound that `keep` of them reach.  Padding (rank 0xFFFF) is not a counter.
__device__ __forceinline__ void ws_bump8_hist(uint32_t* cnt32, const uint4 v, uint32_t* hist) {
  if (!group_live(v)) return;
  unsigned char* const base = reinterpret_cast<unsigned char*>(cnt32);
  const uint32_t d[4] = {v.x, v.y, v.z, v.w};
  uint32_t old_lo[4], old_hi[4], sh_lo[4], sh_hi[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {                              // eight returning atomics in flight
    const uint32_t x = d[j] & 0x8003u, y = d[j] & 0x80030000u;
    sh_lo[j] = (x << 2) + (x >> 11);
    sh_hi[j] = (y >> 14) | (y >> 27);
    old_lo[j] = __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(base + (d[j] & 0x7FFCu)), one_shl_low5(sh_lo[j]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old_hi[j] = __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(base + ((d[j] >> 16) & 0x7FFCu)), one_shl_low5(sh_hi[j]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t was_lo = bfe4_low5(old_lo[j], sh_lo[j]), was_hi = bfe4_low5(old_hi[j], sh_hi[j]);
    if (was_lo != 0 && (d[j] & 0xFFFFu) != kPadRank) atomicAdd(&hist[was_lo + 1], 1u);
    if (was_hi != 0 && (d[j] >> 16) != kPadRank) atomicAdd(&hist[was_hi + 1], 1u);
  }
}
// one posting into the half-window byte layout: rank r of half `h` -> byte r & 0x7FFF; a posting of the
// other half goes to a dump word behind the counters
__device__ __forceinline__ void ws_bump_byte(uint32_t* cnt32, uint32_t r, uint32_t h) {
  const uint32_t word = (r >> 15) == h ? (r & 0x7FFFu) >> 2 : kWsCntWords;
  __hip_atomic_fetch_add(&cnt32[word], 1u << ((r << 3) & 24u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <bool WIDE>
__device__ __forceinline__ void ws_bump8(uint32_t* cnt32, const uint4 v, uint32_t h) {
  if (!group_live(v)) return;
  const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (WIDE) { ws_bump_byte(cnt32, d[j] & 0xFFFFu, h); ws_bump_byte(cnt32, d[j] >> 16, h); }
    else      { ws_bump_pair_nib(cnt32, d[j]); }
  }
}

// A workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global load
// and store in flight (s_waitcnt vmcnt(0)), which would turn each of the kernel's barriers into a full
// memory round trip and defeat the prefetches that are meant to travel across them; wsweep_kernel
// exchanges data between its waves through LDS only.
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// compact_pool for wsweep_kernel (pool <= 256 keys = threads).  The first `n_sorted` keys are in order
// (the needle's state, or what an earlier compaction left) and only the few keys admitted since then
// follow them, so a key finds its place by MERGING: a sorted key is its index plus the new keys below it,
// a new key is a binary search over the sorted ones plus the new keys below it -- a handful of steps
// instead of one comparison per key of the pool (limit 100: ~100 per thread).  Keys are distinct.
__device__ __forceinline__ void ws_compact_pool(unsigned long long* pool, Control* ctl, uint32_t keep, uint32_t n_sorted) {
  const uint32_t tid = threadIdx.x;
  const uint32_t n = min(ctl->pool_n, kWsPool);
  n_sorted = min(n_sorted, n);
  const unsigned long long mine = tid < n ? pool[tid] : kKeyInf;
  uint32_t below = 0;
  if (tid < n) {
    uint32_t j = n_sorted;                                              // (same address in every lane: broadcasts)
    if (j & 1u) { if (j < n) below += pool[j] < mine; ++j; }
#pragma unroll 4
    for (; j + 2 <= n; j += 2) {                                        // two keys per read (four reads in flight)
      const ulonglong2 two = *reinterpret_cast<const ulonglong2*>(pool + j);
      below += uint32_t(two.x < mine) + uint32_t(two.y < mine);
    }
    if (j < n) below += pool[j] < mine;
    if (tid < n_sorted) {
      below += tid;
    } else {
      uint32_t lo = 0, hi = n_sorted;                                   // first sorted key not below `mine`
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pool[mid] < mine) lo = mid + 1; else hi = mid;
      }
      below += lo;
    }
  }
  ws_barrier();
  if (tid < n && below < keep) pool[below] = mine;
  ws_barrier();
  if (tid == 0) {
    ctl->pool_n = min(n, keep);
    ctl->overflow = 0;
    if (n >= keep && keep > 0) ctl->thr = pool[keep - 1];
  }
  ws_barrier();
}

// What the scans need to know about the two counter layouts of wsweep_kernel.
//   WIDE = false: 4-bit counters, the whole window: word w holds ranks 4w..4w+3 (nibbles 0..3) and
//                 32768 + 4w.. (nibbles 4..7), see ws_bump_pair_nib; vector i = words 4i..4i+3
//   WIDE = true : byte counters of half h of the window: vector i holds ranks [32768 h + 16 i, ... + 16)
template <bool WIDE> struct WsLayout {
  static constexpr uint32_t kRanksPerVec = WIDE ? 16 : 32, kBits = WIDE ? 8 : 4, kPerWord = WIDE ? 4 : 8;
  static constexpr uint32_t kLastVec = kWsCntWords / 4 - 1;
  uint32_t bias; bool low; uint32_t lo_r, nv; bool pad_here;
  uint32_t pre;               // a counter >= need has one of these bits set (need >= 2^k: bits k.. of its field)
  __device__ __forceinline__ WsLayout(uint32_t need, uint32_t h, uint32_t wlen) {
    const uint32_t top = 31u - __clz(max(need, 1u));       // floor(log2(need))
    pre = (WIDE ? 0x01010101u * ((0xFFu << top) & 0xFFu) : 0x11111111u * ((0xFu << top) & 0xFu));
    if (WIDE) {
      bias = (0x80u - min(need, 0x7Fu)) * 0x01010101u; low = false;
      lo_r = h * 32768u;
      const uint32_t span = wlen > lo_r ? min(32768u, wlen - lo_r) : 0u;
      nv = (span + 15) / 16;
      pad_here = h == 1;                                   // slot 0xFFFF: last byte of half 1
    } else {
      low = need <= 8;                                     // counter >= need, c = counter, lo = c & 7:
      bias = (low ? 8 - need : 16 - need) * 0x11111111u;   //   need <= 8: c >= 8 or lo + (8 - need) >= 8
      lo_r = 0; nv = (min(wlen, 32768u) + 15) / 16; pad_here = true;   // need > 8: c >= 8 and lo + (16 - need) >= 8
    }
  }
  // one bit per counter that reached `need`, at the top bit of its field
  __device__ __forceinline__ uint32_t hits(uint32_t d) const {
    if (WIDE) return (d + bias) & 0x80808080u;
    const uint32_t t = (d & 0x77777777u) + bias;
    return (low ? (t | d) : (t & d)) & 0x88888888u;
  }
  __device__ __forceinline__ uint4 mask_pad(uint4 v, uint32_t i) const {
    if (pad_here && i == kLastVec) v.w &= WIDE ? 0x00FFFFFFu : 0x0FFFFFFFu;    // slot 0xFFFF counts padding
    return v;
  }
  __device__ __forceinline__ uint32_t rank16(uint32_t i, uint32_t j, uint32_t bit) const {
    if (WIDE) return lo_r + i * 16 + j * 4 + bit / 8;
    const uint32_t nib = bit / 4;                          // nibbles 0..3: lower half of the window, 4..7: upper half
    return (i * 4 + j) * 4 + (nib & 3u) + (nib >> 2) * 32768u;
  }
  __device__ __forceinline__ uint32_t count(uint32_t d, uint32_t bit) const {
    return (d >> (bit / kBits * kBits)) & ((1u << kBits) - 1u);
  }
};

// candidate (in-window rank, exact match count) -> key -> pool
__device__ __forceinline__ void ws_admit(const FindArgs& A, Control* ctl, unsigned long long* pool, unsigned long long thr,
                                         uint32_t T, uint32_t wbase, uint32_t r16, uint32_t total) {
  const uint32_t rank = wbase + r16;
  const unsigned long long key = (static_cast<unsigned long long>(T - total) << 32) | rank;
  bool pass = key <= thr;
  if (A.tomb) pass = pass && ((A.tomb[rank >> 5] >> (rank & 31)) & 1u) == 0;
  if (pass) {
    const uint32_t at = atomicAdd(&ctl->pool_n, 1u);
    if (at < kWsPool) pool[at] = key; else ctl->overflow = 1;
  }
}

// Fast scan: counters that reach `need` go to the candidate list as (rank | count << 16); everything
// read is cleared.  A thread's eight vectors are read back to back into registers (one LDS round trip,
// not eight) and cleared; its hits are counted (one SWAR test per word), ONE atomic reserves their
// places in the list, and a pass over the registers writes them -- instead of one returning atomic per
// hit, which the scan would wait on hit after hit.
template <bool WIDE>
__device__ __forceinline__ void ws_scan_fast(uint4* cnt128, uint32_t* cand, uint32_t* n_cand, uint32_t* cand_ov,
                                             uint32_t need, uint32_t h, uint32_t wlen) {
  const WsLayout<WIDE> Y(need, h, wlen);
  const uint32_t tid = threadIdx.x;
  constexpr uint32_t kPerThread = kWsCntWords / 4 / kWsNT;
  constexpr uint32_t kRound = kPerThread < 4 ? kPerThread : 4;   // vectors of a thread in flight together (all eight -- 51 spilled registers -- measured: 75.8 -> 84.4 ms)
#pragma unroll 1
  for (uint32_t k0 = 0; k0 < kWsCntWords / 4 / kWsNT; k0 += kRound) {
    uint4 v[kRound];
#pragma unroll
    for (uint32_t k = 0; k < kRound; ++k) {
      const uint32_t i = tid + (k0 + k) * kWsNT;
      v[k] = make_uint4(0, 0, 0, 0);
      if (i < Y.nv) v[k] = cnt128[i];
    }
#pragma unroll
    for (uint32_t k = 0; k < kRound; ++k) {
      const uint32_t i = tid + (k0 + k) * kWsNT;
      uint32_t z = 0;
      asm volatile("" : "+v"(z));
      if (i < Y.nv) cnt128[i] = make_uint4(z, z, z, z);
    }
    v[kRound - 1] = Y.mask_pad(v[kRound - 1], tid + (k0 + kRound - 1) * kWsNT);
    // most counters are zero and nearly all are below `need`: one AND per word tells whether any counter
    // of the round can reach it at all, before the exact SWAR test
    uint32_t any = 0;
#pragma unroll
    for (uint32_t k = 0; k < kRound; ++k) any |= (v[k].x | v[k].y | v[k].z | v[k].w) & Y.pre;
    if (any == 0) continue;
    uint32_t n_hit = 0;
#pragma unroll
    for (uint32_t k = 0; k < kRound; ++k)
      n_hit += __popc(Y.hits(v[k].x)) + __popc(Y.hits(v[k].y)) + __popc(Y.hits(v[k].z)) + __popc(Y.hits(v[k].w));
    if (n_hit == 0) continue;
    uint32_t at = atomicAdd(n_cand, n_hit);
    if (at + n_hit > kWsCand) { *cand_ov = 1; continue; }  // the window is swept again, the robust way
#pragma unroll
    for (uint32_t k = 0; k < kRound; ++k) {
      const uint32_t d[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
      if ((d[0] | d[1] | d[2] | d[3]) == 0) continue;
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j) {
        uint32_t m = Y.hits(d[j]);
        while (m) {
          const uint32_t bit = __ffs(m) - 1;
          m &= m - 1;
          cand[at++] = Y.rank16(tid + (k0 + k) * kWsNT, j, bit) | (Y.count(d[j], bit) << 16);
        }
      }
    }
  }
}

// counters of this thread's vectors that reach `need` (nothing is cleared): the cold-start bisection
template <bool WIDE>
__device__ __forceinline__ uint32_t ws_count_hits(const uint4* cnt128, uint32_t need, uint32_t h, uint32_t wlen) {
  const WsLayout<WIDE> Y(need, h, wlen);
  uint32_t mine = 0;
  for (uint32_t i = threadIdx.x; i < Y.nv; i += kWsNT) {
    const uint4 v = Y.mask_pad(cnt128[i], i);
    mine += __popc(Y.hits(v.x)) + __popc(Y.hits(v.y)) + __popc(Y.hits(v.z)) + __popc(Y.hits(v.w));
  }
  return mine;
}

// Robust scan (no threshold yet, or a candidate list that overflowed): nothing is left out of the count,
// so a counter IS the match count and a hit goes straight to the pool if its key beats the threshold --
// only such hits take room, so a flood of ties behind the threshold's rank cannot fill anything.
template <bool WIDE>
__device__ __forceinline__ void ws_scan_robust(const FindArgs& A, uint4* cnt128, Control* ctl, unsigned long long* pool,
                                               unsigned long long thr, uint32_t T, uint32_t wbase, uint32_t need,
                                               uint32_t h, uint32_t wlen) {
  const WsLayout<WIDE> Y(need, h, wlen);
#pragma unroll 1
  for (uint32_t i = threadIdx.x; i < Y.nv; i += kWsNT) {
    const uint4 v = Y.mask_pad(cnt128[i], i);
    uint32_t z = 0;
    asm volatile("" : "+v"(z));
    cnt128[i] = make_uint4(z, z, z, z);
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll 1
    for (uint32_t j = 0; j < 4; ++j) {
      uint32_t m = Y.hits(d[j]);
      while (m) {
        const uint32_t bit = __ffs(m) - 1;
        m &= m - 1;
        ws_admit(A, ctl, pool, thr, T, wbase, Y.rank16(i, j, bit), Y.count(d[j], bit));
      }
    }
  }
}

// phase clocks of wave 0 (stats mode only): FindArgs::stats[kStatWsClocks + phase]
#define WS_CLOCK(i) do { if (STATS(A) && wid == 0) { const unsigned long long t_ = clock64(); ws_clk[i] += t_ - ws_last; ws_last = t_; } } while (0)

__global__ __launch_bounds__(kWsNT, kWsNT / 64) void wsweep_kernel(const FindArgs A, const uint32_t w, const uint32_t n,
                                                          const uint32_t chunk_len, const uint32_t own_pass) {
  __shared__ __attribute__((aligned(16))) uint32_t s_cnt[kWsCntWords + 4];       // + dump word
  __shared__ uint32_t s_cand[kWsCand];
  __shared__ unsigned long long s_pool[kWsPool];
  __shared__ uint32_t s_task_q[kWsChunk], s_task_meta[kWsChunk], s_task_code[kWsChunk];
  __shared__ uint2 s_units[kWsNW][64];          // what a task's owner wave publishes: the units to count (first entry, slice end)
  __shared__ uint32_t s_hot[kWsNW][64];         //   ... and the bitmaps of the slices it left out
  __shared__ WsControl s_ctl;
  uint4* cnt128 = reinterpret_cast<uint4*>(s_cnt);
  Control* ctl = &s_ctl.c;

  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t wbase = w * kWindowRanks;
  const uint32_t wlen = min(kWindowRanks, A.n_refs - wbase);
  const uint32_t wmt = A.win_max_tri[w];
  const uint32_t keep = A.keep;
  const uint2* soff = A.slice_se + size_t(w) * kNumCodes;

  for (uint32_t i = tid; i < (kWsCntWords + 4) / 4; i += kWsNT) cnt128[i] = make_uint4(0, 0, 0, 0);
  // request counters (FindArgs::stats), per wave
  uint32_t st_ent = 0, st_probe = 0, st_tab = 0, st_steps = 0, st_redo = 0, st_tasks = 0, st_compact = 0, st_units = 0;
  unsigned long long ws_clk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ws_last = clock64();

  for (;;) {
    if (tid == 0) { s_ctl.chunk = atomicAdd(A.queue, 1u); s_ctl.n_tasks = 0; s_ctl.n_cand = 0; s_ctl.cand_ov = 0; }
    ws_barrier();
    const uint32_t base = s_ctl.chunk * chunk_len;           // chunk_len <= kWsNT needles per queue pop
    if (base >= n) break;
    // ---- which of these needles have anything to gain from this window? ----------------------
    {
      const uint32_t q = base + tid;
      bool live = q < n && tid < chunk_len;
      uint32_t T = 0, cnt = 0, need = 0;
      if (live) { T = A.q_ntri[q]; live = T <= 64 && A.q_nb[q] != 0; }
      // own_pass: only the needles whose own length class lives in this window pair (their first two tasks,
      // which give them a threshold); otherwise everybody else
      if (live) { const uint32_t own0 = A.q_start[q] & ~1u; live = (w >= own0 && w < own0 + 2) == (own_pass != 0); }
      if (live) {
        cnt = A.counts[q];
        const unsigned long long thr = cnt >= keep ? ws_load_key(A, q, keep - 1) : kKeyInf;
        need = matches_needed(thr, T, wbase);
        live = min(T, wmt) >= need;
      }
      if (live) {
        const uint32_t at = atomicAdd(&s_ctl.n_tasks, 1u);
        s_task_q[at] = q;
        s_task_meta[at] = T | (cnt << 8) | (need << 16);     // (cnt <= keep <= 128, need <= 65: a byte each)
        s_task_code[at] = uint32_t(A.offsets[q]) + q;        // the needle's codes in qcodes (the host checked 32 bits)
      }
    }
    ws_barrier();
    const uint32_t n_tasks = s_ctl.n_tasks;
    WS_CLOCK(0);

    // Tasks go in groups of four.  Wave j OWNS task g + j of the group: it alone holds the needle's slice
    // table for this window (start, end, bitmap of every trigram's slice, one per lane), chooses the slices
    // to leave out and lists the units to count; the other waves only ever see what it publishes through
    // LDS.  The four tables of a group are fetched by the four waves side by side, those of the next group
    // while this one is being worked on.
    uint32_t nx_code = 0, nx_ta = 0, nx_tb = 0, nx_bm = kNoBitmap;
#define WS_FETCH_CODES(ti_)                                                               \
  do {                                                                                    \
    nx_code = 0;                                                                          \
    if ((ti_) < n_tasks && lane < (s_task_meta[ti_] & 0xFFu)) nx_code = A.qcodes[s_task_code[ti_] + lane]; \
  } while (0)
#define WS_FETCH_TABLE(ti_)                                                               \
  do {                                                                                    \
    nx_ta = nx_tb = 0; nx_bm = kNoBitmap;                                                 \
    if ((ti_) < n_tasks && lane < (s_task_meta[ti_] & 0xFFu)) {                           \
      const uint2 se_ = soff[nx_code];                                                    \
      nx_ta = se_.x; nx_tb = se_.y;                                                       \
      nx_bm = nx_tb - nx_ta >= A.dense_min8 ? nx_ta : kNoBitmap;   /* a dense slice: its bitmap sits in front of its postings */ \
    }                                                                                     \
  } while (0)
    // What a task's owner publishes for one pass over the window: which dense slices are left out (the L largest,
    // L <= need - cmin), which units are to be counted; into slot `slot_` of s_units / s_hot / s_ctl.pub_*.
#define WS_PUBLISH(slot_, T_, need_, robust_)                                                   \
  do {                                                                                          \
    const uint32_t l_max = (!(robust_) && (need_) > A.cmin) ? (need_) - A.cmin : 0u;            \
    const uint32_t size = tb0 - ta;                                                             \
    const bool dense = own && bm != kNoBitmap && size > 0;                                      \
    uint32_t bigger = 0;                                                                        \
    if (l_max)                                                                                  \
      for (unsigned long long m = __ballot(dense); m; m &= m - 1) {                             \
        const uint32_t u = __builtin_ctzll(m);                                                  \
        const uint32_t su = __builtin_amdgcn_readlane(size, u);                                 \
        bigger += (su > size || (su == size && u < lane)) ? 1u : 0u;                            \
      }                                                                                         \
    const bool skip = dense && bigger < l_max;                                                  \
    const unsigned long long skipmask = __ballot(skip);                                         \
    const uint32_t L_ = __popcll(skipmask);                                                     \
    if (skip) s_hot[slot_][__popcll(skipmask & ((1ull << lane) - 1ull))] = bm;                  \
    const uint32_t o_tb = skip ? ta : tb0;             /* slice ends after leaving slices out */ \
    /* the units: slice t, 512 postings at a time; lane k keeps unit k */                       \
    uint32_t nu = 0;                                                                            \
    for (unsigned long long m = __ballot(o_tb > ta); m; m &= m - 1) {                           \
      const uint32_t t = __builtin_ctzll(m);                                                    \
      const uint32_t sa = __builtin_amdgcn_readlane(ta, t), sb = __builtin_amdgcn_readlane(o_tb, t); \
      const uint32_t su = slice_units(sa, sb);                                                  \
      if (nu + su <= 64) {                                                                      \
        if (lane >= nu && lane < nu + su) s_units[slot_][lane] = make_uint2(sa + (lane - nu) * 512, sb); \
      }                                                                                         \
      nu += su;                                                                                 \
    }                                                                                           \
    if (nu > 64) {                                     /* too many to list: publish the slice table instead, */ \
      s_units[slot_][lane] = make_uint2(ta, o_tb);     /* every wave walks it (lane t: slice t) */ \
    }                                                                                           \
    if (lane == 0) {                                                                            \
      s_ctl.pub_units[slot_] = nu; s_ctl.pub_L[slot_] = L_;                                     \
      s_ctl.pub_wide[slot_] = min((T_) - L_, wmt) > 15;  /* a cold count could overflow four bits */ \
    }                                                                                           \
  } while (0)
    // a task's state (the needle's best keys so far) is requested a task ahead: thread t holds key t
#define WS_FETCH_STATE(ti_)                                                               \
  do {                                                                                    \
    pf_key = kKeyInf;                                                                     \
    if ((ti_) < n_tasks && tid < ((s_task_meta[ti_] >> 8) & 0xFFu)) pf_key = ws_load_key(A, s_task_q[ti_], tid); \
  } while (0)
    unsigned long long pf_key;
    WS_FETCH_STATE(0u);
    WS_FETCH_CODES(wid);
    WS_FETCH_TABLE(wid);
    WS_FETCH_CODES(kWsNW + wid);

    for (uint32_t g = 0; g < n_tasks; g += kWsNW) {
      // this wave's own task of the group
      const uint32_t my_ti = g + wid;
      const uint32_t my_T = my_ti < n_tasks ? (s_task_meta[my_ti] & 0xFFu) : 0u;
      const bool own = lane < my_T;
      const uint32_t ta = nx_ta, tb0 = nx_tb, bm = nx_bm;     // (waits for the prefetched table)
      WS_FETCH_TABLE(g + kWsNW + wid);                        // next group: its codes arrived a group ago
      if (STATS(A) && my_ti < n_tasks) { st_tab += 2 * my_T; ++st_tasks; }
      // Every wave publishes the first pass of its own task now, the four of them side by side -- from what the
      // filter learnt of the needle (its threshold's bound, whether it has one) -- instead of one after the other
      // with the other three waves waiting at a barrier (a fifth of a task's time, profiles/r03: "skipsel").  One
      // barrier per group makes all four visible.
      if (my_ti < n_tasks) {
        const uint32_t meta_ = s_task_meta[my_ti];
        WS_PUBLISH(wid, my_T, meta_ >> 16, ((meta_ >> 8) & 0xFFu) < keep);
      }
      ws_barrier();

      for (uint32_t j = 0; j < kWsNW && g + j < n_tasks; ++j) {
        const uint32_t ti = g + j;
        const bool owner = wid == j;
        const uint32_t q = s_task_q[ti];
        const uint32_t meta = s_task_meta[ti];
        const uint32_t T = meta & 0xFFu, cnt0 = (meta >> 8) & 0xFFu;
        // ---- the needle's state: its best keys so far (requested a task ago) --------------------
        const unsigned long long my_key = pf_key;
        WS_FETCH_STATE(ti + 1);
        if (tid < cnt0) s_pool[tid] = my_key;
        if (tid == 0) { ctl->pool_n = cnt0; ctl->overflow = 0; }   // (the candidate list was emptied when the last task ended:
                                                                   //  waves push to it while they count, behind no barrier)
        if (tid == keep - 1 || (tid == 0 && cnt0 < keep)) ctl->thr = cnt0 >= keep ? my_key : kKeyInf;
        // (no barrier: nothing of the state is read before the first pass's count barrier -- the bound and the pool's
        // size come from what the filter noted of the needle, the threshold is read behind that barrier)
        bool changed = false, first_pass = true;
        bool robust = cnt0 < keep;                           // no threshold yet: nothing can be left out
        PATH_FLAG(A, q, robust ? kPathWsTask | kPathWsRobust : kPathWsTask);
        WS_CLOCK(1);

        for (;;) {                                           // again after an overflow (rare)
          unsigned long long thr = kKeyInf;
          uint32_t pool_at_start = cnt0, need = meta >> 16;
          if (!first_pass) { thr = ctl->thr; pool_at_start = ctl->pool_n; need = matches_needed(thr, T, wbase); }
          const bool was_first = first_pass;
          if (min(T, wmt) < need) break;                     // (the threshold tightened in an earlier pass)
          // (the first pass was published when the group began; a later one -- the threshold tightened, or the robust
          // way after an overflow -- by the owner now)
          if (!first_pass) {
            if (owner) WS_PUBLISH(j, T, need, robust);
            ws_barrier();
          }
          first_pass = false;
          const uint32_t n_units = s_ctl.pub_units[j], L = s_ctl.pub_L[j];
          const bool wide = s_ctl.pub_wide[j] != 0;
          const uint32_t need_eff = need - L;                // >= cmin >= 1 when L > 0
          if (n_units == 0) break;                           // nothing to count: nothing can reach need_eff >= 1
          ++st_steps;
          PATH_FLAG(A, q, (L ? kPathWsLeftOut : 0u) | (wide ? kPathWsWide : 0u) | (n_units > 64 ? kPathWsTableWalk : 0u));
          WS_CLOCK(6);                                       // the left-out set chosen, the units published

          // The usual pass -- a threshold, four bits enough, the units listed -- finds its candidates WHILE counting
          // (ws_bump8_cross) and then only clears the window's counters, blindly; the others count first and scan.
          const bool cross = !robust && !wide && n_units <= 64;
          if (cross) {
            constexpr uint32_t kAheadX = kWsAhead;             // (5, 6, 8 loads ahead measured: 0 / +1 / +3 %; two units'
                                                               //  sixteen atomics behind one wait: +3 %)
            for (uint32_t k0 = wid; k0 < n_units; k0 += kAheadX * kWsNW) {
              uint4 u[kAheadX];
#pragma unroll
              for (uint32_t i = 0; i < kAheadX; ++i) {
                const uint32_t k = k0 + i * kWsNW;
                const uint2 d = s_units[j][min(k, 63u)];
                const uint32_t c = __builtin_amdgcn_readfirstlane(d.x);
                const uint32_t e = k < n_units ? __builtin_amdgcn_readfirstlane(d.y) : 0u;
                u[i] = load_group(A.ent, c + lane * 8, e);
                if (STATS(A) && k < n_units) { st_ent += min(512u, e - c); ++st_units; }
              }
#pragma unroll
              for (uint32_t i = 0; i < kAheadX; ++i)
                ws_bump8_cross(s_cnt, u[i], need_eff, wlen, s_cand, &s_ctl.n_cand, &s_ctl.cand_ov);
            }
            WS_CLOCK(7);                                     // units loaded and counted
            ws_barrier();
            WS_CLOCK(2);                                     // barrier after the count
            if (was_first) thr = ctl->thr;                   // (the task's state is visible from here on)
            WS_CLOCK(3);
          }
          for (uint32_t h = 0; !cross && h < (wide ? 2u : 1u); ++h) {
            // ---- count: unit k belongs to wave k mod 4; four loads travel together ---------------------
            if (n_units <= 64) {
              for (uint32_t k0 = wid; k0 < n_units; k0 += kWsAhead * kWsNW) {
                uint4 u[kWsAhead];
#pragma unroll
                for (uint32_t i = 0; i < kWsAhead; ++i) {
                  const uint32_t k = k0 + i * kWsNW;
                  const uint2 d = s_units[j][min(k, 63u)];
                  const uint32_t c = __builtin_amdgcn_readfirstlane(d.x);
                  const uint32_t e = k < n_units ? __builtin_amdgcn_readfirstlane(d.y) : 0u;
                  u[i] = load_group(A.ent, c + lane * 8, e);
                  if (STATS(A) && k < n_units) { st_ent += min(512u, e - c); ++st_units; }
                }
#pragma unroll
                for (uint32_t i = 0; i < kWsAhead; ++i) {
                  if (wide) ws_bump8<true>(s_cnt, u[i], h); else ws_bump8<false>(s_cnt, u[i], 0u);
                }
              }
            } else {
              // the published slice table: this wave's units of every slice (unit i of slice t: wave (t + i) mod 4)
              const uint2 d = s_units[j][lane];
              const uint32_t xa = d.x, xb = d.y;
              uint32_t k = 0;
              uint4 pend = make_uint4(kPadPair, kPadPair, kPadPair, kPadPair);
              BLURRILY_FOR_SLOT_UNITS(kWsNW, xa, xb, wid, lane, k, {
                const uint4 v = load_group(A.ent, c, sb);
                if (STATS(A)) { st_ent += min(512u, sb - (c - lane * 8)); ++st_units; }
                if (wide) ws_bump8<true>(s_cnt, pend, h); else ws_bump8<false>(s_cnt, pend, 0u);
                pend = v;
              });
              if (wide) ws_bump8<true>(s_cnt, pend, h); else ws_bump8<false>(s_cnt, pend, 0u);
            }
            WS_CLOCK(7);                                     // units loaded and counted
            ws_barrier();
            WS_CLOCK(2);                                     // barrier after the count
            if (was_first && h == 0) thr = ctl->thr;         // (the task's state is visible from here on)
            // ---- scan: counters that reach need_eff become candidates; everything is cleared -------
            if (!robust) {
              if (wide) ws_scan_fast<true>(cnt128, s_cand, &s_ctl.n_cand, &s_ctl.cand_ov, need_eff, h, wlen);
              else      ws_scan_fast<false>(cnt128, s_cand, &s_ctl.n_cand, &s_ctl.cand_ov, need_eff, 0u, wlen);
            } else {
              uint32_t floor_need = need_eff;                // (== need: nothing is left out)
              if (thr == kKeyInf && T > 1 && !A.tomb) {
                // no threshold at all: admit only counters that can be among the best `keep` of this window
                // (or half of it) alone (cold_start_need's argument), found by bisection over the counter value
                uint32_t lo = 1, hi = min(T, wide ? 127u : 15u);
                while (lo < hi) {
                  const uint32_t mid = (lo + hi + 1) >> 1;
                  if (tid == 0) ctl->tally = 0;
                  ws_barrier();
                  uint32_t mine = wide ? ws_count_hits<true>(cnt128, mid, h, wlen) : ws_count_hits<false>(cnt128, mid, 0u, wlen);
#pragma unroll
                  for (uint32_t dd = 32; dd; dd >>= 1) mine += __shfl_xor(mine, int(dd));
                  if (lane == 0 && mine) atomicAdd(&ctl->tally, mine);
                  ws_barrier();
                  if (ctl->tally >= keep) lo = mid; else hi = mid - 1;
                  ws_barrier();
                }
                floor_need = max(floor_need, lo);
              }
              if (wide) ws_scan_robust<true>(A, cnt128, ctl, s_pool, thr, T, wbase, floor_need, h, wlen);
              else      ws_scan_robust<false>(A, cnt128, ctl, s_pool, thr, T, wbase, min(floor_need, 16u), 0u, wlen);
            }
            // what no vector reached: the padding slot (short window), the dump word of the byte halves
            if (tid == 0) {
              const uint32_t nv_here = wide ? WsLayout<true>(1u, h, wlen).nv : WsLayout<false>(1u, 0u, wlen).nv;
              if (nv_here < kWsCntWords / 4) s_cnt[kWsCntWords - 1] = 0;
              if (wide) s_cnt[kWsCntWords] = 0;
            }
            ws_barrier();
            WS_CLOCK(3);
          }
          // ---- probe: the left-out slices' bitmaps give every candidate its exact match count.  A
          // thread owns candidates tid and tid + 256; four slices' words travel together. ----------------
          if (!robust && !s_ctl.cand_ov) {                   // (an overflowed list is abandoned: the robust pass redoes it all)
            const uint32_t n_cand = s_ctl.n_cand;
            if (STATS(A) && wid == 0) st_probe += n_cand * L;
            const bool has0 = tid < n_cand, has1 = tid + kWsNT < n_cand;
            const uint32_t c0 = has0 ? s_cand[tid] : 0u, c1 = has1 ? s_cand[tid + kWsNT] : 0u;
            const uint32_t r0 = c0 & 0xFFFFu, r1 = c1 & 0xFFFFu;
            uint32_t t0 = c0 >> 16, t1 = c1 >> 16;
            if (cross) {
              // the list holds ranks only: a candidate's count as it stands now that all postings are in (it may
              // have grown past need_eff since the candidate was noted)
              t0 = (s_cnt[(r0 >> 2) & 0x1FFFu] >> (((r0 & 3u) | ((r0 >> 15) << 2)) * 4u)) & 15u;
              t1 = (s_cnt[(r1 >> 2) & 0x1FFFu] >> (((r1 & 3u) | ((r1 >> 15) << 2)) * 4u)) & 15u;
            }
            if (__ballot(has0)) {
              for (uint32_t i0 = 0; i0 < L; i0 += 4) {
                uint32_t x0[4], x1[4];
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i) {
                  x0[i] = x1[i] = 0;
                  if (i0 + i < L) {                                        // (uniform)
                    const uint32_t id = __builtin_amdgcn_readfirstlane(s_hot[j][i0 + i]);
                    const uint32_t* bmw = reinterpret_cast<const uint32_t*>(A.ent + (id - kBitmapSlots));
                    if (has0) x0[i] = bmw[r0 >> 5];
                    if (has1) x1[i] = bmw[r1 >> 5];
                  }
                }
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i) { t0 += (x0[i] >> (r0 & 31)) & 1u; t1 += (x1[i] >> (r1 & 31)) & 1u; }
              }
              if (has0) ws_admit(A, ctl, s_pool, thr, T, wbase, r0, t0);
              if (has1) ws_admit(A, ctl, s_pool, thr, T, wbase, r1, t1);
            }
            ws_barrier();
          }
          if (cross) {
            // the counters, cleared blindly (everybody's reads of them lie behind the barrier above; an overflowed
            // list is not read at all) -- complete before the barriers every way to the next count leads through
            for (uint32_t i = tid; i < kWsCntWords / 4; i += kWsNT) {
              uint32_t z = 0;
              asm volatile("" : "+v"(z));
              cnt128[i] = make_uint4(z, z, z, z);
            }
          }
          WS_CLOCK(4);
          // ---- select ---------------------------------------------------------------------------
          const uint32_t cov = s_ctl.cand_ov;
          const uint32_t pov = ctl->overflow;
          const uint32_t ov = pov | cov, pn = ctl->pool_n;
          if (!ov && pn == pool_at_start) break;                           // nothing was admitted
          ws_barrier();
          if (tid == 0) { s_ctl.n_cand = 0; s_ctl.cand_ov = 0; }
          ++st_compact;
          ws_compact_pool(s_pool, ctl, keep, pool_at_start);
          changed = true;
          if (!ov) break;
          // An overflow: candidates of this window were lost.  Keep the tightened threshold, forget this
          // window's survivors and sweep it again (every pass shrinks the admitted set) -- the robust way
          // if it was the candidate list that overflowed.
          ++st_redo;
          PATH_FLAG(A, q, (cov ? kPathWsCandOv | kPathWsRobust : 0u) | (pov ? kPathWsPoolOv : 0u));
          if (cov) robust = true;
          if (tid == 0) {
            uint32_t jj = 0;
            const uint32_t np = ctl->pool_n;
            for (uint32_t i = 0; i < np; ++i)
              if (uint32_t(s_pool[i]) - wbase >= wlen) s_pool[jj++] = s_pool[i];
            ctl->pool_n = jj;
          }
          ws_barrier();
        }
        // ---- write the state back if it changed --------------------------------------------------
        ws_barrier();
        if (changed) {
          const uint32_t pn = ctl->pool_n;                                  // sorted, <= keep
          uint32_t* st = reinterpret_cast<uint32_t*>(A.results + size_t(q) * A.limit);
          if (tid < pn) {
            const unsigned long long key = s_pool[tid];
            st[2 * tid] = uint32_t(key); st[2 * tid + 1] = uint32_t(key >> 32);
          }
          if (tid == 0) A.counts[q] = pn;
        }
        if (tid == 0) { s_ctl.n_cand = 0; s_ctl.cand_ov = 0; }
        ws_barrier();                                        // pool and control quiet before the next task
        WS_CLOCK(5);
      }
      WS_FETCH_CODES(g + 2 * kWsNW + wid);                   // the group after the next
    }
#undef WS_FETCH_STATE
#undef WS_PUBLISH
#undef WS_FETCH_TABLE
#undef WS_FETCH_CODES
  }
  if (STATS(A) && lane == 0) {
    atomicAdd(&STATS(A)[kStatPostingEntries], static_cast<unsigned long long>(st_ent));
    atomicAdd(&STATS(A)[kStatTableWords], static_cast<unsigned long long>(st_tab));
    atomicAdd(&STATS(A)[kStatTasks], static_cast<unsigned long long>(st_tasks));
    atomicAdd(&STATS(A)[kStatUnits], static_cast<unsigned long long>(st_units));
    if (wid == 0) {
      atomicAdd(&STATS(A)[kStatProbes], static_cast<unsigned long long>(st_probe));
      atomicAdd(&STATS(A)[kStatSteps], static_cast<unsigned long long>(st_steps));
      atomicAdd(&STATS(A)[kStatResweeps], static_cast<unsigned long long>(st_redo));
      atomicAdd(&STATS(A)[kStatCompactions], static_cast<unsigned long long>(st_compact));
      for (int i = 0; i < 8; ++i) atomicAdd(&STATS(A)[kStatWsClocks + i], ws_clk[i]);
    }
  }
}
#undef WS_CLOCK

// ============================================================ small haystacks ================
// An image of a few windows (configs[1]: 235 k words, four windows): the needle-major kernel above sweeps such a needle
// in two steps, and what a needle costs is its CHAIN -- queue pop, scalars, codes, slice table, units, rows: six
// dependent global round trips of ~2 800 clocks each, the cold start's bisection, two scans, two compactions: ~41 000
// clocks -- of which a CU runs two at a time (its LDS holds two workgroups' counters) and is otherwise idle: a fifth of
// the LDS pipe busy.  Here a needle gets FOUR waves and ONE window's 4-bit counters (32 KiB, wsweep_kernel's layout and
// its robust pass: count every slice, scan to the pool against the threshold, cold start by bisection), so that four
// needles' chains run per CU, each wave walking the needle's slice table itself (no publishing turn, no ring).  It owns
// needles with at most 15 distinct trigrams (4 bits suffice whatever the window); the others are listed for the
// byte-counter launch that follows (FindArgs::over_list).  Same answer by construction: every posting counted, every
// counter held against the exact threshold, any window order.
// (Measured and left out: eight waves per needle instead of four -- 512 threads, 64 VGPRs, 69 of them spilled -- 3.5 -> 4.2 ms
// per 100 k needles at configs[1]; a third unit loaded a window ahead -- the kernel sits at 128 VGPRs: 12 spilled -- 3.6 -> 3.7 ms.)
constexpr uint32_t kSmallCand = 1024;
__global__ __launch_bounds__(kWsNT, kWsNT / 64) void find_small_kernel(const FindArgs A) {
  __shared__ __attribute__((aligned(16))) uint32_t s_cnt[kWsCntWords + 4];
  __shared__ unsigned long long s_pool[kWsPool];
  __shared__ Control s_ctl;
  __shared__ uint32_t s_tally[8];                        // the cold start's tallies, one per pass of the search
  __shared__ uint32_t s_hist[16];                        // a cold window's counters that reach m, m = 2 .. 15 (ws_bump8_hist)
  constexpr bool kSmallHist = true;
  __shared__ uint32_t s_nextq[2];                        // the queue slot popped for the needle after the current one, by that needle's parity
  __shared__ uint32_t s_cand[kSmallCand];                // a window's counters at the bound: in-window rank | count << 16
  __shared__ uint32_t s_ncand[2];                        // its length, by the harvest's parity (the other slot is reset meanwhile: no barrier for that)
  uint4* cnt128 = reinterpret_cast<uint4*>(s_cnt);
  Control* ctl = &s_ctl;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t keep = A.keep;
  const uint32_t sel_at = min(keep + max(6u, keep / 2), kWsPool / 2);   // (3 keep + 6, 6 keep + 6: 3.13 -> 3.16 ms per 100 k needles at configs[1])
  constexpr uint32_t kVecs = kWsCntWords / 4 / kWsNT;    // a thread's vectors of the window's counters: eight
  for (uint32_t i = tid; i < (kWsCntWords + 4) / 4; i += kWsNT) cnt128[i] = make_uint4(0, 0, 0, 0);
  if (tid < 8) s_tally[tid] = 0;
  if (tid < 16) s_hist[tid] = 0;
  if (tid < 2) s_ncand[tid] = 0;
  uint32_t hp = 0;                                       // the harvest's parity
  uint32_t st_ent = 0, st_steps = 0, st_redo = 0, st_tasks = 0, st_compact = 0, st_tab = 0;
  ws_barrier();
  // The NEXT needle is set up while the current one is swept -- its queue pop, its scalars, its codes, its first slice
  // table are four dependent global round trips (~2 800 clocks each under load) that a needle of four windows can ill
  // afford in a row: one stage is advanced at the top of every window of the current sweep, whatever is left is caught up
  // on when the needle's turn comes.  stage: 0 nothing yet, 1 pop issued (thread 0 holds the slot), 2 slot in LDS,
  // 3 scalars requested, 4 codes requested, 5 table requested.
  constexpr uint32_t kSmallChunk = 4;
  uint32_t ch_base = 0, ch_left = 0;                     // (thread 0: what is left of the slots it took last)
  uint32_t stage = 0, nx_pop = 0, nx_q = 0, nx_T = 0, nx_nb = 0, nx_qs = 0, nx_code = 0;
  uint32_t nx_par = 0;                                   // (two slots by turns: a needle that is skipped at once -- too long, empty -- has
                                                         //  thread 0 writing the NEXT needle's slot while slow waves still read this one's)
  uint64_t nx_off = 0;
  bool nx_ok = false;
  uint2 nx_se = make_uint2(0, 0);
#define SMALL_ADVANCE()                                                                   \
  do {                                                                                    \
    switch (stage) {                                                                      \
      case 0:                                                                             \
        nx_par ^= 1u;                                                                     \
        if (tid == 0) {                    /* (slots are taken kSmallChunk at a time: see above) */ \
          if (ch_left == 0) { ch_base = atomicAdd(A.queue, kSmallChunk); ch_left = kSmallChunk; } \
          nx_pop = ch_base++; --ch_left;                                                  \
        }                                                                                 \
        stage = 1; break;                                                                 \
      case 1: if (tid == 0) s_nextq[nx_par] = nx_pop; stage = 2; break;                   \
      case 2:                                                                             \
        ws_barrier();                                                                     \
        nx_q = s_nextq[nx_par];                                                           \
        nx_ok = nx_q < A.n_work;                                                          \
        if (nx_ok) { nx_T = A.q_ntri[nx_q]; nx_nb = A.q_nb[nx_q]; nx_off = A.offsets[nx_q]; nx_qs = A.q_start[nx_q]; } \
        stage = 3; break;                                                                 \
      case 3:                                                                             \
        nx_code = 0;                                                                      \
        if (nx_ok && nx_T <= 15 && lane < nx_T) nx_code = A.qcodes[nx_off + nx_q + lane]; \
        stage = 4; break;                                                                 \
      case 4: {                                                                           \
        nx_se = make_uint2(0, 0);                                                         \
        const uint32_t ws_ = (A.n_windows >= 8 && nx_qs < A.n_windows) ? nx_qs : 0u;      \
        if (nx_ok && nx_T <= 15 && nx_nb != 0 && lane < nx_T) nx_se = A.slice_se[size_t(ws_) * kNumCodes + nx_code]; \
        stage = 5; break;                                                                 \
      }                                                                                   \
      default: break;                                                                     \
    }                                                                                     \
  } while (0)
  for (;;) {
    while (stage < 5) SMALL_ADVANCE();                   // (the first needle; a needle whose setup the sweep before did not finish)
    if (!nx_ok) break;
    const uint32_t q = nx_q, T = nx_T, qs = nx_qs;
    const bool empty = nx_nb == 0;
    const uint32_t code = nx_code;
    uint2 se = nx_se, se_next = make_uint2(0, 0);
    stage = 0;
    SMALL_ADVANCE();                                     // the pop for the needle after this one goes out at once
    if (T > 15) {                                        // byte counters: the launch that follows
      if (tid == 0 && T <= 64) A.over_list[atomicAdd(A.over_count, 1u)] = q;
      continue;
    }
    if (empty || keep == 0) {
      if (tid == 0) A.counts[q] = 0;
      continue;
    }
    if (wid == 0) TRACE_MARK_AT(A, 50000u, q, 7u, 0u, 0u);
    if (tid == 0) { ctl->pool_n = 0; ctl->overflow = 0; ctl->thr = kKeyInf; }
    PATH_FLAG(A, q, kPathSmall | kPathNibble);
    ++st_tasks;
    // (the sweep starts at the needle's own length class only where there are enough windows for it to pay, as above)
    const uint32_t ws = (A.n_windows >= 8 && qs < A.n_windows) ? qs : 0u;
    uint32_t n_sorted = 0;                               // keys at the head of the pool that are in order
    // slice table of the first window, every wave its own copy (lane t: trigram t); the next window's travels meanwhile
    // this wave's first two units of the window about to be counted, loaded a window ahead (while the one before is scanned)
    uint4 h0 = make_uint4(kPadPair, kPadPair, kPadPair, kPadPair), h1 = h0;
    bool head_ok = false;
#define SMALL_LOAD_HEAD(ta_, tb_)                                                         \
  do {                                                                                    \
    uint32_t k_ = 0, x0_ = 0, y0_ = 0, x1_ = 0, y1_ = 0;                                  \
    BLURRILY_FOR_SLOT_UNITS(kWsNW, ta_, tb_, wid, lane, k_, {                             \
      x0_ = k_ == 0 ? c : x0_; y0_ = k_ == 0 ? sb : y0_;                                  \
      x1_ = k_ == 1 ? c : x1_; y1_ = k_ == 1 ? sb : y1_;                                  \
    });                                                                                   \
    h0 = load_group(A.ent, x0_, y0_);                                                     \
    h1 = load_group(A.ent, x1_, y1_);                                                     \
    head_ok = true;                                                                       \
  } while (0)
    ws_barrier();                                        // the control block is set
    for (uint32_t i = 0; i < A.n_windows; ++i) {
      if (wid == 0) TRACE_MARK_AT(A, 50000u, q, i, 0u, 0u);
      SMALL_ADVANCE();                                   // one stage of the next needle's setup
      if (wid == 0) TRACE_MARK_AT(A, 50000u, q, i, 0u, 1u);
      const uint32_t w = ws + i < A.n_windows ? ws + i : ws + i - A.n_windows;
      const uint32_t wn = ws + i + 1 < A.n_windows ? ws + i + 1 : ws + i + 1 - A.n_windows;
      if (i + 1 < A.n_windows && lane < T) se_next = A.slice_se[size_t(wn) * kNumCodes + code];
      if (i + 1 >= A.n_windows) se_next = make_uint2(0, 0);
      if (STATS(A)) st_tab += 2 * T;
      const uint32_t wbase = w * kWindowRanks;
      const uint32_t wlen = min(kWindowRanks, A.n_refs - wbase);
      const uint32_t ta = se.x, tb = se.y;
      se = se_next;
      for (;;) {                                         // again after a pool overflow (rare)
        const unsigned long long thr = ctl->thr;
        const uint32_t need = matches_needed(thr, T, wbase);
        if (min(T, A.win_max_tri[w]) < need) { PATH_FLAG(A, q, kPathSkipped); head_ok = false; break; }   // no reference of the window can get in
        if (__ballot(tb > ta) == 0) { head_ok = false; break; }   // nothing of the needle in this window
        ++st_steps;
        // ---- count: unit j of slice t belongs to wave (t + j) mod 4; the first two are here already (or loaded now),
        // the others are loaded and counted in place, one unit's atomics under the next one's load
        const bool cold = thr == kKeyInf && T > 1 && !A.tomb;    // no threshold yet: the window's own bound first
        {
          if (!head_ok) SMALL_LOAD_HEAD(ta, tb);
          head_ok = false;
          auto count = [&](auto hist_too) {
            constexpr bool kHist = decltype(hist_too)::value;
            auto bump = [&](const uint4 grp) { if (kHist) ws_bump8_hist(s_cnt, grp, s_hist); else ws_bump8<false>(s_cnt, grp, 0u); };
            // (the wave's third unit requested BEFORE the two that are here are counted, so that its load travels under
            // both -- measured: 3.25 -> 3.43 ms per 100 k needles; the walk to the third unit delays the first atomics)
            uint32_t k = 0;
            uint4 pend = h1;
            bump(h0);
            BLURRILY_FOR_SLOT_UNITS(kWsNW, ta, tb, wid, lane, k, {
              if (STATS(A)) st_ent += min(512u, sb - (c - lane * 8));
              if (k >= 2) {
                const uint4 v = load_group(A.ent, c, sb);
                bump(pend);
                pend = v;
              }
            });
            bump(pend);
          };
          if (kSmallHist && cold) count(std::true_type{}); else count(std::false_type{});
        }
        if (wid == 0) TRACE_MARK_AT(A, 50000u, q, i, 0u, 2u);
        // the next window's first units go out now: they travel under this window's scan
        if (i + 1 < A.n_windows) SMALL_LOAD_HEAD(se.x, se.y);
        if (wid == 0) TRACE_MARK_AT(A, 50000u, q, i, 0u, 3u);
        ws_barrier();
        if (wid == 0) TRACE_MARK_AT(A, 50000u, q, i, 0u, 4u);
        // ---- scan: the thread's eight vectors of counters are read ONCE, together, and cleared at once; the cold start's
        // bisection and the harvest work on the registers
        const WsLayout<false> Y0(1u, 0u, wlen);
        uint4 v[kVecs];
#pragma unroll
        for (uint32_t j = 0; j < kVecs; ++j) {
          const uint32_t iv = tid + j * kWsNT;
          v[j] = make_uint4(0, 0, 0, 0);
          if (iv < Y0.nv) v[j] = cnt128[iv];
        }
#pragma unroll
        for (uint32_t j = 0; j < kVecs; ++j) {
          const uint32_t iv = tid + j * kWsNT;
          uint32_t z = 0;
          asm volatile("" : "+v"(z));
          if (iv < Y0.nv) cnt128[iv] = make_uint4(z, z, z, z);
        }
        v[kVecs - 1] = Y0.mask_pad(v[kVecs - 1], tid + (kVecs - 1) * kWsNT);
        if (tid == 0 && Y0.nv < kWsCntWords / 4) s_cnt[kWsCntWords - 1] = 0;   // the padding slot's word (a short window)
        if (wid == 0) TRACE_MARK_AT(A, 50000u, q, i, 0u, 5u);
        uint32_t floor_need = need;
        if (kSmallHist && cold) {
          // no threshold yet: only counters that can be among the window's best `keep` (cold_start_need's argument) -- the
          // largest m that `keep` counters reach, read off the count's own histogram (through round 5: a bisection over
          // the counters in registers, a pass and a barrier per probe: 7 800 of a needle's 78 800 clocks)
          PATH_FLAG(A, q, kPathColdStart);
          const uint32_t reach_m = lane >= 2 && lane < 16 ? s_hist[lane] : 0u;
          const unsigned long long ok = __ballot(lane >= 2 && lane <= min(T, 15u) && reach_m >= keep);
          if (ok) floor_need = max(floor_need, 63u - uint32_t(__builtin_clzll(ok)));
        } else if (cold) {
          PATH_FLAG(A, q, kPathColdStart);
          uint32_t lo = 1, hi = min(T, 15u), pass = 0;
          while (lo < hi) {
            // (upward from 2 first -- the bound of a first window is 2 or 3 for most needles: two or three passes where
            // a bisection of 1..15 always takes four --, a bisection of what is left from 4 on: seven passes at most)
            const uint32_t mid = lo < 4 ? lo + 1 : (lo + hi + 1) >> 1;
            const WsLayout<false> Y(mid, 0u, wlen);
            uint32_t mine = 0;
#pragma unroll
            for (uint32_t j = 0; j < kVecs; ++j)
              mine += __popc(Y.hits(v[j].x)) + __popc(Y.hits(v[j].y)) + __popc(Y.hits(v[j].z)) + __popc(Y.hits(v[j].w));
#pragma unroll
            for (uint32_t dd = 32; dd; dd >>= 1) mine += __shfl_xor(mine, int(dd));
            if (lane == 0 && mine) atomicAdd(&s_tally[pass], mine);
            ws_barrier();
            if (s_tally[pass] >= keep) lo = mid; else hi = mid - 1;
            ++pass;
          }
          ws_barrier();                                  // everyone has read the last tally
          if (tid < 8) s_tally[tid] = 0;
          floor_need = max(floor_need, lo);
        }
        if (wid == 0) TRACE_MARK_AT(A, 50000u, q, i, 0u, 6u);
        {
          // harvest in two phases: every thread LISTS its counters at the bound (one SWAR pass to count them, one atomic
          // to reserve their places, a light loop to write rank | count), then the list is worked off a candidate per
          // thread -- key, threshold, pool -- without the divergence of 256 threads walking 32 words each for the hits
          // of a few (9 000 of a first window's 14 000 clocks)
          const WsLayout<false> Y(min(floor_need, 16u), 0u, wlen);
          uint32_t n_hit = 0;
#pragma unroll
          for (uint32_t j = 0; j < kVecs; ++j)           // (one AND per vector first: with a threshold most hold nothing at the bound)
            if (((v[j].x | v[j].y | v[j].z | v[j].w) & Y.pre) != 0)
              n_hit += __popc(Y.hits(v[j].x)) + __popc(Y.hits(v[j].y)) + __popc(Y.hits(v[j].z)) + __popc(Y.hits(v[j].w));
          if (n_hit) {
            uint32_t at = atomicAdd(&s_ncand[hp], n_hit);
            if (at + n_hit <= kSmallCand) {
#pragma unroll
              for (uint32_t j = 0; j < kVecs; ++j) {
                const uint32_t d[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
                if (((d[0] | d[1] | d[2] | d[3]) & Y.pre) == 0) continue;
#pragma unroll
                for (uint32_t jj = 0; jj < 4; ++jj) {
                  uint32_t m = Y.hits(d[jj]);
                  while (m) {
                    const uint32_t bit = __ffs(m) - 1;
                    m &= m - 1;
                    s_cand[at++] = Y.rank16(tid + j * kWsNT, jj, bit) | (Y.count(d[jj], bit) << 16);
                  }
                }
              }
            }
          }
          ws_barrier();
          if (kSmallHist && cold && tid < 16) s_hist[tid] = 0;   // (everyone has read it: ready for the next cold count)
          const uint32_t n_cand = s_ncand[hp];
          if (tid == 0) s_ncand[hp ^ 1u] = 0;            // (the harvest before this one is done with it)
          hp ^= 1u;
          if (n_cand <= kSmallCand) {
            for (uint32_t c2 = tid; c2 < n_cand; c2 += kWsNT) {
              const uint32_t e2 = s_cand[c2];
              ws_admit(A, ctl, s_pool, thr, T, wbase, e2 & 0xFFFFu, e2 >> 16);
            }
          } else {
            // more counters at the bound than the list holds (a flood of ties): the list is dropped, every thread admits
            // its own hits -- only keys that beat the threshold take room in the pool, so the passes shrink whatever ties
            // lie behind the threshold's rank (wsweep_kernel's robust scan)
#pragma unroll
            for (uint32_t j = 0; j < kVecs; ++j) {       // (unrolled: a dynamic index would put the vectors in scratch)
              const uint32_t d[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
              for (uint32_t jj = 0; jj < 4; ++jj) {
                uint32_t m = Y.hits(d[jj]);
                while (m) {
                  const uint32_t bit = __ffs(m) - 1;
                  m &= m - 1;
                  ws_admit(A, ctl, s_pool, thr, T, wbase, Y.rank16(tid + j * kWsNT, jj, bit), Y.count(d[jj], bit));
                }
              }
            }
          }
        }
        ws_barrier();
        if (wid == 0) TRACE_MARK_AT(A, 50000u, q, i, 0u, 7u);
        const uint32_t ov = ctl->overflow, pn = ctl->pool_n;
        if (!(ov || pn > sel_at || (thr == kKeyInf && pn >= keep))) break;
        ws_barrier();                                    // (everyone has read the pool's state)
        ++st_compact;
        PATH_FLAG(A, q, ov ? kPathCompaction | kPathResweep : kPathCompaction);
        ws_compact_pool(s_pool, ctl, keep, n_sorted);
        n_sorted = ctl->pool_n;
        if (!ov) break;
        // the pool overflowed: candidates of this window were lost -- keep the tightened threshold, forget the
        // window's survivors, sweep it again (every pass shrinks the admitted set)
        ++st_redo;
        head_ok = false;                                 // (the units loaded ahead are the NEXT window's)
        if (tid == 0) {
          uint32_t jj = 0;
          const uint32_t np = ctl->pool_n;
          for (uint32_t k2 = 0; k2 < np; ++k2)
            if (uint32_t(s_pool[k2]) - wbase >= wlen) s_pool[jj++] = s_pool[k2];
          ctl->pool_n = jj;
        }
        ws_barrier();
        n_sorted = ctl->pool_n;
      }
    }
#undef SMALL_LOAD_HEAD
    // ---- emit: best `keep` in final order
    if (wid == 0) TRACE_MARK_AT(A, 50000u, q, 7u, 0u, 1u);
    ws_barrier();
    ws_compact_pool(s_pool, ctl, keep, n_sorted);
    const uint32_t nres = ctl->pool_n;
    trigram_match_t* out = A.results + size_t(q) * A.limit;
    if (tid < nres) {
      const unsigned long long key = s_pool[tid];
      const uint32_t rk = uint32_t(key);
      trigram_match_t r;
      r.reference = A.ref_of_rank[rk];
      r.matches = T - uint32_t(key >> 32);
      r.weight = A.weight_of_rank[rk];
      out[tid] = r;
    }
    if (tid == 0) A.counts[q] = nres;
    ws_barrier();                                        // pool reads done before the next needle resets it
    if (wid == 0) TRACE_MARK_AT(A, 50000u, q, 7u, 0u, 2u);
  }
#undef SMALL_ADVANCE
  if (STATS(A) && lane == 0) {
    atomicAdd(&STATS(A)[kStatPostingEntries], static_cast<unsigned long long>(st_ent));
    atomicAdd(&STATS(A)[kStatTableWords], static_cast<unsigned long long>(st_tab));
    if (wid == 0) {
      atomicAdd(&STATS(A)[kStatSteps], static_cast<unsigned long long>(st_steps));
      atomicAdd(&STATS(A)[kStatTasks], static_cast<unsigned long long>(st_tasks));
      atomicAdd(&STATS(A)[kStatCompactions], static_cast<unsigned long long>(st_compact));
      atomicAdd(&STATS(A)[kStatResweeps], static_cast<unsigned long long>(st_redo));
    }
  }
}

// ============================================================ one needle, now ==================
// blurrily_storage_find -- the reference's only call shape (ext/blurrily/map_ext.c:131-162, bin/bench:93-95): ONE
// needle, the caller waiting.  Through round 4 it went the batch's way: a staged copy in, tokenise_wave_kernel,
// find_kernel in latency mode, merge_parts_small_kernel, a copy out, a stream synchronise -- six operations, each a
// packet the command processor orders behind the one before (68 us at Geonames scale, 39 of them the find kernel's:
// queue pop, needle scalars, codes, a learning sweep of the needle's own length class, then its own range).  Here it
// is ONE launch and no copy:
//   * the host tokenises (csrc/tokeniser.h: what put() runs) and passes the needle's codes AS KERNEL ARGUMENTS;
//   * workgroup g owns windows [g * per, (g + 1) * per) -- one window each at Geonames scale, so every workgroup runs
//     exactly ONE step: slice table (every wave its own copy, lane t = trigram t), the wave's units with four loads
//     in flight (a workgroup is alone on its CU: 128 VGPRs), count, barrier, select -- no queue, no learning sweep,
//     no ring;
//   * SELECT takes a window's best `keep` counters EXACTLY, whatever ties it holds (one_select): every thread reads
//     its four vectors of counters once; a bisection over the counter value IN REGISTERS finds the largest c that at
//     least `keep` counters reach; counters above c enter the pool, and of the counters AT c only the lowest ranks
//     that are still missing -- found by a prefix count in rank order (wave w owns a contiguous 4 KiB of counters).
//     A window of thousands of identical strings -- thousands of counters tied at the bound, the common case of a
//     needle made of popular words -- therefore costs what any window costs; admitting every tie, overflowing the
//     pool and sweeping the window again (scan_window's way, made for a sweep that arrives with a threshold) took
//     30 us in such windows, measured;
//   * a workgroup leaves its best `keep` keys in global memory and sets its flag; the grid's LAST workgroup waits for
//     the flags, loads all the lists at once, keeps the keys not above the keep-th smallest list head, sorts those few, looks the
//     references and weights up and writes rows, count and a sequence word straight into host-coherent pinned
//     memory, where the host thread is polling.
// Same answer by construction: every posting of every window counted, a window's best `keep` candidates that beat the
// workgroup's threshold taken exactly, the best `keep` of the union of per-workgroup best `keep` lists (nothing can be
// lost).  Served: needles of at most 64 distinct trigrams, limits up to kOneMaxKeep, images without tombstones or
// pending puts (c_abi.hip: find_one); everything else goes the batch's way as before.
constexpr uint32_t kOnePool = 512;
constexpr int      kOneThreads = 1024;                   // (512 -- eight waves, 256 VGPRs each -- measured: 35.4 against 33.5 us)
constexpr uint32_t kOneAhead = 4;                        // units a wave loads before it counts the first of them

// find_one_kernel's third counter layout: ONE window in 4-bit counters, 32 KiB -- wsweep_kernel's (ws_bump_pair_nib: word
// (r >> 2) & 0x1FFF, nibble (r & 3) | (r >> 15) << 2: the low nibbles of a word hold ranks 4 w .. 4 w + 3 of the window's
// lower half, the high ones ranks 32768 + 4 w ..).  A workgroup that owns a single window (every workgroup up to 256
// windows) then reads and selects over 2 048 vectors instead of the pair layout's 4 096, half of whose nibbles it would
// never touch: the select is VALU work per word (bisection 4.5 us, ties 4 us of a find's 31 at Geonames scale).
struct Nib1 {};
template <> struct Packing<Nib1> : Packing<Nib> {};
template <> struct ScanTraits<Nib1> : ScanTraits<Nib> {
  static constexpr uint32_t kVecs = kWsCntWords / 4;                // 2 048
  static constexpr uint32_t kHalf = 32768;
  // vector i holds ranks 16 i .. 16 i + 15 of the lower half and kHalf + 16 i .. of the upper half
  static __device__ __forceinline__ uint32_t nvec(uint32_t wlen) { return (min(wlen, kHalf) + 15) / 16; }
  // counter index = 8 * word + nibble
  static __device__ __forceinline__ uint32_t rank_of(uint32_t wbase, uint32_t idx) {
    return wbase + ((idx >> 3) << 2) + (idx & 3u) + ((idx >> 2) & 1u) * kHalf;
  }
  // slot 0xFFFF -- the padding -- is the top nibble of the last word: cleared here where the scan does not reach it,
  // masked by the select where it does (a full window)
  static __device__ __forceinline__ void clear_unreached_pad(uint4* cnt128, uint32_t nv, uint32_t tid) {
    if (nv < kVecs && tid == 0) reinterpret_cast<uint32_t*>(cnt128)[kVecs * 4 - 1] = 0;
  }
};

// this wave's units of one step: lane t holds trigram t's slice of the step's (even) window (a0, b0) and, with 4-bit
// counters, of the odd one (a1, b1); unit k of the step belongs to wave k mod 16
template <typename CT, int NT>
__device__ __forceinline__ void one_count(const uint16_t* __restrict__ ent, uint32_t* cnt32, const uint32_t a0,
                                          const uint32_t b0, const uint32_t a1, const uint32_t b1, const uint32_t wid,
                                          const uint32_t lane) {
  constexpr bool kNib = std::is_same<CT, Nib>::value;
  constexpr uint32_t kNW = NT / 64;
  const uint32_t units0 = slice_units(a0, b0), units1 = kNib ? slice_units(a1, b1) : 0u;
  const uint32_t incl = wave_inclusive_sum(units0 + units1);
  const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
  const uint32_t excl = incl - units0 - units1;
  for (uint32_t k0 = wid; k0 < total; k0 += kOneAhead * kNW) {
    uint4 v[kOneAhead];
    uint32_t half[kOneAhead];
#pragma unroll
    for (uint32_t i = 0; i < kOneAhead; ++i) {
      const uint32_t k = k0 + i * kNW;
      v[i] = make_uint4(kPadPair, kPadPair, kPadPair, kPadPair);
      half[i] = 0;
      if (k < total) {                                   // (uniform)
        const uint32_t t = uint32_t(__builtin_ctzll(__ballot(incl > k)));
        const uint32_t ji = k - __builtin_amdgcn_readlane(excl, t), u0 = __builtin_amdgcn_readlane(units0, t);
        const bool odd = ji >= u0;
        const uint32_t start = odd ? __builtin_amdgcn_readlane(a1, t) + (ji - u0) * 512u
                                   : __builtin_amdgcn_readlane(a0, t) + ji * 512u;
        const uint32_t end = odd ? __builtin_amdgcn_readlane(b1, t) : __builtin_amdgcn_readlane(b0, t);
        half[i] = odd ? 1u : 0u;
        v[i] = load_group(ent, start + lane * 8, end);
      }
    }
#pragma unroll
    for (uint32_t i = 0; i < kOneAhead; ++i) {
      if constexpr (std::is_same<CT, Nib1>::value) ws_bump8<false>(cnt32, v[i], 0u);
      else bump_unit<CT>(cnt32, v[i], half[i]);
    }
  }
}

// The window's counters are complete: take its best `keep` candidates that beat the threshold -- EXACTLY that many at
// most, whatever ties there are -- into the pool, and clear the counters.  Wave w owns vectors [256 w, 256 w + 256) of
// the 4 096 (lane l its j-th round's vector 256 w + 64 j + l: conflict-free reads, and (w, j, l) ascending is rank
// ascending inside a window -- with 4-bit counters inside each of the pair's two windows, the even one's ranks first).
struct OneShared {
  uint32_t tally[12];           // the bisection's tallies, one per pass and bound
  uint32_t above[16], tie0[16];  // per wave: counters above the bound, counters at it (even | odd window << 16, clamped)
};
template <typename CT, int NT>
__device__ __forceinline__ void one_select(const FindArgs& A, uint4* cnt128, const uint32_t T, const uint32_t keep, Control* ctl,
                                           unsigned long long* pool, OneShared* sh, const uint32_t wbase,
                                           const uint32_t wlen) {
  using S = ScanTraits<CT>;
  using P = Packing<CT>;
  constexpr bool kNib1 = std::is_same<CT, Nib1>::value;             // one window, 4-bit: lower half in a word's low nibbles
  constexpr bool kNib = std::is_same<CT, Nib>::value || kNib1;       // two rank ranges per word (a pair's windows / a window's halves)
  constexpr uint32_t kNW = NT / 64, kR = S::kVecs / NT;      // rounds: a wave owns kR * 64 consecutive vectors
  static_assert(kNW * kR * 64 == S::kVecs && kR >= 1 && kNW <= 16, "the waves share the vectors evenly");
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t nvec = S::nvec(wlen);
  uint4 v[kR];
#pragma unroll
  for (uint32_t j = 0; j < kR; ++j) {
    const uint32_t i = (wid * kR + j) * 64 + lane;
    v[j] = make_uint4(0, 0, 0, 0);
    if (i < nvec) v[j] = cnt128[i];
  }
#pragma unroll
  for (uint32_t j = 0; j < kR; ++j) {                    // cleared at once: the rest works on the registers
    const uint32_t i = (wid * kR + j) * 64 + lane;
    if (i < nvec) cnt128[i] = make_uint4(0, 0, 0, 0);
  }
  S::clear_unreached_pad(cnt128, nvec, tid);
  if (kNib1) {                                           // (the padding's nibble, where the scan reaches it)
#pragma unroll
    for (uint32_t j = 0; j < kR; ++j)
      if ((wid * kR + j) * 64 + lane == S::kVecs - 1) v[j].w &= 0x0FFFFFFFu;
  }
  if (A.tomb) {
    // References deleted since the image was built (storage.c:584-612) must not take a place: their counters go to zero
    // in the registers, before anything is counted.  Vector i holds the byte cells of in-window ranks 16 i .. 16 i + 15
    // -- of BOTH windows of a pair with 4-bit counters (even window: low nibbles) --, byte 3 - (r & 3) of word (r >> 2) & 3:
    // sixteen tombstone bits per window, aligned (a window starts at a multiple of sixteen ranks).
    static_assert(kWindowRanks % 16 == 0, "a vector's tombstone bits sit in one half of one word");
#pragma unroll
    for (uint32_t j = 0; j < kR; ++j) {
      const uint32_t i = (wid * kR + j) * 64 + lane;
      if (i >= nvec) continue;
      const uint32_t r0 = wbase + 16 * i, r1 = r0 + (kNib1 ? 32768u : kWindowRanks);
      const uint32_t dead0 = r0 < A.n_refs ? (A.tomb[r0 >> 5] >> (r0 & 31u)) & 0xFFFFu : 0u;
      const uint32_t dead1 = kNib && r1 < A.n_refs ? (A.tomb[r1 >> 5] >> (r1 & 31u)) & 0xFFFFu : 0u;
      if ((dead0 | dead1) == 0) continue;
      uint32_t d[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        // rank 4 k + b of the vector is dead: byte 3 - b of word k goes (one window: nibble b, and 4 + b for the upper
        // half).  The four bits of a word are spread by arithmetic -- no table of masks, which the compiler would keep
        // in registers for the kernel's life (that was the kernel's one spilled VGPR).
        const uint32_t n0 = (dead0 >> (4 * k)) & 0xFu, n1 = (dead1 >> (4 * k)) & 0xFu;
        uint32_t gone;
        if (kNib1) {
          auto nibbles = [](uint32_t n) {                // bit b -> bit 4 b
            n = (n | (n << 6)) & 0x0303u;
            return (n | (n << 3)) & 0x1111u;
          };
          gone = (nibbles(n0) | (nibbles(n1) << 16)) * 0xFu;
        } else {
          auto bytes = [](uint32_t n) {                  // bit b -> bit 8 (3 - b): reversed, then bit j -> bit 8 j (no two partial products meet)
            return ((__brev(n) >> 28) * 0x00204081u) & 0x01010101u;
          };
          gone = kNib ? bytes(n0) * 0x0Fu | bytes(n1) * 0xF0u : bytes(n0) * 0xFFu;
        }
        d[k] &= ~gone;
      }
      v[j] = make_uint4(d[0], d[1], d[2], d[3]);
    }
  }
  // counters of this thread that reach `bound`
  auto reach = [&](const uint32_t bound) {
    const typename S::Need nq = S::prepare(bound);
    uint32_t n = 0;
#pragma unroll
    for (uint32_t j = 0; j < kR; ++j)
      n += __popc(S::hits(v[j].x, nq)) + __popc(S::hits(v[j].y, nq)) + __popc(S::hits(v[j].z, nq)) + __popc(S::hits(v[j].w, nq));
    return n;
  };
  uint32_t pass = 0;
  auto tally = [&](const uint32_t bound) {               // the workgroup's counters that reach `bound` (one barrier)
    const uint32_t incl = wave_inclusive_sum(reach(bound));
    if (lane == 63 && incl) atomicAdd(&sh->tally[pass], incl);
    __syncthreads();
    return sh->tally[pass++];
  };
  // the bound a counter has to reach to beat the threshold (1: no threshold yet); c: the largest value >= that bound
  // which at least `keep` counters reach -- or the bound itself, with fewer reaching it: then all of them are taken
  const unsigned long long thr = ctl->thr;
  const uint32_t cap = min(T, S::kMaxCount);
  uint32_t lo = matches_needed(thr, T, wbase), hi = cap;
  bool all = false;
  if (lo > cap) {
    all = true; lo = cap + 1;                            // nothing of this window can enter
  } else if (tally(lo) < keep) {
    all = true;
  } else {
    // (The bound read off a histogram the COUNT keeps -- find_small_kernel's cold start, ws_bump8_hist, a row per wave --
    // measured here: median workgroup's count 3.2 -> 3.8 us, the SLOWEST's count barrier 5.0 -> 8.2 us -- windows where
    // thousands of postings meet a counter at work queue on a handful of histogram words -- and the find's p50 28.4 ->
    // 31.5 us, p90 33 -> 89.  The bisection's five passes cost every workgroup the same 3 us; left as it is.)
    while (lo < hi) {                                    // (at most seven passes with byte counters, four with 4-bit ones)
      const uint32_t mid = (lo + hi + 1) >> 1;
      if (tally(mid) >= keep) lo = mid; else hi = mid - 1;
    }
  }
  const uint32_t c = lo;
  ONE_MARK(A, 6);
  // per lane and round: counters above c, counters AT c in the even / odd window
  const typename S::Need at_c = S::prepare(min(c, cap)), above_c = S::prepare(min(c + 1, S::kMaxCount));
  const bool has_above = c + 1 <= cap, none = c > cap;
  // a field's top bit, by window of the pair (by half of the one window)
  constexpr uint32_t kEven = kNib1 ? 0x00008888u : kNib ? 0x08080808u : 0x80808080u, kOdd = kNib1 ? 0x88880000u : kNib ? 0x80808080u : 0u;
  auto masks = [&](const uint32_t wv, uint32_t& up, uint32_t& tie) {
    const uint32_t r = none ? 0u : S::hits(wv, at_c);
    up = has_above ? S::hits(wv, above_c) : 0u;
    tie = all ? 0u : r & ~up;
    if (all) up = r;                                     // (fewer than keep reach the bound: every one of them is taken)
  };
  // per lane and round j: ties of the even window in bits 15:0, of the odd one in bits 31:16 (at most 16 each: one
  // inclusive sum serves both); whether the round holds anything above the bound: bit j of up_rounds
  uint32_t n_up = 0, up_rounds = 0, nt[kR];
#pragma unroll
  for (uint32_t j = 0; j < kR; ++j) {
    const uint32_t d[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
    uint32_t ups = 0;
    nt[j] = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      uint32_t up, tie;
      masks(d[k], up, tie);
      ups += __popc(up);
      nt[j] += __popc(tie & kEven) + (kNib ? __popc(tie & kOdd) << 16 : 0u);
    }
    n_up += ups;
    up_rounds |= ups ? 1u << j : 0u;
  }
  uint32_t in[kR];                                      // inclusive sums over the lanes, per round (both windows, packed)
#pragma unroll
  for (uint32_t j = 0; j < kR; ++j) in[j] = wave_inclusive_sum(nt[j]);
  const uint32_t up_incl = wave_inclusive_sum(n_up);
  if (lane == 63) {
    // (a wave's ties: at most 4 096 per window; published clamped to 2 047 -- sixteen waves' sum then stays inside 16
    // bits, and a clamped figure is still at least any quota, which is all a prefix has to tell)
    uint32_t tot = 0;
#pragma unroll
    for (uint32_t j = 0; j < kR; ++j) tot += in[j];
    sh->above[wid] = up_incl;
    sh->tie0[wid] = min(tot & 0xFFFFu, 2047u) | (min(tot >> 16, 2047u) << 16);   // (a lane-63 figure: the wave's)
  }
  __syncthreads();
  // what lies in front of this wave: lanes 0..15 hold the sixteen waves' figures
  const uint32_t wa = lane < kNW ? sh->above[lane] : 0u, wt = lane < kNW ? sh->tie0[lane] : 0u;
  const uint32_t ia = wave_inclusive_sum(wa), it = wave_inclusive_sum(wt);
  const uint32_t n_above = __builtin_amdgcn_readlane(ia, kNW - 1), n_tie0 = __builtin_amdgcn_readlane(it, kNW - 1) & 0xFFFFu;
  const uint32_t quota = all ? 0u : keep - min(keep, n_above);            // ties still wanted, lowest ranks first
  const uint32_t front = __builtin_amdgcn_readlane(it, wid) - __builtin_amdgcn_readlane(wt, wid);   // ties in front of this wave, packed
  uint32_t run0 = front & 0xFFFFu;                       // even-window ties in front of this wave
  uint32_t run1 = n_tie0 + (front >> 16);                // odd-window ties: behind ALL even ones
  auto admit = [&](const uint32_t cnt, const uint32_t rank) {
    const unsigned long long key = (static_cast<unsigned long long>(T - cnt) << 32) | rank;
    if (key <= thr) {                                    // (the bound came from the threshold's match count; its rank decides ties with it)
      const uint32_t at = atomicAdd(&ctl->pool_n, 1u);
      if (at < kOnePool) pool[at] = key;                 // (at most keep + keep keys: never full)
    }
  };
#pragma unroll
  for (uint32_t j = 0; j < kR; ++j) {
    const uint32_t i = (wid * kR + j) * 64 + lane;
    const uint32_t d[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
    const uint32_t mine_ex = in[j] - nt[j];              // this lane's ties of the round come behind these (packed)
    uint32_t at0 = run0 + (mine_ex & 0xFFFFu), at1 = run1 + (mine_ex >> 16);
    const uint32_t round_tot = __builtin_amdgcn_readlane(in[j], 63);
    run0 += round_tot & 0xFFFFu;
    run1 += round_tot >> 16;
    // (a lane with nothing above the bound and all of its ties behind the quota -- nearly every lane of a window of
    // thousands of ties -- has nothing to admit)
    if (!((up_rounds >> j) & 1u) && !((nt[j] & 0xFFFFu) && at0 < quota) && !(kNib && (nt[j] >> 16) && at1 < quota)) continue;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      uint32_t up, tie;
      masks(d[k], up, tie);
      // ties in rank order: inside a word the byte positions DESCEND with the rank (one window: the nibbles ascend)
      uint32_t m = at0 < quota ? tie & kEven : 0u;       // (behind the quota: nothing of this lane's any more)
      while (m) {
        const uint32_t bit = kNib1 ? __ffs(m) - 1u : 31u - __clz(m);
        m &= ~(1u << bit);
        if (at0++ < quota) {
          const uint32_t pos = bit / P::kBits;
          admit((d[k] >> (pos * P::kBits)) & P::kMask, S::rank_of(wbase, (i * 4 + k) * P::kPerWord + pos));
        }
      }
      if (kNib) {
        m = at1 < quota ? tie & kOdd : 0u;
        while (m) {
          const uint32_t bit = kNib1 ? __ffs(m) - 1u : 31u - __clz(m);
          m &= ~(1u << bit);
          if (at1++ < quota) {
            const uint32_t pos = bit / P::kBits;
            admit((d[k] >> (pos * P::kBits)) & P::kMask, S::rank_of(wbase, (i * 4 + k) * P::kPerWord + pos));
          }
        }
      }
      while (up) {
        const uint32_t bit = __ffs(up) - 1;
        up &= up - 1;
        const uint32_t pos = bit / P::kBits;
        admit((d[k] >> (pos * P::kBits)) & P::kMask, S::rank_of(wbase, (i * 4 + k) * P::kPerWord + pos));
      }
    }
  }
  __syncthreads();
  if (tid < 12) sh->tally[tid] = 0;
}

// one step of a workgroup's windows: windows [w, w_end) of ONE pair (4-bit counters: both windows of the pair count
// in one byte's nibbles, wbase is the even window's) or the one window w (byte counters)
template <typename CT, int NT>
__device__ __forceinline__ void one_step(const FindArgs& A, const uint32_t T, const uint32_t code, uint32_t* cnt32,
                                         unsigned long long* pool, Control* ctl, OneShared* sh, const uint32_t w,
                                         const uint32_t w_end) {
  constexpr bool kNib = std::is_same<CT, Nib>::value;      // (Nib1: one window, like the byte counters)
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t w_even = kNib ? (w & ~1u) : w;
  uint32_t a0 = 0, b0 = 0, a1 = 0, b1 = 0;
  if (lane < T) {
    if (!kNib || w == w_even) { const uint2 se = A.slice_se[size_t(w_even) * kNumCodes + code]; a0 = se.x; b0 = se.y; }
    if (kNib && w_even + 1 >= w && w_even + 1 < w_end) {
      const uint2 se = A.slice_se[size_t(w_even + 1) * kNumCodes + code]; a1 = se.x; b1 = se.y;
    }
  }
  if ((__ballot(b0 > a0) | __ballot(b1 > a1)) == 0) return;     // nothing of the needle in these windows (uniform)
  ONE_MARK(A, 2);
  const uint32_t wbase = w_even * kWindowRanks;
  const uint32_t wlen = min((kNib ? 2u : 1u) * kWindowRanks, A.n_refs - wbase);
  one_count<CT, NT>(A.ent, cnt32, a0, b0, a1, b1, wid, lane);
  ONE_MARK(A, 3);
  __syncthreads();
  ONE_MARK(A, 4);
  one_select<CT, NT>(A, reinterpret_cast<uint4*>(cnt32), T, A.keep, ctl, pool, sh, wbase, wlen);
  ONE_MARK(A, 5);
  // keep the best `keep` of what the pool holds now (at most twice that); the threshold follows
  compact_pool<NT>(pool, ctl, kOnePool, A.keep);
}

struct OneArgs {
  uint16_t codes[kOneMaxNeedles][64];   // a needle's distinct trigram codes, ascending (tokeniser.c:59-119, by the host)
  uint32_t T[kOneMaxNeedles];           // how many
  uint32_t per;                  // windows per workgroup
  unsigned long long* part_keys; // [needle][grid.x * keep]: a workgroup's best keys ascending, padded with kKeyInf
  uint32_t* flags;               // [needle][grid.x] workgroup g's list is complete: the launch's sequence word
  trigram_match_t* out_rows;     // host-coherent pinned memory: [needle][kOneMaxKeep]
  uint32_t* out_count;           //   ... [needle][2]: rows, the sequence word (written last)
  uint32_t seq;
};

// find_one_kernel's last workgroup: the best `keep` of the G lists of `keep` slots (its own, the last, still in its pool)
// into `pool` (the head of s_counters), unsorted.
// A list is sorted, a key is level << 32 | rank with level = T - matches, and list g's windows lie in front of list
// g + 1's: LOWER ranks.  The union's order is therefore: by level, inside a level by list, inside a list by place -- and
// its best `keep` are every key of a level below L, the first level at which the levels' sizes add up to `keep`, plus
// the first keep - (keys below L) keys of level L in list order: of every list a PREFIX.  Nothing is sorted to find them:
//   * every slot's LEVEL goes to LDS as a byte (the lists' padding, kKeyInf, counts as one more level): all of a
//     thread's loads of a round in flight at once, one global round trip per 8 192 slots;
//   * for every (list, level l = 1 .. T) the place where the list's levels reach l -- a binary search over the list's
//     bytes, a wave per level, a lane per list; their sum over the lists is the number of keys BELOW level l, which
//     gives L, and the places themselves say where a list's keys of level L start and end: work per (list, level),
//     not per slot (a walk over the slots, however lean, costs the VALU 12 900 x 25 instructions at Geonames scale and
//     limit 100: 2 us a pass);
//   * one wave's prefix sums over the lists turn the places into the length of every list's passing prefix and its
//     offset in the output; output slot i finds its (list, place) by a binary search over the offsets and loads that key.
// Exactly min(keep, keys there are) keys arrive, and are put in order by counting, for every key, the keys below it --
// eight threads per key, each over an eighth of the keys, all of its reads in flight at once (compact_pool's rank count,
// made for pools that change under a sweep, walks them in a dependent loop of two keys a read: 4 us for 100 keys on
// the two waves that hold them).
// (Through round 5's first version: the keys not above the keep-th smallest list head, and of those the ones at place p
// of a list with r smaller heads in front where r + p < keep -- up to keep (keep + 1) / 2 keys, sorted by compact_pool's
// bitonic network: 32 us at Geonames scale and limit 100, 64 us on a haystack of massive ties, of a find's 70 and 104.
// Before that, one wave advancing one list per round: a dependent load from memory per row.  And a version of THIS
// merge with a thread's slots in registers, an unrolled loop over the 30 of the largest grid times the largest limit
// with a uniform skip per unused slot: every jump was an instruction-cache miss, 3 us at limit 10 -- this code runs
// once per launch, cold: its loops are rolled.)
template <int NT>
__device__ __forceinline__ void one_merge(const FindArgs& A, const unsigned long long* part_keys, uint32_t* s_counters,
                                          const unsigned long long* s_pool, Control* ctl, const uint32_t nres,
                                          const uint32_t T, const uint32_t keep, const uint32_t g, const uint32_t G) {
  constexpr uint32_t kPadLevel = 64;                                   // (a needle's levels: 0 .. T - 1 <= 63)
  constexpr uint32_t kRound = 8, kNW = NT / 64;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  uint32_t* const w32 = s_counters;                                    // (the counters are done with)
  unsigned long long* const pool = reinterpret_cast<unsigned long long*>(w32);   // [kOnePool]
  uint32_t* const below = w32 + 2 * kOnePool;                          // [128] keys of a level below l, l = 1 .. T
  uint32_t* const off = below + 128;                                   // [kOneMaxGrid] where a list's passing prefix starts in the output
  uint32_t* const place = off + kOneMaxGrid;                           // [128] an arriving key's place in the output order
  unsigned char* const reach = reinterpret_cast<unsigned char*>(place + 128);   // [65][kOneMaxGrid] the place where a list's levels reach l
  unsigned char* const lvl = reach + 65 * kOneMaxGrid;                 // [G * keep] a slot's level
  static_assert((2 * kOnePool + 128 + kOneMaxGrid + 128) * 4 + 65 * kOneMaxGrid + kOneMaxGrid * kOneMaxKeep <= kWindowSize &&
                kOnePool >= kOneMaxKeep && kOneMaxKeep <= 128 && NT % 128 == 0, "merge scratch");
  const uint32_t own0 = g * keep, total = G * keep;                    // (this workgroup's own list -- the last -- comes from its pool)
  if (tid == 0) { ctl->pool_n = 0; ctl->overflow = 0; ctl->thr = kKeyInf; ctl->adm_n = 0; }
  if (tid < 128) place[tid] = 0;
  auto slot = [&](const uint32_t idx) -> unsigned long long {
    if (idx < own0) return __hip_atomic_load(&part_keys[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return idx - own0 < nres ? s_pool[idx - own0] : kKeyInf;
  };
#pragma unroll 1
  for (uint32_t base = tid; base < total; base += kRound * NT) {       // (base - tid: uniform)
    unsigned long long k[kRound];
#pragma unroll
    for (uint32_t j = 0; j < kRound; ++j) {
      const uint32_t idx = base + j * NT;
      k[j] = kKeyInf;
      if (idx < total) k[j] = slot(idx);
    }
#pragma unroll
    for (uint32_t j = 0; j < kRound; ++j) {
      const uint32_t idx = base + j * NT;
      if (idx < total) lvl[idx] = static_cast<unsigned char>(min(uint32_t(k[j] >> 32), kPadLevel));
    }
  }
  __syncthreads();                                       // the levels are there
  ONE_MARK(A, 10);
#pragma unroll 1
  for (uint32_t l = 1 + wv; l <= T; l += kNW) {          // a wave per level (uniform), a lane per list
    uint32_t sum = 0;
#pragma unroll 1
    for (uint32_t g0 = 0; g0 < G; g0 += 64) {
      const uint32_t li = g0 + lane;
      uint32_t lo = 0;
      if (li < G) {
        const unsigned char* const row = lvl + li * keep;
        uint32_t n = keep;
        while (n) {                                      // the first place whose level is not below l
          const uint32_t half = n >> 1;
          if (row[lo + half] < l) { lo += half + 1; n -= half + 1; } else { n = half; }
        }
        reach[l * kOneMaxGrid + li] = static_cast<unsigned char>(lo);
      }
      sum += lo;
    }
    sum = wave_inclusive_sum(sum);
    if (lane == 63) below[l] = sum;                      // (this wave's alone)
  }
  __syncthreads();
  // every wave for itself: lane j holds the keys of levels up to j
  const uint32_t upto = lane < T ? below[lane + 1] : 0u;
  const unsigned long long reached = __ballot(lane < T && upto >= keep);
  const uint32_t L = reached ? uint32_t(__builtin_ctzll(reached)) : kPadLevel;      // (kPadLevel: fewer than keep keys in all, every one passes)
  const uint32_t quota = L < kPadLevel ? keep - (L ? __builtin_amdgcn_readlane(upto, L - 1) : 0u) : 0u;
  ONE_MARK(A, 11);
  if (tid < 64) {                                        // lane t: lists 4 t .. 4 t + 3
    uint32_t at[4], cnt[4], sum = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
      const uint32_t li = 4 * tid + i;
      at[i] = 0; cnt[i] = 0;
      if (li < G) {
        at[i] = L == kPadLevel ? reach[T * kOneMaxGrid + li] : (L ? reach[L * kOneMaxGrid + li] : 0u);
        if (L != kPadLevel) cnt[i] = reach[(L + 1) * kOneMaxGrid + li] - at[i];
      }
      sum += cnt[i];
    }
    uint32_t in_front = wave_inclusive_sum(sum) - sum;   // keys of level L in the lists in front
    sum = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {                   // a list's passing prefix: its keys below L, and of level L what the quota still wants
      at[i] += min(cnt[i], quota > in_front ? quota - in_front : 0u);
      in_front += cnt[i];
      sum += at[i];
    }
    uint32_t o = wave_inclusive_sum(sum) - sum;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) { off[4 * tid + i] = o; o += at[i]; }
    if (tid == 63) ctl->pool_n = o;
  }
  __syncthreads();
  const uint32_t n_out = ctl->pool_n;
  if (tid < n_out) {                                     // the list whose prefix holds output slot tid: the last with off <= tid
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t step = kOneMaxGrid / 2; step; step >>= 1)
      if (off[lo + step] <= tid) lo += step;
    pool[tid] = slot(lo * keep + (tid - off[lo]));
  }
  __syncthreads();
  ONE_MARK(A, 15);
  {
    constexpr uint32_t kParts = NT / 128, kShare = 128 / kParts;
    const uint32_t i = tid & 127u, first = (tid >> 7) * kShare;
    const unsigned long long mine = i < n_out ? pool[i] : kKeyInf;
    uint32_t lower = 0;
#pragma unroll
    for (uint32_t j = 0; j < kShare; ++j)
      if (first + j < n_out) lower += pool[first + j] < mine ? 1u : 0u;   // (one address per wave and read: broadcasts)
    if (i < n_out && lower) atomicAdd(&place[i], lower);
    __syncthreads();
    const unsigned long long key = tid < n_out ? pool[tid] : kKeyInf;
    const uint32_t to = tid < n_out ? place[tid] : 0u;
    __syncthreads();
    if (tid < n_out) pool[to] = key;                     // (keys are distinct: the places are a permutation)
    __syncthreads();
  }
}

template <int NT>
__global__ __launch_bounds__(NT) void find_one_kernel(const FindArgs A, const OneArgs O) {
  __shared__ __attribute__((aligned(16))) uint32_t s_counters[kWindowSize / 4];
  __shared__ __attribute__((aligned(16))) unsigned long long s_pool[kOnePool];
  __shared__ Control s_ctl;
  __shared__ OneShared s_sh;
  uint4* const cnt128 = reinterpret_cast<uint4*>(s_counters);
  Control* const ctl = &s_ctl;
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t g = blockIdx.x, G = gridDim.x;
  ONE_MARK(A, 0);
  // a row of the grid per needle (a handful of needles share the launch: blockIdx.y); its lists, flags and rows
  const uint32_t nd_i = blockIdx.y;
  const uint32_t T = O.T[nd_i];
  unsigned long long* const part_keys = O.part_keys + size_t(nd_i) * G * A.keep;
  uint32_t* const flags = O.flags + size_t(nd_i) * G;
  trigram_match_t* const out_rows = O.out_rows + size_t(nd_i) * kOneMaxKeep;
  uint32_t* const out_count = O.out_count + 2 * nd_i;
  // the needle's codes: lane t = trigram t (a load from the kernel-argument segment)
  const uint32_t code = lane < T ? O.codes[nd_i][lane] : 0u;
  for (uint32_t i = tid; i < kWindowSize / 16; i += NT) cnt128[i] = make_uint4(0, 0, 0, 0);
  if (tid == 0) { ctl->pool_n = 0; ctl->overflow = 0; ctl->thr = kKeyInf; ctl->adm_n = 0; ctl->pend_n[0] = 0; ctl->pend_n[1] = 0; }
  if (tid < 12) s_sh.tally[tid] = 0;
  __syncthreads();
  ONE_MARK(A, 1);
  const uint32_t w0 = g * O.per, w1 = min(A.n_windows, w0 + O.per);
  for (uint32_t w = w0; w < w1;) {
    // 4-bit counters, both windows of a pair per step: any window for a needle of at most 15 trigrams, and for ANY
    // needle the leading windows whose references have at most 15 (nib_windows: an even count) -- find_kernel's rule
    if (T <= 15 || w < A.nib_windows) {
      const uint32_t w_end = min(w1, (w | 1u) + 1u);
      if (w_end - w == 1) one_step<Nib1, NT>(A, T, code, s_counters, s_pool, ctl, &s_sh, w, w_end);   // (a single window: 32 KiB)
      else one_step<Nib, NT>(A, T, code, s_counters, s_pool, ctl, &s_sh, w, w_end);
      w = w_end;
    } else {
      one_step<uint8_t, NT>(A, T, code, s_counters, s_pool, ctl, &s_sh, w, w + 1);
      ++w;
    }
  }
  ONE_MARK(A, 7);
  const uint32_t nres = ctl->pool_n;                     // (sorted: every step ends with a compaction)
  // ---- this workgroup's best keys (the list padded to `keep` slots, so that nobody needs its length), then its
  // flag; the grid's LAST workgroup waits for everybody's flag and merges ------------------------------------------
  // Keys and flags are written THROUGH to memory (agent-scope atomic stores: an XCD's L2 is not coherent with the
  // others') and the flag goes out when the keys' stores have been acknowledged; the merging workgroup reads both the
  // same way.  (First version: plain stores, __threadfence() -- a write-back of the XCD's whole L2, the sixteen
  // workgroups of an XCD one after the other: 14 us -- and a ticket, atomicAdd on one word by 129 workgroups, performed
  // at the memory side one after the other: another 13 us.  A workgroup that only spins on flags needs every other
  // workgroup to RUN, not to be co-resident: the grid never exceeds the CUs, and nobody waits for the spinner.)
  const uint32_t keep = A.keep;
  if (g != G - 1) {
    for (uint32_t i = tid; i < keep; i += NT)
      __hip_atomic_store(&part_keys[size_t(g) * keep + i], i < nres ? s_pool[i] : kKeyInf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ONE_MARK(A, 8);
    if (tid == 0) __hip_atomic_store(&flags[g], O.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  ONE_MARK(A, 8);
  {
    uint32_t pending;
    do {                                                 // lanes 0..G-2 of the first waves watch one flag each
      const uint32_t f = tid < G - 1 ? __hip_atomic_load(&flags[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : O.seq;
      pending = __syncthreads_or(f != O.seq);
    } while (pending);
  }
  ONE_MARK(A, 9);
  one_merge<NT>(A, part_keys, s_counters, s_pool, ctl, nres, T, keep, g, G);   // (leaves the keys in order, ctl->pool_n of them)
  unsigned long long* const pool = reinterpret_cast<unsigned long long*>(s_counters);
  ONE_MARK(A, 12);
  const uint32_t n_out = ctl->pool_n;
  if (tid < n_out) {
    const unsigned long long key = pool[tid];
    const uint32_t rk = uint32_t(key);
    trigram_match_t row;
    row.reference = A.ref_of_rank[rk];
    row.matches = T - uint32_t(key >> 32);
    row.weight = A.weight_of_rank[rk];
    out_rows[tid] = row;
  }
  if (tid == 0) out_count[0] = n_out;
  ONE_MARK(A, 13);
  if (tid < ((n_out + 63u) & ~63u) || tid < 64) __threadfence_system();   // rows and count are in host memory before the sequence word
  __syncthreads();
  ONE_MARK(A, 14);
  if (tid == 0) __hip_atomic_store(&out_count[1], O.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// keys -> rows, in place: a needle's keys occupy the first 8 bytes of every 12-byte row slot's
// worth of its result area (2 words per key, packed), so rows are written from the last to the
// first -- row i lands on keys >= i only.  One lane per needle.
__global__ void finalize_rows_kernel(const FindArgs A, const uint32_t n) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const uint32_t T = A.q_ntri[q];
  if (T > 64 || A.q_nb[q] == 0) return;                    // not a needle of the window-major sweep
  const uint32_t cnt = A.counts[q];
  uint32_t* st = reinterpret_cast<uint32_t*>(A.results + size_t(q) * A.limit);
  for (uint32_t i = cnt; i-- > 0;) {
    const uint32_t rank = st[2 * i], miss = st[2 * i + 1];
    const uint32_t ref = A.ref_of_rank[rank], weight = A.weight_of_rank[rank];
    st[3 * i] = ref; st[3 * i + 1] = T - miss; st[3 * i + 2] = weight;
  }
}

// dynamic LDS of find_kernel (the counters are static)
size_t find_dynamic_lds_bytes(uint32_t pool_cap) {
  static_assert(sizeof(Control) <= 64, "the unit ring sits 64 bytes behind the control block");
  return size_t(pool_cap) * 8 + 2 * kCodeChunk * 4 + 64 + sizeof(UnitRing) + 2 * size_t(ring_units_for(pool_cap)) * 8 +
         (pool_cap <= 1024 ? 2 * kPendMax * 4 : 0) + 16;  // (pending lists: only where slices can be left out)
}
size_t find_lds_bytes(size_t counter_bytes, uint32_t pool_cap) {
  return size_t(kWindowSize) * counter_bytes + find_dynamic_lds_bytes(pool_cap);
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
  if (t.n <= 16384 && t.max_len && t.max_len <= 63) {          // short needles, not a huge batch: a wave per needle
    hipLaunchKernelGGL(tokenise_wave_kernel, dim3(t.n), dim3(64), 0, stream, t.packed, t.offsets, t.n, t.code_total,
                       t.qcodes, t.q_ntri, t.q_nb, t.start_win, t.q_start);
    BLURRILY_HIP_TRY(hipGetLastError());
    return 0;
  }
  const uint32_t block = 128;
  const uint32_t grid = (t.n + block - 1) / block;
  hipLaunchKernelGGL(tokenise_kernel, dim3(grid), dim3(block), 0, stream, t.packed, t.offsets, t.n,
                     t.code_total, t.qcodes, t.q_ntri, t.q_nb, t.big_list, t.big_count, t.mid_list, t.mid_count,
                     t.start_win, t.q_start);
  BLURRILY_HIP_TRY(hipGetLastError());
  return 0;
}

int launch_normalise(const char* in, const uint64_t* offsets, uint32_t n, char* out, uint32_t* non_ascii,
                     hipStream_t stream) {
  if (n == 0) return 0;
  const uint32_t block = 128;
  hipLaunchKernelGGL(normalise_kernel, dim3((n + block - 1) / block), dim3(block), 0, stream, in, offsets, n, out,
                     non_ascii);
  BLURRILY_HIP_TRY(hipGetLastError());
  return 0;
}

// whether a pass that keeps `keep` rows can leave slices out of the needle-major count (sweep_role: coop_can_leave): the
// pool's tail has to hold what a glance lets pass beside a step's pending list
bool find_can_leave(uint32_t keep) {
  const uint32_t cap = find_pool_cap(keep);
  const uint32_t sel = std::min(keep + std::max(6u, keep / 2), cap / 2);      // select_at()
  return cap <= 1024 && sel + 32 <= adm_max(cap);
}

uint32_t find_pool_cap(uint32_t keep) {
  // (the small pool's spare LDS is the unit ring's; at limit 100 the ring gains nothing and 512 entries cost
  // 1.5 % in compactions, configs[4])
  uint32_t cap = keep <= 64 ? 512 : 1024;
  while (cap < 4 * keep && cap < 4096) cap <<= 1;
  return cap;
}

// A kernel's dynamic-LDS ceiling is set once per device (a process may hold maps on several).
static bool first_launch_on_this_device(std::atomic<uint64_t>& seen) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  const uint64_t bit = 1ull << dev;
  return (seen.fetch_or(bit) & bit) == 0;
}

template <typename CT, int NT, bool RANGED, bool SHORT, bool LEAVE = false>
static int launch_find_tr(const FindArgs& a, uint32_t grid, hipStream_t stream) {
  const size_t lds = find_dynamic_lds_bytes(a.pool_cap);
  static std::atomic<uint64_t> attr_done{0};
  if (first_launch_on_this_device(attr_done))
    BLURRILY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&find_kernel<CT, NT, RANGED, SHORT, LEAVE>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024 - int(kWindowSize * sizeof(CT))));   // static counters
  hipLaunchKernelGGL((find_kernel<CT, NT, RANGED, SHORT, LEAVE>), dim3(grid), dim3(NT), lds, stream, a);
  BLURRILY_HIP_TRY(hipGetLastError());
  return 0;
}

template <typename CT, int NT>
static int launch_find_t(const FindArgs& a, uint32_t grid, hipStream_t stream) {
  if constexpr (sizeof(CT) == 1) {
    if (a.short_only) {
      // (slices can be left out of a step's count: a limit of at most 64 -- the 512-entry pool --, not phase 1 of the
      // window-major sweep; sweep_role's coop_can_leave asks the same)
      const bool leave = a.nm_cmin != 0 && a.pool_cap <= 1024 && !a.own_only && find_can_leave(a.keep);
      if (a.ranges > 1) return leave ? launch_find_tr<CT, NT, true, true, true>(a, grid, stream)
                                     : launch_find_tr<CT, NT, true, true>(a, grid, stream);
      return leave ? launch_find_tr<CT, NT, false, true, true>(a, grid, stream)
                   : launch_find_tr<CT, NT, false, true>(a, grid, stream);
    }
  }
  if (a.ranges > 1) return launch_find_tr<CT, NT, true, false>(a, grid, stream);
  return launch_find_tr<CT, NT, false, false>(a, grid, stream);
}

int launch_merge_parts(const FindArgs& a, uint32_t n_items, hipStream_t stream) {
  if (n_items == 0) return 0;
  if (a.keep <= 64 && a.ranges <= 256) {
    hipLaunchKernelGGL(merge_parts_small_kernel, dim3(n_items), dim3(64), 0, stream, a);
    BLURRILY_HIP_TRY(hipGetLastError());
    return 0;
  }
  constexpr int NT = 1024;                                    // keep <= 1024 keys per range
  const size_t lds = size_t(a.pool_cap) * 8 + sizeof(Control) + 16;
  static std::atomic<uint64_t> attr_done{0};
  if (first_launch_on_this_device(attr_done))
    BLURRILY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&merge_parts_kernel<NT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL((merge_parts_kernel<NT>), dim3(n_items), dim3(NT), lds, stream, a);
  BLURRILY_HIP_TRY(hipGetLastError());
  return 0;
}

int launch_wsweep(const FindArgs& a, uint32_t w, uint32_t n, uint32_t n_cus, bool own_pass, hipStream_t stream) {
  if (n == 0) return 0;
  // needles per queue pop: whole 256-thread filters for big batches, smaller chunks when there are too
  // few needles to give every resident workgroup (four per CU) several chunks
  uint32_t chunk_len = kWsChunk;
  while (chunk_len > 16 && (n + chunk_len - 1) / chunk_len < n_cus * 4u * 4u) chunk_len >>= 1;
  const uint32_t chunks = (n + chunk_len - 1) / chunk_len;
  const uint32_t grid = std::min(chunks, n_cus * 4u);
  hipLaunchKernelGGL(wsweep_kernel, dim3(grid), dim3(kWsNT), 0, stream, a, w, n, chunk_len, own_pass ? 1u : 0u);
  BLURRILY_HIP_TRY(hipGetLastError());
  return 0;
}

int launch_find_small(const FindArgs& a, uint32_t n_cus, hipStream_t stream) {
  if (a.n_work == 0) return 0;
  const uint32_t grid = std::min(a.n_work, n_cus * 4u);       // four workgroups per CU (LDS: 35 KiB each)
  hipLaunchKernelGGL(find_small_kernel, dim3(grid), dim3(kWsNT), 0, stream, a);
  BLURRILY_HIP_TRY(hipGetLastError());
  return 0;
}

int launch_find_one(const FindArgs& a, const uint16_t* codes, const uint32_t* T, uint32_t n_needles, uint32_t per, uint32_t grid,
                    unsigned long long* part_keys, uint32_t* flags, trigram_match_t* out_rows,
                    uint32_t* out_count, uint32_t seq, hipStream_t stream) {
  OneArgs o;
  for (uint32_t i = 0; i < kOneMaxNeedles; ++i) {
    o.T[i] = i < n_needles ? T[i] : 0u;
    for (uint32_t t = 0; t < 64; ++t) o.codes[i][t] = i < n_needles && t < T[i] ? codes[i * 64 + t] : uint16_t(0);
  }
  o.per = per; o.part_keys = part_keys; o.flags = flags;
  o.out_rows = out_rows; o.out_count = out_count; o.seq = seq;
  // ONE needle: its workgroups have their CUs to themselves (24 KiB of dynamic LDS nobody touches keep a second one off:
  // the configuration the 31 us were measured in); a handful of needles fill the chip twice over: two per CU
  constexpr size_t kOneAloneLds = 24 * 1024;
  static std::atomic<uint64_t> attr_done{0};
  if (first_launch_on_this_device(attr_done))
    BLURRILY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&find_one_kernel<kOneThreads>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, int(kOneAloneLds)));
  hipLaunchKernelGGL((find_one_kernel<kOneThreads>), dim3(grid, n_needles), dim3(kOneThreads), n_needles == 1 ? kOneAloneLds : 0, stream, a, o);
  BLURRILY_HIP_TRY(hipGetLastError());
  return 0;
}

int launch_finalize_rows(const FindArgs& a, uint32_t n, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(finalize_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a, n);
  BLURRILY_HIP_TRY(hipGetLastError());
  return 0;
}

int launch_merge_rows(const trigram_match_t* a_rows, const uint32_t* a_counts, const trigram_match_t* b_rows,
                      const uint32_t* b_counts, uint32_t n, uint32_t limit, trigram_match_t* out,
                      uint32_t* out_counts, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(merge_rows_kernel, dim3((n + 127) / 128), dim3(128), 0, stream, a_rows, a_counts, b_rows,
                     b_counts, n, limit, out, out_counts);
  BLURRILY_HIP_TRY(hipGetLastError());
  return 0;
}

int find_threads() { return kFindThreads; }

uint32_t find_wgs_per_cu() {
  const uint32_t by_lds = uint32_t((160 * 1024) / find_lds_bytes(1, 1024));
  const uint32_t by_waves = 32u / uint32_t(kFindThreads / 64);
  return std::max(1u, std::min(by_lds, by_waves));
}

int launch_find(const FindArgs& a, bool long_needles, uint32_t grid, hipStream_t stream) {
  if (grid == 0) return 0;
  if (!long_needles) return launch_find_t<uint8_t, kFindThreads>(a, grid, stream);
  return launch_find_t<uint16_t, kFindThreads>(a, grid, stream);
}

BLURRILY_KERNELS_END
}  // namespace blurrily
