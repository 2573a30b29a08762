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

// The kernels themselves, by sweep (one translation unit: they share the counters, the pool and the slice walks; the
// counted build -- find_kernels_counted.hip -- compiles all of it once more in namespace blurrily::counted):
#include "kernels/tokenise.inc"
#include "kernels/counters.inc"
#include "kernels/pipelined.inc"
#include "kernels/needle_major.inc"
#include "kernels/merge.inc"
#include "kernels/window_major.inc"
#include "kernels/small.inc"
#include "kernels/one.inc"

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

template <typename CT, int NT, bool RANGED, bool SHORT>
static int launch_find_tr(const FindArgs& a, uint32_t grid, hipStream_t stream) {
  const size_t lds = find_dynamic_lds_bytes(a.pool_cap);
  static std::atomic<uint64_t> attr_done{0};
  if (first_launch_on_this_device(attr_done))
    BLURRILY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&find_kernel<CT, NT, RANGED, SHORT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024 - int(kWindowSize * sizeof(CT))));   // static counters
  hipLaunchKernelGGL((find_kernel<CT, NT, RANGED, SHORT>), dim3(grid), dim3(NT), lds, stream, a);
  BLURRILY_HIP_TRY(hipGetLastError());
  char name[64];
  std::snprintf(name, sizeof name, "find_kernel<%s,%d,%s,%s>", sizeof(CT) == 1 ? "uint8_t" : "uint16_t", NT,
                RANGED ? "true" : "false", SHORT ? "true" : "false");
  ::blurrily::note_launch(name);
  return 0;
}

template <typename CT, int NT>
static int launch_find_t(const FindArgs& a, uint32_t grid, hipStream_t stream) {
  if constexpr (sizeof(CT) == 1) {
    if (a.short_only)                  // (needles of up to 64 trigrams; whether slices are left out: coop_can_leave, at run time)
      return a.ranges > 1 ? launch_find_tr<CT, NT, true, true>(a, grid, stream) : launch_find_tr<CT, NT, false, true>(a, grid, stream);
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

int launch_merge_parts_pinned(const FindArgs& a, uint32_t n_items, trigram_match_t* rows, uint32_t* words, uint32_t seq,
                              hipStream_t stream) {
  if (n_items == 0) return 0;
  constexpr int NT = 1024;
  static_assert(kOneMaxKeep <= NT, "a thread per row");
  const size_t lds = size_t(a.pool_cap) * 8 + sizeof(Control) + 16;
  hipLaunchKernelGGL((merge_parts_pinned_kernel<NT>), dim3(n_items), dim3(NT), lds, stream, a, rows, words, seq);
  BLURRILY_HIP_TRY(hipGetLastError());
  ::blurrily::note_launch("merge_parts_pinned_kernel");
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
  ::blurrily::note_launch("wsweep_kernel");
  return 0;
}

int launch_find_small(const FindArgs& a, uint32_t n_cus, hipStream_t stream) {
  if (a.n_work == 0) return 0;
  const uint32_t grid = std::min(a.n_work, n_cus * 4u);       // four workgroups per CU (LDS: 35 KiB each)
  hipLaunchKernelGGL(find_small_kernel, dim3(grid), dim3(kWsNT), 0, stream, a);
  BLURRILY_HIP_TRY(hipGetLastError());
  ::blurrily::note_launch("find_small_kernel");
  return 0;
}

int launch_find_one(const FindArgs& a, const uint16_t* codes, const uint32_t* T, uint32_t n_needles, uint32_t per, uint32_t grid,
                    unsigned long long* part_keys, uint32_t* flags, trigram_match_t* out_rows,
                    uint32_t* out_count, uint32_t seq, hipStream_t stream, uint32_t n_cus, const uint16_t* codes_far,
                    const uint32_t* T_far, uint32_t* tickets) {
  OneArgs o;
  const bool far = codes_far != nullptr;               // (the needles' codes are read from host-coherent memory: see OneArgs)
  for (uint32_t i = 0; i < kOneMaxNeedles; ++i) {
    o.T[i] = !far && i < n_needles ? T[i] : 0u;
    for (uint32_t t = 0; t < 64; ++t) o.codes[i][t] = !far && i < n_needles && t < T[i] ? codes[i * 64 + t] : uint16_t(0);
  }
  o.codes_far = codes_far; o.T_far = T_far; o.tickets = tickets;
  o.per = per; o.part_keys = part_keys; o.flags = flags;
  o.out_rows = out_rows; o.out_count = out_count; o.seq = seq;
  // ONE needle: its workgroups have their CUs to themselves (24 KiB of dynamic LDS nobody touches keep a second one off:
  // the configuration the 31 us were measured in); a handful of needles fill the chip twice over: two per CU
  constexpr size_t kOneAloneLds = 24 * 1024;
  static std::atomic<uint64_t> attr_done{0};
  if (first_launch_on_this_device(attr_done))
    BLURRILY_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&find_one_kernel<kOneThreads>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, int(kOneAloneLds)));
  // (... as long as there is a CU for each: an image of more window pairs than CUs lets two share one rather than wait)
  const bool alone = n_needles == 1 && grid <= n_cus;
  hipLaunchKernelGGL((find_one_kernel<kOneThreads>), dim3(grid, n_needles), dim3(kOneThreads), alone ? kOneAloneLds : 0, stream, a, o);
  BLURRILY_HIP_TRY(hipGetLastError());
  ::blurrily::note_launch("find_one_kernel<1024>");
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
