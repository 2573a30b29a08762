// device_index.hip -- build and upload the HBM-resident index (device_index.h).
#include "device_index.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace blurrily {

namespace {

#define BLURRILY_HIP_TRY(expr)                                                        \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      std::fprintf(stderr, "blurrily_hip: %s failed: %s\n", #expr, hipGetErrorString(e_)); \
      errno = (e_ == hipErrorOutOfMemory) ? ENOMEM : EIO;                             \
      return -1;                                                                      \
    }                                                                                 \
  } while (0)

// Sorted view of a bucket: the bucket itself when already ascending, else a
// sorted scratch copy (the host bucket is left untouched: only find() and
// save() sort in place, like the reference).
const Entry* sorted_view(const Bucket& bk, std::vector<Entry>& scratch) {
  bool asc = true;
  for (uint32_t j = 1; j < bk.used; ++j)
    if (bk.e[j - 1].ref >= bk.e[j].ref) { asc = false; break; }
  if (asc) return bk.e;
  scratch.assign(bk.e, bk.e + bk.used);
  std::sort(scratch.begin(), scratch.end(),
            [](const Entry& l, const Entry& r) { return l.ref < r.ref; });
  return scratch.data();
}

// Sort with a pool of host threads: equal chunks sorted independently, then merged pairwise.
template <typename T, typename Less>
void parallel_sort(std::vector<T>& v, unsigned n_threads, Less less, bool stable) {
  const size_t n = v.size();
  if (n < (1u << 16) || n_threads < 2) {
    if (stable) std::stable_sort(v.begin(), v.end(), less); else std::sort(v.begin(), v.end(), less);
    return;
  }
  unsigned parts = 1;
  while (parts * 2 <= n_threads && parts < 64) parts *= 2;
  std::vector<size_t> cut(parts + 1);
  for (unsigned i = 0; i <= parts; ++i) cut[i] = n * i / parts;
  {
    std::vector<std::thread> pool;
    for (unsigned i = 0; i < parts; ++i)
      pool.emplace_back([&, i]() {
        if (stable) std::stable_sort(v.begin() + cut[i], v.begin() + cut[i + 1], less);
        else std::sort(v.begin() + cut[i], v.begin() + cut[i + 1], less);
      });
    for (auto& th : pool) th.join();
  }
  for (unsigned width = 1; width < parts; width *= 2) {     // std::inplace_merge is stable
    std::vector<std::thread> pool;
    for (unsigned i = 0; i + width < parts; i += 2 * width)
      pool.emplace_back([&, i, width]() {
        std::inplace_merge(v.begin() + cut[i], v.begin() + cut[i + width],
                           v.begin() + cut[std::min(parts, i + 2 * width)], less);
      });
    for (auto& th : pool) th.join();
  }
}

}  // namespace

void device_index_free(DeviceIndex* ix) {
  if (!ix) return;
  if (ix->d_ref_of_rank)    (void)hipFree(ix->d_ref_of_rank);
  if (ix->d_weight_of_rank) (void)hipFree(ix->d_weight_of_rank);
  if (ix->d_slice_se)       (void)hipFree(ix->d_slice_se);
  if (ix->d_ent)            (void)hipFree(ix->d_ent);
  if (ix->d_code_total)     (void)hipFree(ix->d_code_total);
  if (ix->d_win_max_tri)    (void)hipFree(ix->d_win_max_tri);
  if (ix->d_start_win)      (void)hipFree(ix->d_start_win);
  if (ix->d_tomb)           (void)hipFree(ix->d_tomb);
  *ix = DeviceIndex();
}

int device_index_build(const HostIndex& host, DeviceIndex* out, const IndexBuildOptions& opt) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    std::fprintf(stderr, "blurrily_hip: no usable HIP device (find has no CPU fallback)\n");
    errno = ENODEV;
    return -1;
  }
  int dev = 0;
  BLURRILY_HIP_TRY(hipGetDevice(&dev));
  // BLURRILY_BUILD_TRACE=1: wall time of every build stage on stderr
  const bool trace = build_trace();
  auto t_last = std::chrono::steady_clock::now();
  auto stage = [&](const char* what) {
    if (!trace) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "blurrily_hip: index build: %-28s %8.1f ms\n", what,
                 std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };

  unsigned n_threads = host_threads();

  // ---- 1. reference table: every live reference once, ascending ------------
  // Each string owns exactly one leading trigram "**c" (code 784*sym(c)), so
  // the 27 leading buckets list every reference exactly once.
  std::vector<Entry> refs;
  uint64_t lead = 0, nnz = 0;
  for (uint32_t s = 0; s + 1 < uint32_t(kBase); ++s) lead += host.bucket(s * kBase * kBase).used;
  for (uint32_t t = 0; t < kNumCodes; ++t) nnz += host.bucket(t).used;
  // a small image (the delta image of a few pending puts above all) is not worth a thread pool
  n_threads = unsigned(std::max<uint64_t>(1, std::min<uint64_t>(n_threads, nnz / 32768)));
  if (lead == host.total_refs()) {
    refs.reserve(lead);
    uint32_t hi_ref = 0;
    for (uint32_t s = 0; s + 1 < uint32_t(kBase); ++s) {
      const Bucket& bk = host.bucket(s * kBase * kBase);
      refs.insert(refs.end(), bk.e, bk.e + bk.used);
      for (uint32_t j = 0; j < bk.used; ++j) hi_ref = std::max(hi_ref, bk.e[j].ref);
    }
    if (!refs.empty() && uint64_t(hi_ref) < 4ull * refs.size() + 1024) {
      // references dense in their range (1..N, the usual case): place them instead of sorting them
      std::vector<uint32_t> weight_at(size_t(hi_ref) + 1);
      std::vector<uint8_t> seen(size_t(hi_ref) + 1, 0);
      for (const Entry& e : refs) {
        if (seen[e.ref]) { errno = EPROTO; return -1; }               // a reference listed twice
        seen[e.ref] = 1; weight_at[e.ref] = e.weight;
      }
      size_t k = 0;
      for (uint32_t r = 0; r <= hi_ref; ++r)
        if (seen[r]) { refs[k].ref = r; refs[k].weight = weight_at[r]; ++k; }
    } else {
      parallel_sort(refs, n_threads, [](const Entry& l, const Entry& r) { return l.ref < r.ref; }, false);
    }
  } else {
    refs.reserve(nnz);
    for (uint32_t t = 0; t < kNumCodes; ++t) {
      const Bucket& bk = host.bucket(t);
      refs.insert(refs.end(), bk.e, bk.e + bk.used);
    }
    parallel_sort(refs, n_threads, [](const Entry& l, const Entry& r) {
      return l.ref != r.ref ? l.ref < r.ref : l.weight < r.weight;
    }, false);
    refs.erase(std::unique(refs.begin(), refs.end(),
                           [](const Entry& l, const Entry& r) { return l.ref == r.ref && l.weight == r.weight; }),
               refs.end());
  }
  for (size_t i = 1; i < refs.size(); ++i)
    if (refs[i - 1].ref == refs[i].ref) { errno = EPROTO; return -1; }   // two weights / duplicate
  const uint64_t n_refs64 = refs.size();
  if (n_refs64 > 0xFFFFFFFFull || nnz > 0xFFFFFFF0ull) { errno = EPROTO; return -1; }
  const uint32_t n_refs = uint32_t(n_refs64);
  const uint32_t n_win  = std::max<uint32_t>(1u, uint32_t((n_refs64 + kWindowRanks - 1) / kWindowRanks));

  // ranks follow (weight, reference); sorted_ref / rank_of_pos map a reference to its rank
  std::vector<uint32_t> sorted_ref(n_refs), rank_of_pos(n_refs);
  std::vector<uint32_t> ref_of_rank(n_refs), weight_of_rank(n_refs);
  {
    std::vector<uint32_t> order(n_refs);
    uint32_t w_lo = 0xFFFFFFFFu, w_hi = 0;
    for (uint32_t i = 0; i < n_refs; ++i) {
      order[i] = i; sorted_ref[i] = refs[i].ref;
      w_lo = std::min(w_lo, refs[i].weight); w_hi = std::max(w_hi, refs[i].weight);
    }
    if (n_refs && uint64_t(w_hi) - w_lo < (1u << 20)) {
      // weights in a narrow range (string lengths, the default): a counting sort, stable by construction
      std::vector<uint32_t> at(size_t(w_hi - w_lo) + 2, 0);
      for (uint32_t i = 0; i < n_refs; ++i) at[refs[i].weight - w_lo + 1] += 1;
      for (size_t k = 1; k < at.size(); ++k) at[k] += at[k - 1];
      for (uint32_t i = 0; i < n_refs; ++i) order[at[refs[i].weight - w_lo]++] = i;
    } else {
      parallel_sort(order, n_threads,                              // refs[] is ref-ascending: a stable
                    [&](uint32_t l, uint32_t r) { return refs[l].weight < refs[r].weight; }, true);   // sort by weight
    }
    for (uint32_t rk = 0; rk < n_refs; ++rk) {
      const uint32_t pos = order[rk];
      rank_of_pos[pos] = rk;
      ref_of_rank[rk] = refs[pos].ref;
      weight_of_rank[rk] = refs[pos].weight;
    }
  }
  std::vector<Entry>().swap(refs);
  stage("reference table");

  // ---- 2. rank every posting, count per (window, code) ----------------------
  // Buckets are independent (a code's counts, cursors and slices are touched by one worker
  // only), so steps 2 and 3 run on a pool of host threads that pull codes from a shared counter.
  const uint64_t n_slices = uint64_t(n_win) * kNumCodes;
  if (n_slices + 1 > 0x7FFFFFFFull) { errno = ENOMEM; return -1; }
  std::vector<uint32_t> slice_off(n_slices + 1, 0);
  std::vector<uint32_t> rank(nnz);
  std::vector<uint32_t> code_total(kNumCodes);
  std::vector<uint64_t> bucket_base(kNumCodes + 1, 0);
  for (uint32_t t = 0; t < kNumCodes; ++t) {
    code_total[t] = host.bucket(t).used;
    bucket_base[t + 1] = bucket_base[t] + code_total[t];
  }
  std::atomic<int> failed{0};
  auto parallel_codes = [&](auto&& body) {
    std::atomic<uint32_t> next{0};
    auto worker = [&]() {
      for (;;) {
        const uint32_t t0 = next.fetch_add(64);
        if (t0 >= kNumCodes || failed.load()) return;
        for (uint32_t t = t0; t < std::min(t0 + 64, kNumCodes); ++t) body(t);
      }
    };
    std::vector<std::thread> pool;
    for (unsigned i = 1; i < n_threads; ++i) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
  };

  // References that are dense in their range (1..N, the usual case) are ranked through a direct table:
  // one load per posting instead of a galloping search (Geonames scale: 1.3 s -> a fifth of that).
  std::vector<uint32_t> rank_by_ref;
  const uint32_t max_ref = n_refs ? sorted_ref[n_refs - 1] : 0u;
  const bool dense_refs = n_refs > 0 && uint64_t(max_ref) < 4ull * n_refs + 1024;
  if (dense_refs) {
    rank_by_ref.assign(size_t(max_ref) + 1, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < n_refs; ++i) rank_by_ref[sorted_ref[i]] = rank_of_pos[i];
  }
  parallel_codes([&](uint32_t t) {
    const Bucket& bk = host.bucket(t);
    if (!bk.used) return;
    std::vector<Entry> scratch;
    const Entry* e = sorted_view(bk, scratch);
    uint32_t* out_rank = rank.data() + bucket_base[t];
    if (dense_refs) {
      for (uint32_t j = 0; j < bk.used; ++j) {
        const uint32_t ref = e[j].ref;
        const uint32_t rk = ref <= max_ref ? rank_by_ref[ref] : 0xFFFFFFFFu;
        // a reference missing from the table, listed twice in the bucket, or with a second weight
        if (rk == 0xFFFFFFFFu || (j > 0 && e[j - 1].ref >= ref) || weight_of_rank[rk] != e[j].weight) {
          failed.store(1);
          return;
        }
        out_rank[j] = rk;
        slice_off[uint64_t(rk / kWindowRanks) * kNumCodes + t + 1] += 1;
      }
      return;
    }
    uint64_t pos = 0;                         // references ascend inside a bucket: gallop from the last hit
    for (uint32_t j = 0; j < bk.used; ++j) {
      const uint32_t ref = e[j].ref;
      uint64_t lo = pos, step = 1;
      while (lo + step < n_refs && sorted_ref[lo + step] < ref) { lo += step; step <<= 1; }
      uint64_t hi = std::min<uint64_t>(lo + step, n_refs ? n_refs - 1 : 0);
      while (lo < hi) {                       // first position in [lo, hi] with sorted_ref >= ref
        const uint64_t mid = (lo + hi) >> 1;
        if (sorted_ref[mid] < ref) lo = mid + 1; else hi = mid;
      }
      // a reference missing from the table, listed twice in the bucket, or with a second weight
      if (lo >= n_refs || sorted_ref[lo] != ref || (j > 0 && lo < pos) ||
          weight_of_rank[rank_of_pos[lo]] != e[j].weight) {
        failed.store(1);
        return;
      }
      const uint32_t rk = rank_of_pos[lo];
      out_rank[j] = rk;
      slice_off[uint64_t(rk / kWindowRanks) * kNumCodes + t + 1] += 1;
      pos = lo + 1;
    }
  });
  if (failed.load()) { errno = EPROTO; return -1; }
  stage("rank postings");

  // pad every slice to a multiple of eight entries (16 bytes), a dense one gets its bitmap in front of its
  // postings (device_index.h), then prefix-sum
  uint64_t n_slots = 0;
  double mean_hit_slice = 0.0;
  {
    double sq = 0.0;
    for (uint32_t t = 0; t < kNumCodes; ++t) sq += double(code_total[t]) * double(code_total[t]);
    mean_hit_slice = nnz ? sq / double(nnz) / double(n_win) : 0.0;
  }
  // dense: the PADDED length reaches dense_min8 -- what a kernel can tell from the slice table alone
  uint32_t dense_min8 = (std::max(64u, std::min(opt.dense_min, kWindowSize)) + 7u) & ~7u;
  {
    // The bitmaps live inline in `ent`, whose slots are numbered in 32 bits: a low "dense_min" on a large map would
    // overflow that with bitmaps alone (8 KiB per dense slice).  Rather than fail every later find, the threshold is
    // raised for THIS build until the image fits (said on stderr; past a window's size no slice is dense).
    auto slots_with = [&](uint32_t dm8) {
      uint64_t t = 0;
      for (uint64_t i = 0; i < n_slices; ++i) {
        const uint32_t len = (slice_off[i + 1] + 7u) & ~7u;
        t += len + (len >= dm8 ? kBitmapSlots : 0u);
      }
      return t;
    };
    const uint32_t asked = dense_min8;
    while (dense_min8 <= kWindowSize && slots_with(dense_min8) > 0xFFFF0000ull) dense_min8 *= 2;
    if (dense_min8 != asked)
      std::fprintf(stderr, "blurrily_hip: \"dense_min\" %u would put the image past 2^32 slots: %u for this build\n", asked, dense_min8);
  }
  auto is_dense = [&](uint32_t len) { return ((len + 7u) & ~7u) >= dense_min8; };
  double dense_share = 0.0;
  {
    double all = 0.0, dense = 0.0;                             // (slice_off[i + 1] still holds slice i's length here)
    for (uint64_t i = 0; i < n_slices; ++i) {
      const double len = double(slice_off[i + 1]);
      all += len * len;
      if (is_dense(slice_off[i + 1])) dense += len * len;
    }
    dense_share = all > 0.0 ? dense / all : 0.0;
  }
  // What the window-major sweep can save: it leaves out the few LARGEST dense slices of a needle (at most
  // need - cmin of them, a handful), so what counts is how much of a needle's postings sits in the haystack's few
  // hottest trigrams -- mean_hit_slice minus the same figure without the kHotCodes biggest buckets -- times the
  // trigrams a needle as long as the haystack's strings has.  (A haystack whose postings are spread over
  // hundreds of middling buckets -- Geonames scale -- has a high mean_hit_slice and little to leave out.)
  double ws_gain = 0.0;
  if (nnz && n_refs) {
    constexpr size_t kHotCodes = 4;
    std::vector<uint32_t> top(code_total.begin(), code_total.end());
    std::partial_sort(top.begin(), top.begin() + kHotCodes, top.end(), std::greater<uint32_t>());
    double hot_sq = 0.0;
    for (size_t i = 0; i < kHotCodes; ++i) hot_sq += double(top[i]) * double(top[i]);
    ws_gain = hot_sq / double(nnz) / double(n_win) * (double(nnz) / double(n_refs));
  }
  uint32_t n_bitmaps = 0;
  for (uint64_t i = 0; i < n_slices; ++i) {
    uint32_t len = (slice_off[i + 1] + 7u) & ~7u;
    if (len >= dense_min8) { len += kBitmapSlots; ++n_bitmaps; }
    n_slots += len;
    if (n_slots > 0xFFFF0000ull) { errno = EPROTO; return -1; }
    slice_off[i + 1] = uint32_t(n_slots);
  }

  // ---- 3. scatter into window-major slices; deal every unit bank-aware ---------
  // The kernel bumps one packed byte counter per posting with an LDS atomic whose bank is
  // (rank >> 2) & 31; a wave instruction serialises lanes of one 32-lane half that hit the same
  // bank (46 % of the LDS-active cycles with postings in arbitrary order,
  // profiles/r01_pmc_lds.txt) or the same word.  Inside each unit (the 64 x 16 bytes one wave
  // loads; lane l's j-th element is bumped by instruction j) the postings are therefore dealt
  // in (bank, rank) order round-robin over the 16 instruction-halves, so one half sees each
  // bank -- and each counter word -- about once.
  std::vector<uint16_t> ent(n_slots + kEntPad, kPadRank);
  parallel_codes([&](uint32_t t) {
    const uint32_t used = code_total[t];
    if (!used) return;
    const uint32_t* rk = rank.data() + bucket_base[t];
    // where the postings of this code's slice of window w start (behind the bitmap of a dense slice, zeroed here)
    std::vector<uint32_t> fill(n_win, 0), post0(n_win, 0);
    std::vector<uint8_t> dense(n_win, 0);
    for (uint32_t w = 0; w < n_win; ++w) {
      const uint64_t i = uint64_t(w) * kNumCodes + t;
      post0[w] = slice_off[i];
      if (slice_off[i + 1] - slice_off[i] >= dense_min8) {                                // (a slice belongs to one worker: no race)
        dense[w] = 1;
        post0[w] += kBitmapSlots;
        std::fill(ent.begin() + slice_off[i], ent.begin() + slice_off[i] + kBitmapSlots, uint16_t(0));
      }
    }
    for (uint32_t j = 0; j < used; ++j) {
      const uint32_t w = rk[j] / kWindowRanks, r = rk[j] % kWindowRanks;
      ent[post0[w] + fill[w]++] = uint16_t(r);
      if (dense[w]) ent[post0[w] - kBitmapSlots + (r >> 4)] |= uint16_t(1u << (r & 15));   // little-endian: bit r of the u32 words
    }
    std::vector<uint16_t> tmp;
    for (uint32_t w = 0; w < n_win; ++w) {
      const uint32_t m = fill[w];
      if (m < 16) continue;                                   // a lane or two: nothing to arrange
      uint16_t* sl = ent.data() + post0[w];
      const uint32_t padded = (m + 7u) & ~7u;
      for (uint32_t u0 = 0; u0 < padded; u0 += 512) {         // one wave-load at a time
        const uint32_t len = std::min(512u, padded - u0), G = len / 8;
        const uint32_t real = std::min(len, m - u0);          // padding sentinels sit at the end
        tmp.assign(sl + u0, sl + u0 + real);
        std::sort(tmp.begin(), tmp.end(), [](uint16_t x, uint16_t y) {
          const uint32_t bx = (x >> 2) & 31u, by = (y >> 2) & 31u;
          return bx != by ? bx < by : x < y;
        });
        // 16 instruction-halves: h = 2*j + (lane >= 32); capacity = live lanes in that half
        uint32_t cap[16];
        for (uint32_t h = 0; h < 16; ++h) cap[h] = (h & 1) ? (G > 32 ? G - 32 : 0) : std::min(G, 32u);
        std::fill(sl + u0, sl + u0 + len, kPadRank);
        // (Dealt greedily instead -- half after half one posting from every bank that still has any, the fullest
        // banks first, doubles only when no bank is untouched; CPU model: 25.8 LDS cycles per full unit against 31.9
        // for the round robin below, 16 being the floor -- measured in round 2 on the configs[2] haystack, 300 k
        // needles: SQ_LDS_BANK_CONFLICT -37 %, SQ_LDS_IDX_ACTIVE -10 %, and the kernel 1 % SLOWER, 351.3 vs 347.7 ms
        // per 500 k needles, same box.  The code went with round 3.)
        uint32_t cnt[16];
        for (uint32_t h = 0; h < 16; ++h) cnt[h] = 0;
        uint32_t h = 0;
        for (uint32_t i = 0; i < real; ++i) {
          while (cnt[h] == cap[h]) h = (h + 1) & 15;
          const uint32_t lane = (h & 1) * 32 + cnt[h]++;
          sl[u0 + lane * 8 + (h >> 1)] = tmp[i];
          h = (h + 1) & 15;
        }
        // a live lane keeps a real rank in its first slot: the kernel tells a loaded group from
        // an idle lane's eight sentinels by that slot alone
        for (uint32_t l = 0; l < G; ++l) {
          uint16_t* g = sl + u0 + l * 8;
          if (g[0] != kPadRank) continue;
          for (uint32_t j = 1; j < 8; ++j)
            if (g[j] != kPadRank) { std::swap(g[0], g[j]); break; }
        }
      }
    }
  });
  stage("scatter + deal units");
  // what the kernels read: {start, end} of every slice's postings (a dense slice's bitmap sits in front of start)
  std::vector<uint2> slice_se(n_slices);
  for (uint64_t i = 0; i < n_slices; ++i) {
    const uint32_t a = slice_off[i], b = slice_off[i + 1];
    slice_se[i] = make_uint2(b - a >= dense_min8 ? a + kBitmapSlots : a, b);
  }
  std::vector<uint32_t>().swap(slice_off);
  // per-window bound and per-weight start window
  std::vector<uint32_t> win_max_tri(n_win, 0), start_win(256, n_win - 1);
  {
    std::vector<uint16_t> ntri(n_refs, 0);
    for (uint64_t i = 0; i < nnz; ++i) ntri[rank[i]] += 1;     // <= 19 683 codes per reference
    for (uint32_t r = 0; r < n_refs; ++r) {
      uint32_t& mx = win_max_tri[r / kWindowRanks];
      mx = std::max<uint32_t>(mx, ntri[r]);
    }
    uint32_t r = 0;
    for (uint32_t L = 0; L < 256; ++L) {
      while (r < n_refs && weight_of_rank[r] < L) ++r;
      start_win[L] = r < n_refs ? r / kWindowRanks : n_win - 1;
    }
  }
  std::vector<uint32_t>().swap(rank);
  stage("window bounds");

  // ---- 4. upload -------------------------------------------------------------
  DeviceIndex ix;
  ix.device = dev; ix.n_refs = n_refs; ix.n_windows = n_win; ix.n_entries = nnz; ix.n_slots = n_slots;
  ix.built_from = host.generation();
  ix.n_bitmaps = n_bitmaps;
  ix.dense_min8 = dense_min8;
  std::copy(start_win.begin(), start_win.end(), ix.h_start_win);
  ix.mean_hit_slice = mean_hit_slice;
  ix.dense_share = dense_share;
  ix.ws_gain = ws_gain;
  while (ix.nib_windows + 1 < n_win && win_max_tri[ix.nib_windows] <= 15 && win_max_tri[ix.nib_windows + 1] <= 15)
    ix.nib_windows += 2;
  auto up = [&](auto** dptr, const auto& v, size_t min_elems) -> int {   // v may be a temporary
    using T = typename std::remove_reference<decltype(v)>::type::value_type;
    const size_t bytes = std::max(v.size(), min_elems) * sizeof(T);
    BLURRILY_HIP_TRY(hipMalloc(reinterpret_cast<void**>(dptr), bytes));
    if (!v.empty()) BLURRILY_HIP_TRY(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    ix.device_bytes += bytes;
    return 0;
  };
  if (up(&ix.d_ref_of_rank, ref_of_rank, 1) || up(&ix.d_weight_of_rank, weight_of_rank, 1) ||
      up(&ix.d_slice_se, slice_se, 1) || up(&ix.d_ent, ent, 1) || up(&ix.d_code_total, code_total, 1) ||
      up(&ix.d_win_max_tri, win_max_tri, 1) || up(&ix.d_start_win, start_win, 1) ||
      up(&ix.d_tomb, std::vector<uint32_t>((size_t(n_refs) + 31) / 32 + 1, 0u), 1)) {
    const int e = errno;
    device_index_free(&ix);
    errno = e;
    return -1;
  }
  stage("upload");
  ix.h_sorted_ref.swap(sorted_ref);
  ix.h_rank_of_pos.swap(rank_of_pos);
  device_index_free(out);
  *out = std::move(ix);
  return 0;
}

int device_index_clone(const DeviceIndex& src, int dst_device, DeviceIndex* out) {
  int prev = 0;
  BLURRILY_HIP_TRY(hipGetDevice(&prev));
  DeviceIndex ix;
  ix.device = dst_device; ix.n_refs = src.n_refs; ix.n_windows = src.n_windows; ix.nib_windows = src.nib_windows;
  ix.n_entries = src.n_entries; ix.n_slots = src.n_slots; ix.built_from = src.built_from;
  ix.n_bitmaps = src.n_bitmaps; ix.dense_min8 = src.dense_min8;
  ix.mean_hit_slice = src.mean_hit_slice; ix.dense_share = src.dense_share; ix.ws_gain = src.ws_gain;
  bool failed = hipSetDevice(dst_device) != hipSuccess;
  auto copy = [&](auto** dptr, const auto* from, size_t elems) {
    using T = typename std::remove_const<typename std::remove_pointer<decltype(from)>::type>::type;
    const size_t bytes = std::max<size_t>(elems, 1) * sizeof(T);
    if (failed || hipMalloc(reinterpret_cast<void**>(dptr), bytes) != hipSuccess) { failed = true; return; }
    ix.device_bytes += bytes;
    if (elems && hipMemcpyPeer(*dptr, dst_device, from, src.device, elems * sizeof(T)) != hipSuccess) failed = true;
  };
  copy(&ix.d_ref_of_rank, src.d_ref_of_rank, src.n_refs);
  copy(&ix.d_weight_of_rank, src.d_weight_of_rank, src.n_refs);
  copy(&ix.d_slice_se, src.d_slice_se, size_t(src.n_windows) * kNumCodes);
  copy(&ix.d_ent, src.d_ent, size_t(src.n_slots) + kEntPad);
  copy(&ix.d_code_total, src.d_code_total, kNumCodes);
  copy(&ix.d_win_max_tri, src.d_win_max_tri, src.n_windows);
  copy(&ix.d_start_win, src.d_start_win, 256);
  std::copy(src.h_start_win, src.h_start_win + 256, ix.h_start_win);
  copy(&ix.d_tomb, src.d_tomb, (size_t(src.n_refs) + 31) / 32 + 1);
  if (failed) {
    std::fprintf(stderr, "blurrily_hip: cloning the device image onto device %d failed\n", dst_device);
    device_index_free(&ix);
    (void)hipSetDevice(prev);
    errno = ENOMEM;
    return -1;
  }
  (void)hipSetDevice(prev);
  device_index_free(out);
  *out = std::move(ix);
  return 0;
}

int64_t device_index_rank_of(const DeviceIndex& ix, uint32_t ref) {
  const auto it = std::lower_bound(ix.h_sorted_ref.begin(), ix.h_sorted_ref.end(), ref);
  if (it == ix.h_sorted_ref.end() || *it != ref) return -1;
  return ix.h_rank_of_pos[size_t(it - ix.h_sorted_ref.begin())];
}

}  // namespace blurrily
