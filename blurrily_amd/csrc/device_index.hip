// device_index.hip -- build and upload the HBM-resident index (device_index.h).
#include "device_index.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstring>
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

}  // namespace

void device_index_free(DeviceIndex* ix) {
  if (!ix) return;
  if (ix->d_ref_of_rank)    (void)hipFree(ix->d_ref_of_rank);
  if (ix->d_weight_of_rank) (void)hipFree(ix->d_weight_of_rank);
  if (ix->d_slice_off)      (void)hipFree(ix->d_slice_off);
  if (ix->d_ent)            (void)hipFree(ix->d_ent);
  if (ix->d_code_total)     (void)hipFree(ix->d_code_total);
  *ix = DeviceIndex();
}

int device_index_build(const HostIndex& host, DeviceIndex* out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    std::fprintf(stderr, "blurrily_hip: no usable HIP device (find has no CPU fallback)\n");
    errno = ENODEV;
    return -1;
  }
  int dev = 0;
  BLURRILY_HIP_TRY(hipGetDevice(&dev));

  // ---- 1. reference table: every live reference once, ascending ------------
  // Each string owns exactly one leading trigram "**c" (code 784*sym(c)), so
  // the 27 leading buckets list every reference exactly once.
  std::vector<Entry> refs;
  uint64_t lead = 0, nnz = 0;
  for (uint32_t s = 0; s + 1 < uint32_t(kBase); ++s) lead += host.bucket(s * kBase * kBase).used;
  for (uint32_t t = 0; t < kNumCodes; ++t) nnz += host.bucket(t).used;
  if (lead == host.total_refs()) {
    refs.reserve(lead);
    for (uint32_t s = 0; s + 1 < uint32_t(kBase); ++s) {
      const Bucket& bk = host.bucket(s * kBase * kBase);
      refs.insert(refs.end(), bk.e, bk.e + bk.used);
    }
    std::sort(refs.begin(), refs.end(), [](const Entry& l, const Entry& r) { return l.ref < r.ref; });
  } else {
    refs.reserve(nnz);
    for (uint32_t t = 0; t < kNumCodes; ++t) {
      const Bucket& bk = host.bucket(t);
      refs.insert(refs.end(), bk.e, bk.e + bk.used);
    }
    std::sort(refs.begin(), refs.end(), [](const Entry& l, const Entry& r) {
      return l.ref != r.ref ? l.ref < r.ref : l.weight < r.weight;
    });
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
    for (uint32_t i = 0; i < n_refs; ++i) { order[i] = i; sorted_ref[i] = refs[i].ref; }
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t l, uint32_t r) { return refs[l].weight < refs[r].weight; });  // refs[] is ref-ascending
    for (uint32_t rk = 0; rk < n_refs; ++rk) {
      const uint32_t pos = order[rk];
      rank_of_pos[pos] = rk;
      ref_of_rank[rk] = refs[pos].ref;
      weight_of_rank[rk] = refs[pos].weight;
    }
  }
  std::vector<Entry>().swap(refs);

  // ---- 2. rank every posting, count per (window, code) ----------------------
  const uint64_t n_slices = uint64_t(n_win) * kNumCodes;
  if (n_slices + 1 > 0x7FFFFFFFull) { errno = ENOMEM; return -1; }
  std::vector<uint32_t> slice_off(n_slices + 1, 0);
  std::vector<uint32_t> rank(nnz);
  std::vector<uint32_t> code_total(kNumCodes);
  std::vector<Entry> scratch;
  {
    uint64_t idx = 0;
    for (uint32_t t = 0; t < kNumCodes; ++t) {
      const Bucket& bk = host.bucket(t);
      code_total[t] = bk.used;
      if (!bk.used) continue;
      const Entry* e = sorted_view(bk, scratch);
      uint64_t pos = 0;                       // references ascend inside a bucket: gallop from the last hit
      for (uint32_t j = 0; j < bk.used; ++j) {
        const uint32_t ref = e[j].ref;
        uint64_t lo = pos, step = 1;
        while (lo + step < n_refs && sorted_ref[lo + step] < ref) { lo += step; step <<= 1; }
        uint64_t hi = std::min<uint64_t>(lo + step, n_refs ? n_refs - 1 : 0);
        while (lo < hi) {                     // first position in [lo, hi] with sorted_ref >= ref
          const uint64_t mid = (lo + hi) >> 1;
          if (sorted_ref[mid] < ref) lo = mid + 1; else hi = mid;
        }
        if (lo >= n_refs || sorted_ref[lo] != ref || (j > 0 && lo < pos)) { errno = EPROTO; return -1; }
        const uint32_t rk = rank_of_pos[lo];
        if (weight_of_rank[rk] != e[j].weight) { errno = EPROTO; return -1; }   // one weight per reference
        rank[idx++] = rk;
        slice_off[uint64_t(rk / kWindowRanks) * kNumCodes + t + 1] += 1;
        pos = lo + 1;                         // strictly ascending: a duplicate ref fails the lookup above
      }
    }
  }
  // pad every slice to a multiple of eight entries (16 bytes), then prefix-sum
  uint64_t n_slots = 0;
  for (uint64_t i = 0; i < n_slices; ++i) {
    const uint32_t len = (slice_off[i + 1] + 7u) & ~7u;
    slice_off[i + 1] = 0;                     // becomes the running end below
    n_slots += len;
    if (n_slots > 0xFFFF0000ull) { errno = EPROTO; return -1; }
    slice_off[i + 1] = uint32_t(n_slots);
  }

  // ---- 3. scatter into window-major slices ----------------------------------
  std::vector<uint16_t> ent(n_slots + kEntPad, kPadRank);
  {
    std::vector<uint32_t> cursor(slice_off.begin(), slice_off.end() - 1);
    uint64_t idx = 0;
    for (uint32_t t = 0; t < kNumCodes; ++t) {
      const uint32_t used = host.bucket(t).used;
      for (uint32_t j = 0; j < used; ++j) {
        const uint32_t r = rank[idx++];
        ent[cursor[uint64_t(r / kWindowRanks) * kNumCodes + t]++] = uint16_t(r % kWindowRanks);
      }
    }
  }
  std::vector<uint32_t>().swap(rank);

  // ---- 4. upload -------------------------------------------------------------
  DeviceIndex ix;
  ix.device = dev; ix.n_refs = n_refs; ix.n_windows = n_win; ix.n_entries = nnz; ix.n_slots = n_slots;
  ix.built_from = host.generation();
  auto up = [&](auto** dptr, const auto& v, size_t min_elems) -> int {
    using T = typename std::remove_reference<decltype(v)>::type::value_type;
    const size_t bytes = std::max(v.size(), min_elems) * sizeof(T);
    BLURRILY_HIP_TRY(hipMalloc(reinterpret_cast<void**>(dptr), bytes));
    if (!v.empty()) BLURRILY_HIP_TRY(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    ix.device_bytes += bytes;
    return 0;
  };
  if (up(&ix.d_ref_of_rank, ref_of_rank, 1) || up(&ix.d_weight_of_rank, weight_of_rank, 1) ||
      up(&ix.d_slice_off, slice_off, 1) || up(&ix.d_ent, ent, 1) || up(&ix.d_code_total, code_total, 1)) {
    const int e = errno;
    device_index_free(&ix);
    errno = e;
    return -1;
  }
  device_index_free(out);
  *out = ix;
  return 0;
}

}  // namespace blurrily
