// device_index.h -- the read-only, HBM-resident form of the trigram index.
//
// Layout (see DESIGN.md "Data layout in HBM"):
//   * references are compacted to ranks 0..N-1 in ascending (weight, reference)
//     order.  A reference has one weight in every bucket (storage.c:418 writes
//     the same {reference, weight} into each of them), so the reference's
//     result order -- matches descending, weight ascending (storage.c:129-138),
//     reference ascending among full ties (spec/integration_spec.rb:37-42) --
//     is exactly "matches descending, rank ascending": the kernel never needs
//     a weight while it counts and selects.  ref_of_rank / weight_of_rank are
//     the side tables read when result rows are written.
//   * the rank space is cut into windows of kWindowRanks = 65 520 ranks; the
//     postings are stored window-major: for window w, for trigram code t, the
//     16-bit in-window ranks of the references whose string contains t, sorted,
//     each 512-entry unit (one wave-load) stored transposed so that one LDS
//     atomic instruction of the kernel covers consecutive sorted ranks.
//     slice_se[w * kNumCodes + t] = {where the slice's postings start in `ent`, where they end}
//     (a CSR over (window, code)); one extra element closes the last slice.
//     Every slice starts on a 16-byte boundary and is padded to a multiple of
//     eight entries with the sentinel 0xFFFF (the one in-window rank no
//     reference uses), so the kernel counts whole 16-byte groups without any
//     range check; counter slot 0xFFFF is a scratch slot the scan ignores.
//   * a DENSE slice -- at least dense_min8 entries, padding included -- additionally exists as a bitmap over the
//     window's ranks, stored INLINE in `ent` in front of its postings: the kBitmapSlots entries (8 KiB) before
//     slice_se[].x are the bitmap (bit r set iff in-window rank r holds the code).  A kernel tells a dense slice by
//     its length alone (end - start >= dense_min8), so there is no table of bitmap numbers to load, a kernel that
//     leaves nothing out never sees the bitmaps (one 8-byte load hands it a slice's bounds), and every image carries
//     the bitmaps of its dense slices (round 4; rounds 2-3 kept bm_id[] + bitmaps[] beside `ent`, only on images the
//     window-major sweep could run on).  Both sweeps leave
//     dense slices out of the count where the threshold allows and ask the bitmap about the few ranks that matter
//     (find_kernels.hip: sweep_coop, wsweep_kernel).
//   * code_total[t] = used[t] of the reference's bucket t (storage.c:501), for
//     the matched-entries metric.
//   * win_max_tri[w]: the largest number of postings (= distinct trigrams) any one
//     reference of window w has -- no reference of w can match more trigrams than
//     that, so a window whose bound is below the admission threshold is skipped.
//     start_win[L]: the window holding the first rank with weight >= L; a sweep
//     starts at the window of the needle's own length (weight defaults to strlen,
//     storage.c:409), where its best matches live, so the threshold tightens early.
// An entry costs 2 bytes in HBM instead of the reference's 8-byte
// (reference, weight) pair; the algorithmic byte count of DESIGN.md keeps the
// reference's 8 bytes.
#pragma once
#include <cstdint>

#include <hip/hip_vector_types.h>
#include <vector>

#include "host_index.h"

namespace blurrily {

constexpr uint32_t kWindowBits  = 16;   // (15 -- half the counters, 4 workgroups of 512 per CU -- was measured slower; find_kernels.hip assumes 16)
constexpr uint32_t kWindowSize  = 1u << kWindowBits;    // counter slots per window (LDS)
// Ranks per window: the last 16 counter slots stay free of references -- slot 0xFFFF is the padding
// sentinel's, and with the other 15 unused as well the needle-major scan never has to read (and mask)
// the 16-byte vector that holds it: three VALU instructions per scanned vector less.
constexpr uint32_t kWindowRanks = kWindowSize - 16;
constexpr uint16_t kPadRank     = uint16_t(kWindowSize - 1);
constexpr uint32_t kEntPad     = 64;   // u16 slack after the last entry (16-byte over-reads)
// A (window, code) slice with at least this many postings also exists as a BITMAP of the window
// (kWindowSize bits = 8 KiB; bit r set iff in-window rank r holds the code): the window-major sweep
// leaves such slices out of the count and asks the bitmap about the few ranks that matter instead.
constexpr uint32_t kDenseMin     = 1024;                // default of IndexBuildOptions::dense_min
constexpr uint32_t kBitmapWords  = kWindowSize / 32;    // u32 words per bitmap
constexpr uint32_t kBitmapSlots  = kWindowSize / 16;    // the same in 16-bit entries of `ent` (8 KiB: slices stay 16-byte aligned)
constexpr uint32_t kNoBitmap     = 0xFFFFFFFFu;

struct DeviceIndex {
  int       device        = -1;
  uint32_t  n_refs        = 0;
  uint32_t  n_windows     = 0;
  uint32_t  nib_windows   = 0;             // windows [0, nib_windows) (an even count) hold no reference with more than
                                         // 15 distinct trigrams: ANY needle can count them in 4 bits
  uint64_t  n_entries     = 0;        // real postings (padding excluded)
  uint64_t  n_slots       = 0;        // entries of `ent` including slice padding
  uint64_t  device_bytes  = 0;
  uint64_t  built_from    = 0;        // HostIndex::generation() this was built from
  uint32_t* d_ref_of_rank    = nullptr;   // [n_refs]
  uint32_t* d_weight_of_rank = nullptr;   // [n_refs]
  uint2*    d_slice_se       = nullptr;   // [n_windows * kNumCodes] {start, end} of every slice's postings in d_ent
  uint16_t* d_ent            = nullptr;   // [n_slots + kEntPad]
  uint32_t* d_code_total     = nullptr;   // [kNumCodes]
  uint32_t* d_win_max_tri    = nullptr;   // [n_windows] most postings any one reference of the window has
  uint32_t* d_start_win      = nullptr;   // [256] window holding the first rank whose weight is >= the index
  uint32_t  h_start_win[256] = {};        //       ... the host's copy (needles tokenised on the host: c_abi.hip, find_few)
  uint32_t* d_tomb           = nullptr;   // [(n_refs+31)/32] bit r: rank r was deleted after the build
  uint32_t  n_bitmaps        = 0;         // dense slices (each starts with its bitmap, inline in d_ent)
  uint32_t  dense_min8       = 0;         // a slice spanning at least this many entries is dense (a multiple of 8)
  // postings a needle's trigram finds in one window, on average, when needle trigrams are distributed
  // like the haystack's postings: (sum of used[t]^2 / sum of used[t]) / n_windows.  The window-major
  // sweep pays a fixed price per (needle, window) and saves in proportion to the postings it leaves
  // out, so it is taken only where slices are big (c_abi.hip).
  double    mean_hit_slice   = 0.0;
  // share of those postings that sit in slices of at least dense_min postings -- the ones the window-major sweep
  // can leave out of the count: sum of len^2 over dense (window, code) slices / over all slices
  double    dense_share      = 0.0;
  // the part of mean_hit_slice that comes from the haystack's four biggest buckets x postings per reference:
  // the postings a needle as long as the haystack's strings can expect to LEAVE OUT per window -- what the
  // window-major sweep's fixed price per (needle, window) is paid from (measured table in DESIGN.md section 5)
  double    ws_gain          = 0.0;
  // host copies for mapping a reference to its rank (deletes after the build)
  std::vector<uint32_t> h_sorted_ref;     // references ascending
  std::vector<uint32_t> h_rank_of_pos;    // rank of h_sorted_ref[i]
};

// How an image is built and which of them the window-major sweep may be taken on (blurrily_storage_set_option;
// c_abi.hip holds the per-map copy).  Every image carries the bitmaps of its dense slices (Geonames scale: 17 k of
// them, 143 MB, inline in `ent`).
struct IndexBuildOptions {
  bool     ws_enabled     = true;
  uint32_t ws_min_windows = 8;      // fewer windows: the needle-major sweep is taken whatever the batch
  uint32_t ws_min_slice   = 1550;   // least DeviceIndex::mean_hit_slice of an image the sweep may be taken on at all:
                                    // below it the sweep lost on every haystack measured; above
                                    // it c_abi.hip MEASURES the choice per class of batch on first use
  uint32_t dense_min      = kDenseMin;
  bool ws_can_run(uint32_t n_windows, double mean_hit_slice) const {
    return ws_enabled && n_windows >= ws_min_windows && mean_hit_slice >= double(ws_min_slice);
  }
};

// Build the device image of `host` on the current HIP device.  Returns 0, or
// -1 with errno: ENODEV (no device), ENOMEM, EPROTO (postings violate the
// invariants put() guarantees: duplicate reference inside a bucket, a
// reference with two different weights, a reference missing from the leading
// buckets).
int  device_index_build(const HostIndex& host, DeviceIndex* out, const IndexBuildOptions& opt = IndexBuildOptions());
void device_index_free(DeviceIndex* ix);
// A copy of `src` on HIP device `dst_device` (which may be src's own): device to device, nothing is rebuilt on the
// host.  The copy holds no host-side reference table (deletes are mapped to ranks on the original).  0, or -1 + errno.
int  device_index_clone(const DeviceIndex& src, int dst_device, DeviceIndex* out);
// Rank of `ref` in the device image, or -1 if the image does not hold it.
int64_t device_index_rank_of(const DeviceIndex& ix, uint32_t ref);

}  // namespace blurrily
