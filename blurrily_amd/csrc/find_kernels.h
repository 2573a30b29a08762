// find_kernels.h -- launch interface of the find hot path (find_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/blurrily_storage.h"
#include "device_index.h"

namespace blurrily {

struct TokeniseArgs {
  const char*     packed;      // needle bytes (device)
  const uint64_t* offsets;     // [n+1] byte offsets (device)
  uint32_t        n;
  const uint32_t* code_total;  // [kNumCodes] bucket sizes
  uint16_t*       qcodes;      // [offsets[n] + n] distinct codes, needle q at offsets[q]+q
  uint32_t*       q_ntri;      // [n] distinct trigram count per needle
  uint32_t*       q_nb;        // [n] nb_entries per needle (storage.c:498-502)
  uint32_t*       big_list;    // [n] needles with > 127 distinct trigrams
  uint32_t*       big_count;   // [1]
  uint32_t*       mid_list;    // [n] needles with 65..127 distinct trigrams
  uint32_t*       mid_count;   // [1]
  const uint32_t* start_win;   // [256] index table: first window of a weight
  uint32_t*       q_start;     // [n] window a needle's sweep starts at
  uint32_t        max_len;     // longest needle in bytes if the host knows it, else 0
};

struct FindArgs {
  // index
  const uint2*    slice_se;    // [n_windows * kNumCodes] {start, end} of a slice's postings in `ent`
  const uint16_t* ent;
  const uint32_t* ref_of_rank;
  const uint32_t* weight_of_rank;
  uint32_t        n_refs;
  uint32_t        n_windows;
  // needles
  const uint64_t* offsets;
  const uint16_t* qcodes;
  const uint32_t* q_ntri;
  const uint32_t* q_nb;
  const uint32_t* q_start;     // [n] window the sweep starts at (the needle's own length class)
  const uint32_t* win_max_tri; // [n_windows] match-count bound per window
  uint32_t        nib_windows; // windows [0, nib_windows): no reference with more than 15 trigrams (4-bit counters suffice)
  const uint32_t* tomb;        // bit r set: rank r was deleted after the image was built (nullptr: none)
  // dense slices start with their bitmap (device_index.h): a slice spanning at least this many entries is dense
  uint32_t        dense_min8;
  // needle-major sweep (sweep_coop): with a threshold, the largest slices of at least nm_dense postings are left out
  // of a step's count -- at most need - nm_cmin of them per window -- and settled per candidate through their bitmaps
  // (nm_cmin 0: nothing is ever left out)
  uint32_t        nm_dense;
  uint32_t        nm_cmin;
  // window-major sweep (wsweep_kernel)
  uint32_t        own_only;    // find_kernel: sweep only the window pair of the needle's own length class and leave
                               // the best keys (not rows) in `results` as the needle's state for wsweep_kernel
  uint32_t        cmin;        // wsweep: a skipped slice must leave at least this many matches to count (>= 1)
  const uint32_t* work_list;   // nullptr: slots are needle ids, long needles skipped
  const uint32_t* n_work_dev;  // when set, the number of slots is read from the device
  uint32_t        n_work;
  uint32_t*       queue;       // [1] zeroed before launch
  // results
  trigram_match_t* results;    // [n * limit]
  uint32_t*       counts;      // [n]
  uint32_t        limit;       // row stride
  uint32_t        keep;        // rows wanted from this pass
  uint32_t        pass_base;   // rows already delivered by earlier passes
  uint32_t        pool_cap;    // power of two, >= 4*keep
  unsigned long long* floor;   // [n] last key delivered by the previous pass (multi-pass only)
  // latency mode (small batches): every needle's windows are cut into `ranges` tasks
  uint32_t        short_only;  // byte-counter launches: own only needles with <= 64 distinct trigrams
  uint32_t        ranges;      // 0/1: off
  unsigned long long* part_keys;   // [n_items * ranges * keep] best keys of every task
  uint32_t*       part_count;  // [n_items * ranges]
  // small-haystack sweep (find_small_kernel): needles it does not own (16..64 distinct trigrams) are listed here for the
  // byte-counter launch that follows
  uint32_t*       over_list;   // [n]
  uint32_t*       over_count;  // [1] zeroed before launch
  unsigned long long* phase_clocks;  // profiling builds only (make profile), else nullptr
  // optional request counters of a launch (nullptr: off; see kStat* below) -- what the kernels asked of
  // the memory system and of the LDS, counted exactly from wave-uniform values
  unsigned long long* stats;
  // counted build only: which paths of the kernels a needle's find went through, one word per needle, OR of
  // kPath* below (blurrily_storage_find_path_flags: the parity tests sample every class)
  uint32_t*       path_flags;
};

// bits of FindArgs::path_flags
enum : uint32_t {
  kPathNibble       = 1u << 0,    // swept windows with 4-bit counters, two per step (sweep_coop<Nib>)
  kPathByte         = 1u << 1,    // swept windows with byte counters (sweep_coop<uint8_t>)
  kPathColdStart    = 1u << 2,    // no threshold yet: the first scan's bound found by bisection
  kPathResweep      = 1u << 3,    // a step swept again after the candidate pool overflowed
  kPathCompaction   = 1u << 4,    // the pool compacted in mid-sweep (threshold tightened)
  kPathSkipped      = 1u << 5,    // steps stepped over: no reference of the window can reach the threshold
  kPathRingOverflow = 1u << 6,    // more units than the descriptor ring holds: every wave walked the table
  kPathPipelined    = 1u << 7,    // sweep_pipelined (65..128 distinct trigrams, wide counters)
  kPathWide         = 1u << 8,    // 16-bit counters (more than 127 distinct trigrams)
  kPathChunked      = 1u << 9,    // slice table staged through LDS in chunks (more than 128 distinct trigrams)
  kPathRanged       = 1u << 10,   // latency mode: the needle's windows cut into ranges, merged afterwards
  kPathMultiPass    = 1u << 11,   // a later pass of a limit larger than the pool (floor key)
  kPathTombstone    = 1u << 12,   // a candidate rejected because its reference was deleted since the build
  kPathOwnOnly      = 1u << 13,   // phase 1 of the window-major sweep (own length class only)
  kPathWsTask       = 1u << 14,   // window-major sweep: at least one (needle, window) task ran
  kPathWsLeftOut    = 1u << 15,   //   ... with dense slices left out of the count and probed through bitmaps
  kPathWsRobust     = 1u << 16,   //   ... the robust scan (no threshold yet, or after a candidate-list overflow)
  kPathWsCandOv     = 1u << 17,   //   ... the candidate list overflowed
  kPathWsPoolOv     = 1u << 18,   //   ... the candidate pool overflowed
  kPathWsWide       = 1u << 19,   //   ... byte counters over the two halves of the window
  kPathWsTableWalk  = 1u << 20,   //   ... more than 64 units: the waves walked the published slice table
  kPathNmLeftOut    = 1u << 21,   // needle-major sweep: a candidate settled through the bitmaps of left-out slices
  kPathSmall        = 1u << 22,   // small-haystack sweep (find_small_kernel: four waves per needle, one window per step)
};

// slots of FindArgs::stats
enum : uint32_t {
  kStatPostingEntries = 0,   // 16-bit postings loaded (x2 = bytes); each is also one LDS-atomic lane
  kStatSteps          = 1,   // sweep steps (count -> barrier -> scan -> barrier)
  kStatTableWords     = 2,   // slice-table words loaded (x4 = bytes)
  kStatTasks          = 3,   // needles (or needle ranges) swept
  kStatCompactions    = 4,   // candidate-pool compactions
  kStatResweeps       = 5,   // windows swept again after a pool overflow
  kStatUnits          = 6,   // wave-loads of postings (up to 512 each) issued by the window-major sweep
  kStatProbes         = 7,   // bitmap words read for candidates of the window-major sweep (x4 = bytes)
  kStatSlots          = 8,   // what blurrily_storage_find_stats reports
  kStatWsClocks       = 8,   // + 0..5: shader clocks of wave 0 per phase of wsweep_kernel, summed over workgroups
                             //   (filter, task set-up, count, scan, probe, select + write-back); debugging aid
  kStatAllSlots       = 16,
};

// Every find-kernel launch notes its name here (c_abi.hip keeps, per map, the kernels of the last batch's short-needle
// launches in launch order: blurrily_storage_last_kernels -- what a bench line names instead of guessing).
void note_launch(const char* kernel_name);

uint32_t find_pool_cap(uint32_t keep);
bool find_can_leave(uint32_t keep);   // limits whose pool leaves room for the needle-major sweep's settled candidates (up to ~150)
int find_threads();
uint32_t find_wgs_per_cu();   // resident byte-counter workgroups per CU (LDS and wave limits)   // workgroup size of the find kernel (BLURRILY_FIND_THREADS, default 1024)
int launch_tokenise(const TokeniseArgs& t, hipStream_t stream);
// Blurrily::Map#normalize_string for ASCII needles, in place or not (find_kernels.hip: normalise_kernel)
int launch_normalise(const char* in, const uint64_t* offsets, uint32_t n, char* out, uint32_t* non_ascii,
                     hipStream_t stream);
int launch_find(const FindArgs& a, bool long_needles, uint32_t grid, hipStream_t stream);
int launch_merge_parts(const FindArgs& a, uint32_t n_items, hipStream_t stream);
// ... the merged rows of needle q into host-coherent memory instead: rows[q][kOneMaxKeep], words[q] = {count, seq} (seq last,
// behind a system-scope fence: the host polls it), and *a.queue back to zero.  a.keep <= kOneMaxKeep.
int launch_merge_parts_pinned(const FindArgs& a, uint32_t n_items, trigram_match_t* rows, uint32_t* words, uint32_t seq,
                              hipStream_t stream);
// Small-haystack sweep over needles [0, a.n_work): four waves and one window's 4-bit counters per needle, four workgroups
// per CU; needles with 16..64 distinct trigrams are appended to a.over_list.  keep <= kSmallMaxKeep, single pass.
int launch_find_small(const FindArgs& a, uint32_t n_cus, hipStream_t stream);
constexpr uint32_t kSmallMaxKeep = 64, kSmallMaxWindows = 8;
// ONE needle, the caller waiting (blurrily_storage_find) -- or a handful, or a server's coalesced FINDs (up to
// kMidMaxNeedles): one launch, no copies.  Row i of the grid is needle i's: workgroup g of it sweeps windows
// [g * per, (g + 1) * per) and leaves its best keys; one workgroup of the row merges the row's lists and writes rows, count
// and `seq` into host-coherent memory (out_rows[i][kOneMaxKeep], out_count[i][2]).  Up to kOneMaxNeedles needles travel
// as kernel arguments (codes[i * 64 ..], T[i] of them; codes_far / T_far / tickets nullptr) and the row's LAST workgroup
// merges, waiting for the others' flags (any value but `seq`: a launch's workgroups set theirs to it); more needles are
// read from host-coherent memory (codes_far [n][64], T_far [n]) and the workgroup that finishes last merges (tickets [n],
// zero between launches).  T <= 64 distinct trigrams, a.keep <= kOneMaxKeep, grid <= kOneMaxGrid; a.tomb: the select
// drops deleted references; part_keys [n_needles][grid * keep], flags [n_needles][grid].  Timed build only.
// (kOneMaxGrid: what the merge's scratch -- a window's counters -- holds at kOneMaxKeep; 288 lists let an image of up to 576
// windows give every workgroup ONE window pair: four times configs[2]'s haystack, 515 windows, 258 workgroups)
constexpr uint32_t kOneMaxKeep = 120, kOneMaxGrid = 288, kOneMaxNeedles = 16, kMidMaxNeedles = 128;
int launch_find_one(const FindArgs& a, const uint16_t* codes, const uint32_t* T, uint32_t n_needles, uint32_t per, uint32_t grid,
                    unsigned long long* part_keys, uint32_t* flags, trigram_match_t* out_rows,
                    uint32_t* out_count, uint32_t seq, hipStream_t stream, uint32_t n_cus, const uint16_t* codes_far = nullptr,
                    const uint32_t* T_far = nullptr, uint32_t* tickets = nullptr);
// Window-major sweep of window `w` over needles [0, n) (those with <= 64 distinct trigrams); a.queue must
// be a zeroed word of its own.  own_pass: only the needles whose own length class lives in this window
// pair -- the launches that seed the states (counts[] zeroed before the first of them); else the others.
int launch_wsweep(const FindArgs& a, uint32_t w, uint32_t n, uint32_t n_cus, bool own_pass, hipStream_t stream);
// Turn the needles' states (keys) into result rows, in place.
int launch_finalize_rows(const FindArgs& a, uint32_t n, hipStream_t stream);
constexpr uint32_t kWsMaxKeep = 128;   // largest limit the window-major sweep serves

// the same launches of the build that keeps FindArgs::stats (find_kernels_counted.hip)
namespace counted {
int launch_find(const FindArgs& a, bool long_needles, uint32_t grid, hipStream_t stream);
int launch_find_small(const FindArgs& a, uint32_t n_cus, hipStream_t stream);
int launch_wsweep(const FindArgs& a, uint32_t w, uint32_t n, uint32_t n_cus, bool own_pass, hipStream_t stream);
int launch_finalize_rows(const FindArgs& a, uint32_t n, hipStream_t stream);
}
// Merge, per needle, two result lists that are each in result order (base image and delta image
// hold disjoint references) into the first `limit` rows of `out`.
int launch_merge_rows(const trigram_match_t* a_rows, const uint32_t* a_counts, const trigram_match_t* b_rows,
                      const uint32_t* b_counts, uint32_t n, uint32_t limit, trigram_match_t* out,
                      uint32_t* out_counts, hipStream_t stream);

}  // namespace blurrily
