// c_abi.hip -- extern "C" entry points of libblurrily_hip.so
// (include/blurrily_storage.h).  Part 1 mirrors ext/blurrily/storage.h:36-117.
#include "../../include/blurrily_storage.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "device_index.h"
#include "find_kernels.h"
#include "host_index.h"

using namespace blurrily;

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

// A map's device image lives on the HIP device that was current when it was built.  Every entry
// point that touches the image runs inside a DeviceScope: the current device is switched to the
// map's and restored on the way out, so a caller (torch, another map) may leave any device current.
struct DeviceScope {
  int  prev = -1;
  bool changed = false;
  explicit DeviceScope(int want) {
    if (want >= 0 && hipGetDevice(&prev) == hipSuccess && prev != want) changed = hipSetDevice(want) == hipSuccess;
  }
  ~DeviceScope() { if (changed) (void)hipSetDevice(prev); }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};

// deletes since the last find: set their bits in the tombstone bitmap, in stream order with the find
__global__ void apply_tombstones_kernel(uint32_t* __restrict__ tomb, const uint32_t* __restrict__ ranks, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicOr(&tomb[ranks[i] >> 5], 1u << (ranks[i] & 31));
}

// Device scratch that lives as long as the map and only ever grows.
struct DeviceBuffer {
  void*  p = nullptr;
  size_t bytes = 0;
  int reserve(size_t want, hipStream_t stream) {
    if (want <= bytes) return 0;
    if (p) { (void)hipStreamSynchronize(stream); (void)hipFree(p); p = nullptr; bytes = 0; }
    const size_t grow = std::max(want, bytes + bytes / 2);
    BLURRILY_HIP_TRY(hipMalloc(&p, grow));
    bytes = grow;
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

}  // namespace

// Mutations since the base device image was built (DESIGN.md "Mutation and device sync").
struct PendingPut {
  std::string needle;
  uint32_t    weight;
};

struct trigram_map_t;
namespace {
// where note_launch() writes: the `last_kernels` of the map whose batch is being enqueued on this thread (run_find_on, find_few)
thread_local std::string* t_launch_names = nullptr;
struct NameScope {                                     // the launches inside it note their kernels' names in *s
  std::string* prev;
  explicit NameScope(std::string* s) : prev(t_launch_names) { t_launch_names = s; }
  ~NameScope() { t_launch_names = prev; }
};
}
namespace blurrily {
void note_launch(const char* kernel_name) {
  std::string* s = t_launch_names;
  if (!s) return;
  // distinct names, launch order (a window-major batch launches wsweep_kernel once per window)
  const std::string name(kernel_name);
  size_t at = 0;
  while (at <= s->size()) {
    const size_t end = s->find('+', at);
    const std::string have = s->substr(at, end == std::string::npos ? std::string::npos : end - at);
    if (have == name) return;
    if (end == std::string::npos) break;
    at = end + 1;
  }
  if (!s->empty()) *s += '+';
  *s += name;
}
}
// "devices" > 1: one more copy of the map's device side, on another visible device (or, with more replicas than
// devices, on one that already has one): clones of the primary's images, a map object of its own for the scratch
// buffers, events and measured choices its finds need, a stream, and staging for its shard of a batch.
struct Replica {
  int            device = -1;
  bool           same_device = false;   // it sits on the primary's own device (more replicas than devices)
  bool           peer_access = false;   // its device and the primary's reach each other's memory directly (both ways, enabled)
  trigram_map_t* side = nullptr;        // dev / delta / d_code_total_now are the clones; host == nullptr; mirror_of = the primary
  uint64_t       base_builds = 0, delta_image_version = 0, log_version = 0;   // the primary's, as of the clones
  hipStream_t    stream = nullptr;
  hipEvent_t     ev_done = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
  DeviceBuffer   d_in, d_out;           // [offsets | needles] of the batch, [rows | counts | nb_entries] of its shard
};

struct trigram_map_t {
  HostIndex*  host = nullptr;
  const trigram_map_t* mirror_of = nullptr;   // a replica's side map: the mutation log that counts is this map's
  std::vector<Replica> replicas;        // "devices" - 1 of them
  uint32_t    n_devices = 1;            // option "devices": shards of a large batch (the primary's included)
  hipEvent_t  ev_ready = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;   // multi-device batches
  DeviceIndex dev;                      // base image
  // log of puts/deletes the base image does not contain yet
  std::unordered_map<uint32_t, PendingPut> pending;   // by reference
  size_t      n_tomb = 0;               // base references deleted since the build
  std::vector<uint32_t> tomb_queue;     // their ranks, not yet on the device (applied by the next find, on its stream)
  DeviceBuffer ws_tomb;
  uint64_t    log_version = 0;          // bumped by every logged mutation
  uint64_t    delta_version = 0;        // log_version the delta image / code totals were built from
  uint64_t    delta_puts_version = 0;   // bumped when the set of pending puts changes (the delta image's content)
  uint64_t    delta_image_version = 0;  // delta_puts_version the delta image was built from
  bool        log_overflow = false;     // the log outgrew its budget: the next find rebuilds the base
  uint64_t    base_builds = 0;
  HostIndex*  delta_host = nullptr;
  DeviceIndex delta;                    // image of `pending` only
  uint32_t*   d_code_total_now = nullptr;   // [kNumCodes] bucket sizes of the whole map (base run's nb_entries)
  DeviceBuffer ws_base_rows, ws_base_counts, ws_delta_rows, ws_delta_counts;
  // tunables of the window-major sweep (blurrily_storage_set_option; defaults from the measured gate)
  IndexBuildOptions build_opt;          // ws_enabled, ws_min_windows, ws_min_slice, dense_min
  uint32_t    ws_cmin = 3;              // a left-out slice must leave at least this many counted matches
  uint32_t    nm_cmin = 3;              // the same for the needle-major sweep (0: it leaves nothing out)
  uint32_t    nm_dense = 3072;          // ... which leaves out slices of at least this many postings only (4 096 through round 5;
                                        // round 6, same rows: configs[2] 120.8 -> 119.1 ms per 300 k needles, four times the haystack 135.1 -> 129.9)
  bool        small_sweep = true;       // images of at most kSmallMaxWindows windows: find_small_kernel serves large batches at limits up to 64
  uint32_t    small_min_needles = 4096; // ... from this many needles on (below: two chains per CU are not the limit)
  uint32_t    nm_min_windows = 256;     // ... and, where the choice is not measured, on images of at least this many windows
  uint32_t    ws_min_needles = 16384;   // smaller batches: needle-major
  bool        ws_autotune = true;       // measure the choice per class of batch on first use (run_find_on)
  uint32_t    ws_static_slice = 2200;   // the static rule's mean_hit_slice (autotune off): break-even of the skewed family
  int         ws_choice[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // per class: 0 not measured yet, 1 needle-major, 2 window-major,
                                                   // 3 needle-major with slices left out
  float       ws_tuned_ms[8][3] = {};   // what the measurement saw (needle-major, window-major, slices left out)
  int         last_tuned = -1;          // the class measured most recently ("tuned_*_us" report its figures)
  int         last_sweep = 0;           // which sweep the last large batch of short needles took (1 / 2 / 3; 0: none yet)
  uint32_t latency_tasks = 0;           // option "latency_tasks": tasks latency mode aims at per resident workgroup (0: latency_ranges' rule)
  std::string last_kernels;             // the find kernels the last batch on the base image launched, '+'-joined (blurrily_storage_last_kernels)
  size_t      class_hint = 0;           // a chunked host batch: the WHOLE batch's size decides the class, not the chunk's
  hipEvent_t  tune_ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // a measured choice is WATCHED: the chosen sweep's later batches of the class are bracketed by two events (read at the
  // class's next batch, never waited for); one that ran over 10 % slower per needle than what the measurement saw has
  // the class measured again -- at most once in sixteen batches
  hipEvent_t  watch_ev[8][2] = {};
  bool        watch_pending[8] = {false, false, false, false, false, false, false, false};
  size_t      watch_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  size_t      tuned_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // the batch size a class was measured at (the watch compares like with like)
  float       tuned_us_per_needle[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // of the sweep that was chosen
  uint32_t    retune_holdoff[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t    watch_strikes[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // consecutive batches of the class seen slow (one is noise: another tenant, a clock step)
  uint64_t    retunes = 0;              // classes measured again because a batch ran slow (option "retunes", read-only)
  int         tune_inject = 0;          // (tests) the next measurement sees this sweep at HALF its time: a bad sample to recover from
  int         n_cus = 0;
  bool        timing = false;
  bool        collect_stats = false;    // request counters of the find kernels (FindArgs::stats)
  unsigned long long* d_stats = nullptr;   // [kStatSlots], zeroed by every run_find while collecting
  unsigned long long* d_phase = nullptr;   // [kPhaseWorkgroups][16] phase clocks of the counted build's last launch
  DeviceBuffer ws_flags;                   // [n] path flags of the last find while collecting (FindArgs::path_flags)
  size_t      n_flags = 0;
  double      last_find_ms = 0.0, last_tok_ms = 0.0;
  hipEvent_t  ev[4] = {nullptr, nullptr, nullptr, nullptr};
  DeviceBuffer ws_codes, ws_small, ws_parts, ws_io_in, ws_io_out;
  unsigned char* h_stage = nullptr;     // pinned host staging: [kStageBytes in | kStageBytes out]
  // the single find's own launch (find_one): a stream, host-coherent pinned memory the kernel writes rows, count and a
  // sequence word into, the per-workgroup lists and the ticket on the device
  struct One {
    hipStream_t    stream = nullptr;
    unsigned char* h_out = nullptr;     // per image: [kMidMaxNeedles][kOneMaxKeep] rows | [..][2] count, sequence word | codes | T (kOneHostBytes)
    unsigned char* d_out = nullptr;     // the same memory as the device addresses it
    DeviceBuffer   d_parts;             // per image: [kOneMaxLists][kOneMaxKeep] keys | [kOneMaxLists] flags | [kMidMaxNeedles] tickets
    uint32_t       seq = 0;
    bool           enabled = true;      // option "one_launch"
    uint32_t       min_per = 0;         // option "one_windows_per_wg": at least this many windows per workgroup (0: as few as the grid allows)
    uint32_t       mid_workgroups = 1024;   // option "mid_workgroups": workgroups a launch of more than kOneMaxNeedles needles aims at
    uint32_t       few_max = 24;        // option "few_max": host-buffer batches of up to this many needles share find_one_kernel's launch
                                        // (up to kMidMaxNeedles; from about thirty needles on latency mode's ranges are faster: DESIGN.md)
    uint32_t       mid_max = kMidMaxNeedles;   // option "mid_max": ... and up to this many take latency mode WITHOUT copies: tokenised on the
                                        // host, read from the pinned page, the merged rows written back into it (find_few)
    uint64_t       taken = 0;           // finds served this way (option "one_taken", read-only)
  } one;
  // large host-buffer batches go in chunks through a three-stream pipeline (find_batch_chunked)
  uint32_t    host_chunk = 131072;      // needles per chunk (option "host_chunk"; 0: never chunk)
  struct Pipe {
    hipStream_t s_in = nullptr, s_run = nullptr, s_out = nullptr;
    hipEvent_t  ev_in[2] = {nullptr, nullptr}, ev_run[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    unsigned char* h_in[2] = {nullptr, nullptr};     // pinned
    unsigned char* h_out[2] = {nullptr, nullptr};
    size_t h_in_bytes = 0, h_out_bytes = 0;
    DeviceBuffer d_in[2], d_out[2];
  } pipe;
};

namespace {

size_t log_budget(const trigram_map m) { return std::max<size_t>(4096, m->dev.n_refs / 64); }

void clear_log(trigram_map m) {
  m->pending.clear();
  m->n_tomb = 0;
  m->tomb_queue.clear();
  m->log_overflow = false;
  ++m->log_version;
  m->delta_version = m->log_version;
  if (m->delta.device >= 0) device_index_free(&m->delta);
  delete m->delta_host;
  m->delta_host = nullptr;
}

// (a replica's side map holds clones of the images; the log they were cloned at is the primary's)
const trigram_map_t* log_of(const trigram_map_t* m) { return m->mirror_of ? m->mirror_of : m; }
bool log_empty(const trigram_map_t* m) { const trigram_map_t* l = log_of(m); return l->pending.empty() && l->n_tomb == 0 && !l->log_overflow; }

// Bring the device side up to date with the host index.  Small logs are served by a delta
// image (built from the pending puts only) plus tombstones on the base image; a log past
// 1/64 of the base (or 4096 mutations) triggers a full rebuild.
int ensure_device(trigram_map m) {
  const bool have_base = m->dev.device >= 0;
  if (!have_base || m->log_overflow || m->pending.size() + m->n_tomb > log_budget(m)) {
    if (device_index_build(*m->host, &m->dev, m->build_opt) < 0) return -1;
    ++m->base_builds;
    std::fill(std::begin(m->ws_choice), std::end(m->ws_choice), 0);     // a new image: measure again
    clear_log(m);
    if (m->n_cus == 0) {
      hipDeviceProp_t prop;
      BLURRILY_HIP_TRY(hipGetDeviceProperties(&prop, m->dev.device));
      m->n_cus = prop.multiProcessorCount;
    }
    return 0;
  }
  if (log_empty(m) || m->delta_version == m->log_version) return 0;
  // The delta host index is kept in step by log_put / log_delete; its device image is rebuilt only
  // when the set of pending puts changed (a delete of a base reference is a tombstone, no rebuild).
  if (m->delta_image_version != m->delta_puts_version) {
    if (m->pending.empty()) {
      if (m->delta.device >= 0) device_index_free(&m->delta);
    } else if (device_index_build(*m->delta_host, &m->delta, m->build_opt) < 0) {
      return -1;
    }
    m->delta_image_version = m->delta_puts_version;
  }
  // whole-map bucket sizes (nb_entries of the base run)
  std::vector<uint32_t> totals(kNumCodes);
  for (uint32_t t = 0; t < kNumCodes; ++t) totals[t] = m->host->bucket(t).used;
  if (!m->d_code_total_now)
    BLURRILY_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->d_code_total_now), kNumCodes * sizeof(uint32_t)));
  BLURRILY_HIP_TRY(hipMemcpy(m->d_code_total_now, totals.data(), kNumCodes * sizeof(uint32_t), hipMemcpyHostToDevice));
  m->delta_version = m->log_version;
  return 0;
}

// Host side of put/delete after the base image exists: log what the image is missing.
void log_put(trigram_map m, const char* needle, size_t len, uint32_t ref, uint32_t weight) {
  if (m->dev.device < 0 || m->log_overflow) return;    // no image yet / rebuild pending: nothing to track
  if (m->pending.size() >= log_budget(m)) {            // bulk import: stop logging, rebuild at the next find
    m->pending.clear();
    m->log_overflow = true;
    return;
  }
  if (!m->delta_host) m->delta_host = new HostIndex();
  if (m->delta_host->put(needle, len, ref, weight) < 0) {   // (out of memory) the delta image would miss it:
    m->pending.clear();                                     // fold everything into a rebuilt base instead
    m->log_overflow = true;
    return;
  }
  m->pending[ref] = PendingPut{std::string(needle, len), weight};
  ++m->delta_puts_version;
  ++m->log_version;
}

int log_delete(trigram_map m, uint32_t ref) {
  if (m->dev.device < 0 || m->log_overflow) return 0;
  ++m->log_version;
  if (m->pending.erase(ref)) {                         // never reached the base image
    if (m->delta_host) m->delta_host->del(ref);
    ++m->delta_puts_version;
    return 0;
  }
  const int64_t rk = device_index_rank_of(m->dev, ref);
  if (rk < 0) return 0;
  // the tombstone bit is set by the next find, on that find's stream (apply_tombstones): no
  // synchronous round trip per delete, and ordered with whatever stream the caller finds on
  m->tomb_queue.push_back(uint32_t(rk));
  ++m->n_tomb;
  return 0;
}

// Upload the queued tombstone ranks and set their bits, ordered before the find on `stream`.
int apply_tombstones(trigram_map m, hipStream_t stream) {
  if (m->tomb_queue.empty()) return 0;
  const size_t n = m->tomb_queue.size();
  if (m->ws_tomb.reserve(n * sizeof(uint32_t), stream) < 0) return -1;
  // Deletes are rare on this path: a synchronous copy (the queue may be cleared when it returns, whatever the
  // runtime does with a pageable source), the kernel on the find's stream, and a wait for it -- so that the bits
  // are set for every stream and for device_info / debug reads, not only for finds on this one.
  BLURRILY_HIP_TRY(hipMemcpy(m->ws_tomb.p, m->tomb_queue.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(apply_tombstones_kernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, stream, m->dev.d_tomb,
                     static_cast<const uint32_t*>(m->ws_tomb.p), uint32_t(n));
  BLURRILY_HIP_TRY(hipGetLastError());
  BLURRILY_HIP_TRY(hipStreamSynchronize(stream));
  m->tomb_queue.clear();
  return 0;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr size_t kPhaseWorkgroups = 8192, kPhaseBytes = kPhaseWorkgroups * 16 * 8;
constexpr size_t kStageBytes = 1 << 20;   // pinned staging per direction for small host-buffer batches

// the timed build of the kernels, or (while request counters are collected) the counted one
// Latency mode: the ranges a needle's windows are cut into when n needles are too few to fill `wgs` resident workgroups
// (1: whole needles).  Tasks aimed at: ONE per workgroup up to sixty needles -- every task starts at once, none
// queues behind another's learning sweep --, two beyond (round 6, tools/experiments/r6_run_mid5.sh, host clock at Geonames
// scale, one / two / three / four tasks per workgroup: 32 needles 131 / 179 / 197 / 228 us, 56: 188 / 226 / 244 / 268,
// 64: 216 / 209 / 239 / 274, 128: 319 / 261 / 285 / 302; through round 5 two, and four from a hundred needles on); from
// seven needles per SIXTEEN workgroups on (225 needles on this chip) whole needles win: up to a needle per workgroup their
// time is the slowest needle's, 368 us at Geonames scale whatever the batch, and ranges take 243 us at 129 needles, 298 at
// 160, 310 at 192, 356 at 224, 385 at 256 (round 6: tools/experiments/r6_lat256.py; through round 5 the crossover sat at one
// needle per four workgroups, through round 4 at one per workgroup: the ranged sweep paid a whole learning sweep per task
// and ended with its dearest tasks).
uint32_t latency_ranges(size_t n, uint32_t limit, uint32_t n_windows, size_t wgs, uint32_t tasks_per_wg) {
  if (limit == 0 || limit > 1024 || n * 16 > wgs * 7 || n_windows <= 2) return 1;
  const size_t target_tasks = (tasks_per_wg ? tasks_per_wg : n <= 60 ? 1 : 2) * wgs;      // (option "latency_tasks": 0 = this rule)
  uint32_t ranges = uint32_t(std::min<size_t>((n_windows + 1) / 2, target_tasks / n));   // ranges are whole window pairs
  return std::max<uint32_t>(1u, std::min<uint32_t>(ranges, std::max<uint32_t>(1u, 4096u / limit)));      // (the merge's pool)
}

int do_launch_find(bool counted_build, const FindArgs& a, bool long_needles, uint32_t grid, hipStream_t stream) {
  return counted_build ? counted::launch_find(a, long_needles, grid, stream) : launch_find(a, long_needles, grid, stream);
}

// Enqueue tokenise + find for n device-resident needles.
int run_find_on(trigram_map m, const DeviceIndex& ix, const uint32_t* d_code_total, const uint32_t* d_tomb,
                const char* d_packed, size_t packed_bytes, const uint64_t* d_offsets, size_t n, uint16_t limit,
                trigram_match d_results, uint32_t* d_counts, uint32_t* d_nb, bool maybe_long, bool maybe_mid,
                hipStream_t stream) {
  if (n == 0) return 0;
  if (n > 0xFFFFFFF0ull) { errno = EINVAL; return -1; }
  const bool is_base = &ix == &m->dev;                 // (the delta image of pending puts is searched the same way)
  if (is_base) m->last_sweep = 0;
  NameScope name_scope(is_base ? &m->last_kernels : nullptr);    // the launches below note their kernels' names in the map
  if (is_base) m->last_kernels.clear();

  // scratch: codes | per-needle arrays | scalars
  const size_t code_slots = packed_bytes + n;
  if (m->ws_codes.reserve(align_up(code_slots * sizeof(uint16_t), 256), stream) < 0) return -1;
  const size_t per_n = align_up(n * sizeof(uint32_t), 256);
  const bool multi_pass = limit > 256;             // long needles keep 256 rows per pass, short ones 1024
  const size_t small_bytes = per_n * 6 + (multi_pass ? align_up(n * 8, 256) : 0) + 256;
  if (m->ws_small.reserve(small_bytes, stream) < 0) return -1;
  unsigned char* sp = static_cast<unsigned char*>(m->ws_small.p);
  uint32_t* scalars  = reinterpret_cast<uint32_t*>(sp);            sp += 256;   // [0]=big_count [1]=mid_count [2]=over_count [3..]=queues
  uint32_t* q_ntri   = reinterpret_cast<uint32_t*>(sp);            sp += per_n;
  uint32_t* q_nb_ws  = reinterpret_cast<uint32_t*>(sp);            sp += per_n;
  uint32_t* big_list = reinterpret_cast<uint32_t*>(sp);            sp += per_n;
  uint32_t* mid_list = reinterpret_cast<uint32_t*>(sp);            sp += per_n;
  uint32_t* q_start  = reinterpret_cast<uint32_t*>(sp);            sp += per_n;
  uint32_t* over_list = reinterpret_cast<uint32_t*>(sp);           sp += per_n;   // (small-haystack sweep: needles of 16..64 trigrams)
  unsigned long long* floor = nullptr;
  if (multi_pass) floor = reinterpret_cast<unsigned long long*>(sp);
  uint32_t* q_nb = d_nb ? d_nb : q_nb_ws;
  BLURRILY_HIP_TRY(hipMemsetAsync(scalars, 0, 256, stream));

  if (m->timing) BLURRILY_HIP_TRY(hipEventRecord(m->ev[0], stream));
  TokeniseArgs t{d_packed, d_offsets, uint32_t(n), d_code_total, static_cast<uint16_t*>(m->ws_codes.p),
                 q_ntri, q_nb, big_list, scalars, mid_list, scalars + 1, ix.d_start_win, q_start,
                 maybe_mid ? 0u : 63u};                  // host-buffer batches know their longest needle
  if (launch_tokenise(t, stream) < 0) return -1;
  if (m->timing) {
    BLURRILY_HIP_TRY(hipEventRecord(m->ev[1], stream));
    BLURRILY_HIP_TRY(hipEventRecord(m->ev[2], stream));
  }

  FindArgs a{};
  a.slice_se = ix.d_slice_se; a.ent = ix.d_ent; a.ref_of_rank = ix.d_ref_of_rank;
  a.weight_of_rank = ix.d_weight_of_rank; a.n_refs = ix.n_refs; a.n_windows = ix.n_windows;
  a.offsets = d_offsets; a.qcodes = static_cast<const uint16_t*>(m->ws_codes.p);
  a.q_ntri = q_ntri; a.q_nb = q_nb; a.q_start = q_start; a.win_max_tri = ix.d_win_max_tri; a.nib_windows = ix.nib_windows; a.results = d_results; a.counts = d_counts; a.limit = limit;
  a.floor = floor;
  a.tomb = d_tomb;
  a.dense_min8 = ix.dense_min8;
  a.nm_dense = std::max((m->nm_dense + 7u) & ~7u, ix.dense_min8);
  a.nm_cmin = 0;                                     // (set per launch sequence: see "WHICH sweep" below)
  a.stats = m->collect_stats ? m->d_stats : nullptr;
  const bool cb = a.stats != nullptr;
  if (cb) {                                          // wave 0's phase clocks per workgroup (counted build only)
    if (!m->d_phase) BLURRILY_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->d_phase), kPhaseBytes));
    BLURRILY_HIP_TRY(hipMemsetAsync(m->d_phase, 0, kPhaseBytes, stream));
    a.phase_clocks = m->d_phase;
    a.path_flags = static_cast<uint32_t*>(m->ws_flags.p);   // (sized and zeroed by run_find)
  }
#ifdef BLURRILY_TRACE
  if (!cb) {                                         // (trace build: time stamps of a few needles' steps, timed kernels)
    if (!m->d_phase) BLURRILY_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->d_phase), kPhaseBytes));
    BLURRILY_HIP_TRY(hipMemsetAsync(m->d_phase, 0, kPhaseBytes, stream));
    a.phase_clocks = m->d_phase;
  }
#endif
  // every launch gets its own zeroed queue word (scalars[3..63]); recycled in stream order
  uint32_t queue_slot = 3;
  auto next_queue = [&]() -> uint32_t* {
    if (queue_slot >= 64) {
      if (hipMemsetAsync(scalars + 3, 0, 244, stream) != hipSuccess) return nullptr;
      queue_slot = 3;
    }
    return scalars + queue_slot++;
  };

  if (limit == 0) {
    BLURRILY_HIP_TRY(hipMemsetAsync(d_counts, 0, n * sizeof(uint32_t), stream));
  } else {
    // Latency mode: a batch too small to fill the GPU cuts every needle's windows into ranges
    // swept by different workgroups, then merges the per-range candidates (single pass only).
    const size_t wgs = size_t(m->n_cus) * find_wgs_per_cu();
    const uint32_t ranges = latency_ranges(n, limit, ix.n_windows, wgs, m->latency_tasks);
    if (ranges > 1) {
      const size_t tasks = n * ranges;
      const size_t key_bytes = align_up(tasks * limit * 8, 256);
      if (m->ws_parts.reserve(key_bytes + align_up(tasks * 4, 256), stream) < 0) return -1;
      a.work_list = nullptr; a.n_work_dev = nullptr; a.n_work = uint32_t(tasks);
      a.ranges = ranges;
      a.part_keys = static_cast<unsigned long long*>(m->ws_parts.p);
      a.part_count = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(m->ws_parts.p) + key_bytes);
      a.pass_base = 0; a.keep = limit; a.pool_cap = find_pool_cap(limit);
      if (!(a.queue = next_queue())) { errno = EIO; return -1; }
      // (every task writes its part_count, also the ones the byte-counter kernel skips)
      a.short_only = 1;
      if (do_launch_find(cb, a, false, uint32_t(std::min<size_t>(tasks, wgs)), stream) < 0) return -1;
      uint32_t merge_cap = 1024;
      while (merge_cap < ranges * limit) merge_cap <<= 1;
      a.pool_cap = merge_cap;
      if (launch_merge_parts(a, uint32_t(n), stream) < 0) return -1;
      a.ranges = 0; a.part_keys = nullptr; a.part_count = nullptr; a.short_only = 0;
      if (maybe_mid) {                               // 65..127 distinct trigrams: whole needle per workgroup
        a.work_list = mid_list; a.n_work_dev = scalars + 1; a.n_work = 0;
        a.pool_cap = find_pool_cap(a.keep);
        if (!(a.queue = next_queue())) { errno = EIO; return -1; }
        if (do_launch_find(cb, a, false, uint32_t(std::min<size_t>(n, wgs)), stream) < 0) return -1;
      }
    }
    // Large batches over many windows: the window-major sweep (find_kernels.hip, wsweep_kernel).
    // Phase 1 -- the needle-major kernel over the window pair of every needle's own length class --
    // seeds the needles' states; one launch per window follows; keys become rows at the end.
    auto run_ws = [&]() -> int {
      a.work_list = nullptr; a.n_work_dev = nullptr; a.n_work = uint32_t(n);
      a.pass_base = 0; a.keep = limit; a.pool_cap = find_pool_cap(limit);
      a.cmin = m->ws_cmin;
      // Phase 1: the needle-major kernel over the window pair of every needle's own length class seeds the
      // states (a needle's best matches live there, so its threshold is tight before the other windows are
      // visited).  (Seeding through wsweep_kernel's own robust path instead -- own_pass launches -- was
      // measured: configs[2] 321 -> 355 ms per 300 k needles, configs[4] 82 -> 129 ms: without a threshold
      // the 4-wave task floods its pool again and again where the 16-wave kernel bisects once.  Phase 1 over the
      // ONE window of the length class, in byte counters, the sibling window left to the window-major launches:
      // configs[4] 69.0 -> 71.8 ms per 100 k needles, Geonames scale 266 -> 275 ms per 300 k, round 3.)
      if (!(a.queue = next_queue())) { errno = EIO; return -1; }
      a.short_only = 1; a.own_only = 1;
      if (do_launch_find(cb, a, false, uint32_t(std::min<size_t>(n, wgs)), stream) < 0) return -1;
      a.own_only = 0;
      for (uint32_t w = 0; w < ix.n_windows; ++w) {
        if (!(a.queue = next_queue())) { errno = EIO; return -1; }
        if ((cb ? counted::launch_wsweep(a, w, uint32_t(n), uint32_t(m->n_cus), false, stream)
                : launch_wsweep(a, w, uint32_t(n), uint32_t(m->n_cus), false, stream)) < 0) return -1;
      }
      if ((cb ? counted::launch_finalize_rows(a, uint32_t(n), stream) : launch_finalize_rows(a, uint32_t(n), stream)) < 0) return -1;
      a.short_only = 0;
      if (maybe_mid) {                               // 65..127 distinct trigrams: needle-major, all windows
        a.work_list = mid_list; a.n_work_dev = scalars + 1; a.n_work = 0;
        if (!(a.queue = next_queue())) { errno = EIO; return -1; }
        if (do_launch_find(cb, a, false, uint32_t(std::min<size_t>(n, wgs)), stream) < 0) return -1;
      }
      return 0;
    };
    // needles with <= 127 distinct trigrams, needle-major: byte counters, up to 1024 rows per pass
    auto run_nm = [&]() -> int {
      for (uint32_t base = 0; base < limit; base += 1024) {
        a.work_list = nullptr; a.n_work_dev = nullptr; a.n_work = uint32_t(n);
        a.pass_base = base; a.keep = std::min<uint32_t>(1024, limit - base);
        a.pool_cap = find_pool_cap(a.keep);
        if (!(a.queue = next_queue())) { errno = EIO; return -1; }
        const uint32_t grid = uint32_t(std::min<size_t>(n, wgs));
        a.short_only = 1;                              // needles with <= 64 distinct trigrams
        if (do_launch_find(cb, a, false, grid, stream) < 0) return -1;
        a.short_only = 0;
        if (maybe_mid) {                               // 65..127: the tokeniser's mid list
          a.work_list = mid_list; a.n_work_dev = scalars + 1; a.n_work = 0;
          if (!(a.queue = next_queue())) { errno = EIO; return -1; }
          if (do_launch_find(cb, a, false, uint32_t(std::min<size_t>(n, wgs)), stream) < 0) return -1;
        }
      }
      return 0;
    };
    // An image of a few windows, a large batch, a limit of at most 64: the small-haystack sweep (find_small_kernel) --
    // four waves and one window's 4-bit counters per needle, four needles' chains per CU instead of two -- for the
    // needles of at most 15 trigrams; the ones it lists (16..64) follow through the byte-counter kernel.
    auto run_small = [&]() -> int {
      a.work_list = nullptr; a.n_work_dev = nullptr; a.n_work = uint32_t(n);
      a.pass_base = 0; a.keep = limit; a.pool_cap = find_pool_cap(limit);
      a.over_list = over_list; a.over_count = scalars + 2;
      if (!(a.queue = next_queue())) { errno = EIO; return -1; }
      if ((cb ? counted::launch_find_small(a, uint32_t(m->n_cus), stream) : launch_find_small(a, uint32_t(m->n_cus), stream)) < 0) return -1;
      a.work_list = over_list; a.n_work_dev = scalars + 2; a.n_work = 0;
      a.over_list = nullptr; a.over_count = nullptr;
      if (!(a.queue = next_queue())) { errno = EIO; return -1; }
      a.short_only = 1;
      if (do_launch_find(cb, a, false, uint32_t(std::min<size_t>(n, wgs)), stream) < 0) return -1;
      a.short_only = 0;
      if (maybe_mid) {                                 // 65..127: the tokeniser's mid list
        a.work_list = mid_list; a.n_work_dev = scalars + 1; a.n_work = 0;
        if (!(a.queue = next_queue())) { errno = EIO; return -1; }
        if (do_launch_find(cb, a, false, uint32_t(std::min<size_t>(n, wgs)), stream) < 0) return -1;
      }
      return 0;
    };
    // WHICH sweep serves the batch's short needles.  Three can: the needle-major sweep as it was through round 3
    // (1: every posting of every needle trigram counted), the needle-major sweep that leaves the largest dense
    // slices out of a step's count and settles candidates through bitmaps (3: "nm_cmin" > 0, limits up to 64), and
    // the window-major sweep (2: an image whose mean_hit_slice reaches "ws_min_slice", batches from "ws_min_needles"
    // on, limits up to 128).  No statistic of the image predicts the winner across kinds of haystack and of needles
    // (at the same mean_hit_slice one family of haystacks wins 1.4x with the window-major sweep
    // where another loses 0.7x; leaving slices out wins 14 % on a haystack four times Geonames scale, 3 % at
    // Geonames scale, and LOSES 9 % there on needles without a close match), so the choice is MEASURED: the first
    // batch of a class -- limit up to / above 32, by batch size 129.. / 16 384.. / 65 536.. / 262 144.. -- on an image runs
    // every sweep it can take (they give the same rows; that one call waits for them), the plain sweep twice -- the
    // first run of all meets cold caches -- and the fastest serves the class until the image is rebuilt or an option
    // changes; a sweep other than the plain one has to win by 1.5 % (window-major: 5 %, it pays a launch per window).
    // With "ws_autotune" 0, for smaller batches, and while request counters are collected on an unmeasured class, the
    // static rules apply: window-major by the measured table's mean_hit_slice rule, slices left out from 256 windows.
    const uint32_t cmin_opt = m->nm_cmin;
    const bool leave_possible = ranges <= 1 && cmin_opt != 0 && limit <= 1024 && find_can_leave(limit) && ix.n_bitmaps != 0;
    const bool ws_possible = ranges <= 1 && limit <= kWsMaxKeep && n >= m->ws_min_needles && ix.n_bitmaps != 0 &&
                             m->build_opt.ws_can_run(ix.n_windows, ix.mean_hit_slice) && code_slots < 0xFFFFFFFFull;
    auto run_sweep = [&](int which) -> int {           // 1 plain, 2 window-major, 3 slices left out, 4 small haystack
      a.nm_cmin = which == 3 ? cmin_opt : 0u;
      return which == 2 ? run_ws() : which == 4 ? run_small() : run_nm();
    };
    int choice = 1;
    a.nm_cmin = 0;                                     // (latency mode and the long-needle launches leave nothing out)
    const bool small_possible = ranges <= 1 && m->small_sweep && ix.n_windows <= kSmallMaxWindows && limit <= kSmallMaxKeep &&
                                n >= m->small_min_needles;
    if (small_possible) {
      if (run_sweep(4) < 0) return -1;
      if (is_base) m->last_sweep = 4;
    } else if (ranges <= 1) {
      // (a chunk of a host-buffer batch belongs to the class of the WHOLE batch: class_hint)
      const size_t n_cls = std::max(n, m->class_hint);
      const double slice_factor = (n_cls < 65536 ? (limit > 32 ? 4.0 : 1.7) : (limit > 32 ? 1.7 : 1.0));
      const int static_choice = ws_possible && ix.mean_hit_slice >= slice_factor * double(m->ws_static_slice) ? 2
                                : leave_possible && ix.n_windows >= m->nm_min_windows ? 3 : 1;
      // (classes 6 and 7: batches of 129 .. 16 383 needles -- a server's coalesced FINDs; at Geonames scale leaving slices
      // out wins there as it does on large batches: 0.9 -> 0.8 ms for 1 024 needles, 2.6 -> 2.1 for 4 096, 6.4 -> 5.4 for
      // 12 000, which the static rule -- from 256 windows on -- gave away through round 5's first half)
      const int cls = n_cls < 16384 ? (limit > 32 ? 7 : 6) : (limit > 32 ? 3 : 0) + (n_cls < 65536 ? 0 : n_cls < 262144 ? 1 : 2);
      const bool tunable = m->ws_autotune && is_base && n_cls >= 129 && (leave_possible || ws_possible);
      // what the class's last batch took, if it has finished (never waited for): slow against the measurement?
      if (tunable && !cb && m->watch_pending[cls] && hipEventQuery(m->watch_ev[cls][1]) == hipSuccess) {
        float ms = 0.f;
        m->watch_pending[cls] = false;
        // (only a batch of about the size the class was measured at is held against that figure: classes 6 / 7 span 129 ..
        // 16 383 needles, and a small batch's fixed costs -- 1.4 us a needle at 256 against 0.5 at 4 096 -- are not a slow sweep)
        const bool comparable = m->tuned_n[cls] != 0 && m->watch_n[cls] * 2 >= m->tuned_n[cls] && m->watch_n[cls] <= m->tuned_n[cls] * 2;
        if (hipEventElapsedTime(&ms, m->watch_ev[cls][0], m->watch_ev[cls][1]) == hipSuccess && m->watch_n[cls] && comparable &&
            m->tuned_us_per_needle[cls] > 0.f && m->ws_choice[cls] != 0) {
          const float us = 1000.f * ms / float(m->watch_n[cls]);
          // (TWO batches in a row: a single slow one -- seen on a shared box, 2.3 x inside bench.py's three timed steps --
          // would put a measurement of every sweep, twice, into a batch that had nothing wrong)
          if (us <= 1.10f * m->tuned_us_per_needle[cls]) {
            m->watch_strikes[cls] = 0;
          } else if (++m->watch_strikes[cls] >= 2 && m->retune_holdoff[cls] == 0) {
            m->ws_choice[cls] = 0;                     // measured again, below
            m->retune_holdoff[cls] = 16;
            m->watch_strikes[cls] = 0;
            ++m->retunes;
          }
        }
      }
      if (m->retune_holdoff[cls]) --m->retune_holdoff[cls];
      if (!tunable) {
        choice = static_choice;
      } else if (m->ws_choice[cls] != 0 && (m->ws_choice[cls] != 2 || ws_possible) && (m->ws_choice[cls] != 3 || leave_possible)) {
        choice = m->ws_choice[cls];
      } else if (cb) {
        choice = static_choice;                        // (counters must describe ONE sweep: an unmeasured class is not measured here)
      } else {
        if (!m->tune_ev[0]) {
          hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
          for (auto& e : ev)
            if (hipEventCreate(&e) != hipSuccess) {
              for (auto& d : ev) if (d) (void)hipEventDestroy(d);
              errno = EIO;
              return -1;
            }
          for (int i = 0; i < 7; ++i) m->tune_ev[i] = ev[i];
        }
        // every sweep the class can take, TWICE, the better run counting (the first run of all meets cold caches; one
        // sample per sweep with a 1.5 % margin -- rounds 3 and 4 -- sat inside run-to-run noise); the plain sweep goes last,
        // so that the rows in place are its
        const int order[6] = {1, leave_possible ? 3 : 0, ws_possible ? 2 : 0, leave_possible ? 3 : 0, ws_possible ? 2 : 0, 1};
        float ms_of[4] = {0.f, 0.f, 0.f, 0.f};         // by sweep: [1] plain, [2] window-major, [3] slices left out
        BLURRILY_HIP_TRY(hipEventRecord(m->tune_ev[0], stream));
        for (int k = 0; k < 6; ++k) {
          if (order[k] && run_sweep(order[k]) < 0) return -1;
          BLURRILY_HIP_TRY(hipEventRecord(m->tune_ev[k + 1], stream));
        }
        BLURRILY_HIP_TRY(hipEventSynchronize(m->tune_ev[6]));
        for (int k = 0; k < 6; ++k) {
          if (!order[k]) continue;
          float ms = 0.f;
          BLURRILY_HIP_TRY(hipEventElapsedTime(&ms, m->tune_ev[k], m->tune_ev[k + 1]));
          ms_of[order[k]] = ms_of[order[k]] == 0.f ? ms : std::min(ms_of[order[k]], ms);
        }
        if (m->tune_inject >= 1 && m->tune_inject <= 3) { ms_of[m->tune_inject] *= 0.5f; m->tune_inject = 0; }   // (tests)
        choice = 1;
        float best = ms_of[1];
        if (leave_possible && ms_of[3] < 0.97f * ms_of[1]) { choice = 3; best = ms_of[3]; }
        if (ws_possible && ms_of[2] < 0.95f * ms_of[1] && ms_of[2] < best) { choice = 2; best = ms_of[2]; }
        m->ws_choice[cls] = choice;
        m->ws_tuned_ms[cls][0] = ms_of[1]; m->ws_tuned_ms[cls][1] = ms_of[2]; m->ws_tuned_ms[cls][2] = ms_of[3];
        m->tuned_us_per_needle[cls] = 1000.f * best / float(n);
        m->tuned_n[cls] = n;
        m->watch_pending[cls] = false;
        m->last_tuned = cls;
        m->last_sweep = 1;                             // (the rows in place are the plain run's; all give the same)
        a.nm_cmin = 0;
        goto short_needles_done;
      }
      const bool watch = tunable && !cb && m->ws_choice[cls] == choice && m->tuned_us_per_needle[cls] > 0.f;
      if (watch) {
        if (!m->watch_ev[cls][1]) {                    // (both events or none: a half-made pair would be recorded into)
          hipEvent_t e0 = nullptr, e1 = nullptr;
          if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
            if (e0) (void)hipEventDestroy(e0);
            errno = EIO;
            return -1;
          }
          m->watch_ev[cls][0] = e0; m->watch_ev[cls][1] = e1;
        }
        BLURRILY_HIP_TRY(hipEventRecord(m->watch_ev[cls][0], stream));
      }
      if (run_sweep(choice) < 0) return -1;
      if (watch) {
        BLURRILY_HIP_TRY(hipEventRecord(m->watch_ev[cls][1], stream));
        m->watch_pending[cls] = true;
        m->watch_n[cls] = n;
      }
      if (is_base) m->last_sweep = choice;
      a.nm_cmin = 0;
    }
  short_needles_done:
    // longer needles: 16-bit counters, one workgroup per CU, 256 rows per pass
    if (maybe_long) {
      for (uint32_t base = 0; base < limit; base += 256) {
        a.work_list = big_list; a.n_work_dev = scalars; a.n_work = 0;
        a.pass_base = base; a.keep = std::min<uint32_t>(256, limit - base);
        a.pool_cap = 1024;
        if (!(a.queue = next_queue())) { errno = EIO; return -1; }
        const uint32_t grid = uint32_t(std::min<size_t>(n, size_t(m->n_cus)));
        if (do_launch_find(cb, a, true, grid, stream) < 0) return -1;
      }
    }
  }
  if (m->timing) {
    BLURRILY_HIP_TRY(hipEventRecord(m->ev[3], stream));
    BLURRILY_HIP_TRY(hipEventSynchronize(m->ev[3]));
    float ms = 0.f;
    BLURRILY_HIP_TRY(hipEventElapsedTime(&ms, m->ev[0], m->ev[1])); m->last_tok_ms = ms;
    BLURRILY_HIP_TRY(hipEventElapsedTime(&ms, m->ev[2], m->ev[3])); m->last_find_ms = ms;
  }
  return 0;
}

// Enqueue tokenise + find for n device-resident needles on the map's current contents.
int run_find(trigram_map m, const char* d_packed, size_t packed_bytes, const uint64_t* d_offsets, size_t n,
             uint16_t limit, trigram_match d_results, uint32_t* d_counts, uint32_t* d_nb, bool maybe_long,
             bool maybe_mid, hipStream_t stream) {
  if (m->collect_stats) {
    if (!m->d_stats) BLURRILY_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->d_stats), kStatAllSlots * 8));
    BLURRILY_HIP_TRY(hipMemsetAsync(m->d_stats, 0, kStatAllSlots * 8, stream));
    if (m->ws_flags.reserve(std::max<size_t>(n, 1) * sizeof(uint32_t), stream) < 0) return -1;
    BLURRILY_HIP_TRY(hipMemsetAsync(m->ws_flags.p, 0, std::max<size_t>(n, 1) * sizeof(uint32_t), stream));
    m->n_flags = n;
  }
  if (apply_tombstones(m, stream) < 0) return -1;
  if (log_empty(m))
    return run_find_on(m, m->dev, m->dev.d_code_total, nullptr, d_packed, packed_bytes, d_offsets, n, limit,
                       d_results, d_counts, d_nb, maybe_long, maybe_mid, stream);
  // base image (minus tombstones) and delta image hold disjoint references: find on both, merge
  const size_t row_bytes = std::max<size_t>(n * size_t(limit) * sizeof(trigram_match_t), 16);
  if (m->ws_base_rows.reserve(row_bytes, stream) < 0 || m->ws_base_counts.reserve(n * 4, stream) < 0 ||
      m->ws_delta_rows.reserve(row_bytes, stream) < 0 || m->ws_delta_counts.reserve(n * 4, stream) < 0)
    return -1;
  trigram_match base_rows = static_cast<trigram_match>(m->ws_base_rows.p);
  uint32_t* base_counts = static_cast<uint32_t*>(m->ws_base_counts.p);
  trigram_match delta_rows = static_cast<trigram_match>(m->ws_delta_rows.p);
  uint32_t* delta_counts = static_cast<uint32_t*>(m->ws_delta_counts.p);
  if (run_find_on(m, m->dev, m->d_code_total_now, log_of(m)->n_tomb ? m->dev.d_tomb : nullptr, d_packed, packed_bytes,
                  d_offsets, n, limit, base_rows, base_counts, d_nb, maybe_long, maybe_mid, stream) < 0)
    return -1;
  if (log_of(m)->pending.empty()) {
    BLURRILY_HIP_TRY(hipMemsetAsync(delta_counts, 0, n * 4, stream));
  } else if (run_find_on(m, m->delta, m->delta.d_code_total, nullptr, d_packed, packed_bytes, d_offsets, n, limit,
                         delta_rows, delta_counts, nullptr, maybe_long, maybe_mid, stream) < 0) {
    return -1;
  }
  return launch_merge_rows(base_rows, base_counts, delta_rows, delta_counts, uint32_t(n), limit, d_results,
                           d_counts, stream);
}

// ---- "devices" > 1: the batch sharded over replicas of the device image, in ONE process -----------------------
// (the drop-in host is a single process: the reference's server is one reactor, lib/blurrily/server.rb:19-30,
// its glue one call at a time, ext/blurrily/map_ext.c:131-162 -- SURVEY.md section 8(e)'s partition, replicate and
// shard contiguously, behind the C ABI instead of behind torch.distributed)

void free_replica(Replica& r) {
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (r.device >= 0) (void)hipSetDevice(r.device);
  if (r.stream) (void)hipStreamSynchronize(r.stream);
  if (r.side) {
    trigram_map s = r.side;
    if (s->dev.device >= 0) device_index_free(&s->dev);
    if (s->delta.device >= 0) device_index_free(&s->delta);
    if (s->d_code_total_now) (void)hipFree(s->d_code_total_now);
    if (s->d_stats) (void)hipFree(s->d_stats);
    if (s->d_phase) (void)hipFree(s->d_phase);
    s->ws_base_rows.release(); s->ws_base_counts.release(); s->ws_delta_rows.release(); s->ws_delta_counts.release();
    for (auto& e : s->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : s->tune_ev) if (e) (void)hipEventDestroy(e);
    for (auto& w : s->watch_ev) for (auto& e : w) if (e) (void)hipEventDestroy(e);
    s->ws_codes.release(); s->ws_small.release(); s->ws_parts.release(); s->ws_flags.release(); s->ws_tomb.release();
    delete s;
  }
  r.d_in.release(); r.d_out.release();
  for (hipEvent_t e : {r.ev_done, r.ev_t0, r.ev_t1}) if (e) (void)hipEventDestroy(e);
  if (r.stream) (void)hipStreamDestroy(r.stream);
  r = Replica();
  if (prev >= 0) (void)hipSetDevice(prev);
}

// Bring the replicas up to date with the primary's images (which ensure_device has just brought up to date with
// the host index): device-to-device clones of whatever changed -- the base image after a rebuild, the delta image
// when the set of pending puts changed, the tombstone bitmap and the bucket totals when anything was logged.
int ensure_replicas(trigram_map m) {
  int ndev = 0;
  BLURRILY_HIP_TRY(hipGetDeviceCount(&ndev));
  const size_t want = m->n_devices > 1 ? m->n_devices - 1 : 0;
  while (m->replicas.size() > want) { free_replica(m->replicas.back()); m->replicas.pop_back(); }
  while (m->replicas.size() < want) {
    Replica r;
    // replica k lives on the k-th device behind the primary's, round the visible ones (more replicas than devices --
    // the tests' way of running the multi-device path on one GPU -- share devices)
    r.device = (m->dev.device + 1 + int(m->replicas.size())) % ndev;
    r.side = new (std::nothrow) trigram_map_t();
    if (!r.side) { errno = ENOMEM; return -1; }
    r.side->mirror_of = m;
    // The shard's needles reach the replica, and its rows the caller's buffers, by hipMemcpyPeerAsync.  With peer
    // access enabled BOTH ways those copies are the devices' own, point to point (xGMI on an MI355X node); without it
    // the runtime stages them through host memory -- same rows, and said so once on stderr, since that is not the
    // gather SURVEY.md section 8(e) describes.
    r.same_device = r.device == m->dev.device;
    if (!r.same_device) {
      int to = 0, from = 0;
      const bool can = hipDeviceCanAccessPeer(&to, m->dev.device, r.device) == hipSuccess && to &&
                       hipDeviceCanAccessPeer(&from, r.device, m->dev.device) == hipSuccess && from;
      bool on_ = can;
      if (can) {
        for (int pass = 0; pass < 2 && on_; ++pass) {
          DeviceScope here(pass ? r.device : m->dev.device);
          const hipError_t e = hipDeviceEnablePeerAccess(pass ? m->dev.device : r.device, 0);
          if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
          else if (e != hipSuccess) on_ = false;
        }
      }
      r.peer_access = on_;
      if (!on_)
        std::fprintf(stderr, "blurrily_hip: no peer access between device %d and device %d (%s): the rows of that replica "
                             "travel through host memory\n", m->dev.device, r.device, can ? "enabling it failed" : "not offered");
    }
    DeviceScope on(r.device);
    if (hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&r.ev_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&r.ev_t0) != hipSuccess || hipEventCreate(&r.ev_t1) != hipSuccess) {
      free_replica(r);
      errno = EIO;
      return -1;
    }
    m->replicas.push_back(r);
  }
  if (!m->ev_ready) {
    BLURRILY_HIP_TRY(hipEventCreateWithFlags(&m->ev_ready, hipEventDisableTiming));
    BLURRILY_HIP_TRY(hipEventCreate(&m->ev_t0));
    BLURRILY_HIP_TRY(hipEventCreate(&m->ev_t1));
  }
  for (Replica& r : m->replicas) {
    trigram_map s = r.side;
    // options and measured choices follow the primary's
    s->build_opt = m->build_opt; s->ws_cmin = m->ws_cmin; s->nm_cmin = m->nm_cmin; s->nm_dense = m->nm_dense;
    s->ws_min_needles = m->ws_min_needles; s->ws_autotune = m->ws_autotune; s->ws_static_slice = m->ws_static_slice;
    s->nm_min_windows = m->nm_min_windows; s->small_sweep = m->small_sweep; s->small_min_needles = m->small_min_needles;
    for (int c = 0; c < 8; ++c) if (m->ws_choice[c]) s->ws_choice[c] = m->ws_choice[c];
    s->n_cus = 0;
    if (r.base_builds != m->base_builds || s->dev.device < 0) {
      if (device_index_clone(m->dev, r.device, &s->dev) < 0) return -1;
      std::fill(std::begin(s->ws_choice), std::end(s->ws_choice), 0);
      for (int c = 0; c < 8; ++c) s->ws_choice[c] = m->ws_choice[c];
      r.base_builds = m->base_builds;
      r.log_version = ~0ull;                                   // (tombstones and totals below)
      r.delta_image_version = ~0ull;
    }
    if (s->n_cus == 0) {
      hipDeviceProp_t prop;
      BLURRILY_HIP_TRY(hipGetDeviceProperties(&prop, r.device));
      s->n_cus = prop.multiProcessorCount;
    }
    if (r.delta_image_version != m->delta_image_version) {
      if (m->delta.device < 0) { if (s->delta.device >= 0) device_index_free(&s->delta); }
      else if (device_index_clone(m->delta, r.device, &s->delta) < 0) return -1;
      r.delta_image_version = m->delta_image_version;
    }
    if (r.log_version != m->log_version) {
      DeviceScope on(r.device);
      BLURRILY_HIP_TRY(hipMemcpyPeer(s->dev.d_tomb, r.device, m->dev.d_tomb, m->dev.device,
                                     ((size_t(m->dev.n_refs) + 31) / 32 + 1) * sizeof(uint32_t)));
      if (m->d_code_total_now) {
        if (!s->d_code_total_now)
          BLURRILY_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_code_total_now), kNumCodes * sizeof(uint32_t)));
        BLURRILY_HIP_TRY(hipMemcpyPeer(s->d_code_total_now, r.device, m->d_code_total_now, m->dev.device,
                                       kNumCodes * sizeof(uint32_t)));
      }
      r.log_version = m->log_version;
    }
  }
  return 0;
}

// n device-resident needles on the primary's device, results into buffers there, the work sharded contiguously over
// the primary and its replicas: every replica gets the batch's needles by ONE peer copy, searches its shard on its own
// stream and sends its block of rows (and counts, nb_entries) straight into the caller's buffers by peer copies --
// the gather of SURVEY.md section 8(e), point to point over xGMI.  Everything is enqueued: `stream` waits for the
// replicas' events, the host for nothing (timing mode apart).
int run_find_multi_enqueue(trigram_map m, const char* d_packed, size_t packed_bytes, const uint64_t* d_offsets, size_t n,
                           uint16_t limit, trigram_match d_results, uint32_t* d_counts, uint32_t* d_nb, hipStream_t stream) {
  if (apply_tombstones(m, stream) < 0) return -1;             // (the bits are set before the replicas copy the bitmap)
  if (ensure_replicas(m) < 0) return -1;
  const size_t R = m->replicas.size() + 1;
  const int P = m->dev.device;
  BLURRILY_HIP_TRY(hipEventRecord(m->ev_ready, stream));      // the caller's needles are in place behind this
  const bool timing = m->timing;
  auto bound = [&](size_t r) { return n * r / R; };
  const size_t off_bytes = align_up((n + 1) * sizeof(uint64_t), 256);
  for (size_t k = 0; k + 1 < R; ++k) {
    Replica& r = m->replicas[k];
    const size_t a = bound(k + 1), b = bound(k + 2), c = b - a;
    if (c == 0) continue;
    DeviceScope on(r.device);
    const size_t cnt_bytes = align_up(c * sizeof(uint32_t), 256);
    const size_t row_bytes = align_up(std::max<size_t>(c * size_t(limit) * sizeof(trigram_match_t), 16), 256);
    if (r.d_in.reserve(off_bytes + std::max<size_t>(packed_bytes, 16), r.stream) < 0 ||
        r.d_out.reserve(row_bytes + 2 * cnt_bytes, r.stream) < 0)
      return -1;
    unsigned char* in = static_cast<unsigned char*>(r.d_in.p);
    unsigned char* out = static_cast<unsigned char*>(r.d_out.p);
    BLURRILY_HIP_TRY(hipStreamWaitEvent(r.stream, m->ev_ready, 0));
    BLURRILY_HIP_TRY(hipMemcpyPeerAsync(in, r.device, d_offsets, P, (n + 1) * sizeof(uint64_t), r.stream));
    if (packed_bytes)
      BLURRILY_HIP_TRY(hipMemcpyPeerAsync(in + off_bytes, r.device, d_packed, P, packed_bytes, r.stream));
    trigram_match rows = reinterpret_cast<trigram_match>(out);
    uint32_t* counts = reinterpret_cast<uint32_t*>(out + row_bytes);
    uint32_t* nb = reinterpret_cast<uint32_t*>(out + row_bytes + cnt_bytes);
    r.side->timing = false;
    r.side->collect_stats = false;
    if (timing) BLURRILY_HIP_TRY(hipEventRecord(r.ev_t0, r.stream));
    // (the shard's offsets are the batch's own, from its first needle on: they index the whole needle buffer)
    if (run_find(r.side, reinterpret_cast<const char*>(in + off_bytes), packed_bytes,
                 reinterpret_cast<const uint64_t*>(in) + a, c, limit, rows, counts, d_nb ? nb : nullptr, true, true,
                 r.stream) < 0)
      return -1;
    if (timing) BLURRILY_HIP_TRY(hipEventRecord(r.ev_t1, r.stream));
    if (limit)
      BLURRILY_HIP_TRY(hipMemcpyPeerAsync(d_results + a * size_t(limit), P, rows, r.device,
                                          c * size_t(limit) * sizeof(trigram_match_t), r.stream));
    BLURRILY_HIP_TRY(hipMemcpyPeerAsync(d_counts + a, P, counts, r.device, c * sizeof(uint32_t), r.stream));
    if (d_nb) BLURRILY_HIP_TRY(hipMemcpyPeerAsync(d_nb + a, P, nb, r.device, c * sizeof(uint32_t), r.stream));
    BLURRILY_HIP_TRY(hipEventRecord(r.ev_done, r.stream));
  }
  // the primary's own shard, on the caller's stream (its timing mode would wait for it: the replicas are under way)
  const size_t c0 = bound(1);
  m->timing = false;
  if (timing) BLURRILY_HIP_TRY(hipEventRecord(m->ev_t0, stream));
  const int rc = run_find(m, d_packed, packed_bytes, d_offsets, c0, limit, d_results, d_counts, d_nb, true, true, stream);
  m->timing = timing;
  if (rc < 0) return -1;
  if (timing) BLURRILY_HIP_TRY(hipEventRecord(m->ev_t1, stream));
  for (size_t k = 0; k + 1 < R; ++k)
    if (bound(k + 2) > bound(k + 1)) BLURRILY_HIP_TRY(hipStreamWaitEvent(stream, m->replicas[k].ev_done, 0));
  if (timing) {                                               // last_find_kernel_ms: the slowest shard's search
    BLURRILY_HIP_TRY(hipStreamSynchronize(stream));
    float ms = 0.f, worst = 0.f;
    BLURRILY_HIP_TRY(hipEventElapsedTime(&worst, m->ev_t0, m->ev_t1));
    for (size_t k = 0; k + 1 < R; ++k) {
      if (bound(k + 2) == bound(k + 1)) continue;
      DeviceScope on(m->replicas[k].device);
      BLURRILY_HIP_TRY(hipEventElapsedTime(&ms, m->replicas[k].ev_t0, m->replicas[k].ev_t1));
      worst = std::max(worst, ms);
    }
    m->last_find_ms = worst;
    m->last_tok_ms = 0.0;
  }
  return 0;
}

// ... and what a failure half-way must not leave behind: the timing mode switched off, replicas still searching and
// copying into the caller's buffers
int run_find_multi(trigram_map m, const char* d_packed, size_t packed_bytes, const uint64_t* d_offsets, size_t n,
                   uint16_t limit, trigram_match d_results, uint32_t* d_counts, uint32_t* d_nb, hipStream_t stream) {
  const bool timing = m->timing;
  const int rc = run_find_multi_enqueue(m, d_packed, packed_bytes, d_offsets, n, limit, d_results, d_counts, d_nb, stream);
  if (rc < 0) {
    const int e = errno;
    m->timing = timing;
    for (Replica& r : m->replicas) {
      if (!r.stream) continue;
      DeviceScope on(r.device);
      (void)hipStreamSynchronize(r.stream);
    }
    (void)hipStreamSynchronize(stream);
    errno = e;
  }
  return rc;
}

// a batch goes over the replicas when "devices" asks for them and it is big enough to be worth a peer copy per
// device (and no request counters are being collected: they describe one launch sequence)
bool wants_multi(const trigram_map_t* m, size_t n) {
  return m->n_devices > 1 && !m->collect_stats && n >= size_t(1024) * m->n_devices;
}

}  // namespace

extern "C" {

int blurrily_storage_new(trigram_map* haystack) {
  trigram_map m = new (std::nothrow) trigram_map_t();
  if (!m) { errno = ENOMEM; return -1; }
  m->host = new (std::nothrow) HostIndex();
  if (!m->host) { delete m; errno = ENOMEM; return -1; }
  *haystack = m;
  return 0;
}

int blurrily_storage_load(trigram_map* haystack, const char* path) {
  HostIndex* ix = HostIndex::load(path);
  if (!ix) return -1;                                   // errno set by load()
  trigram_map m = new (std::nothrow) trigram_map_t();
  if (!m) { delete ix; errno = ENOMEM; return -1; }
  m->host = ix;
  *haystack = m;
  return 0;
}

int blurrily_storage_close(trigram_map* haystack) {
  trigram_map m = *haystack;
  if (m) {
    DeviceScope scope(m->dev.device);
    for (Replica& r : m->replicas) free_replica(r);
    m->replicas.clear();
    for (hipEvent_t e : {m->ev_ready, m->ev_t0, m->ev_t1}) if (e) (void)hipEventDestroy(e);
    if (m->dev.device >= 0) {
      (void)hipDeviceSynchronize();
      device_index_free(&m->dev);
    }
    if (m->delta.device >= 0) device_index_free(&m->delta);
    delete m->delta_host;
    if (m->d_code_total_now) (void)hipFree(m->d_code_total_now);
    if (m->d_stats) (void)hipFree(m->d_stats);
    if (m->d_phase) (void)hipFree(m->d_phase);
    m->ws_base_rows.release(); m->ws_base_counts.release(); m->ws_delta_rows.release(); m->ws_delta_counts.release();
    for (auto& e : m->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : m->tune_ev) if (e) (void)hipEventDestroy(e);
    for (auto& w : m->watch_ev) for (auto& e : w) if (e) (void)hipEventDestroy(e);
    m->ws_codes.release(); m->ws_small.release(); m->ws_parts.release(); m->ws_io_in.release();
    m->ws_io_out.release(); m->ws_tomb.release(); m->ws_flags.release();
    if (m->h_stage) (void)hipHostFree(m->h_stage);
    if (m->one.stream) { (void)hipStreamSynchronize(m->one.stream); (void)hipStreamDestroy(m->one.stream); }
    if (m->one.h_out) (void)hipHostFree(m->one.h_out);
    m->one.d_parts.release();
    for (int i = 0; i < 2; ++i) {
      if (m->pipe.h_in[i]) (void)hipHostFree(m->pipe.h_in[i]);
      if (m->pipe.h_out[i]) (void)hipHostFree(m->pipe.h_out[i]);
      m->pipe.d_in[i].release(); m->pipe.d_out[i].release();
      for (hipEvent_t e : {m->pipe.ev_in[i], m->pipe.ev_run[i], m->pipe.ev_out[i]}) if (e) (void)hipEventDestroy(e);
    }
    for (hipStream_t s : {m->pipe.s_in, m->pipe.s_run, m->pipe.s_out}) if (s) (void)hipStreamDestroy(s);
    delete m->host;
    delete m;
  }
  *haystack = nullptr;
  return 0;
}

void blurrily_storage_mark(trigram_map) {}

int blurrily_storage_save(trigram_map haystack, const char* path) { return haystack->host->save(path); }

int blurrily_storage_put(trigram_map haystack, const char* needle, uint32_t reference, uint32_t weight) {
  const size_t len = std::strlen(needle);
  const int added = haystack->host->put(needle, len, reference, weight);
  if (added > 0) log_put(haystack, needle, len, reference, weight);
  return added;
}

long blurrily_storage_put_many(trigram_map haystack, const char* packed, const uint64_t* offsets,
                               const uint32_t* references, const uint32_t* weights, size_t n) {
  // A bulk import -- no device image to keep in step, or more strings than the mutation log takes:
  // the image is rebuilt at the next find anyway -- goes through the parallel host path.
  const bool tracked = haystack->dev.device >= 0 && !haystack->log_overflow;
  if (n >= 65536 && (!tracked || n > log_budget(haystack))) {
    const long added = haystack->host->put_many(packed, offsets, references, weights, n);
    if (tracked && added > 0) { haystack->pending.clear(); haystack->log_overflow = true; }
    return added;
  }
  long total = 0;
  for (size_t i = 0; i < n; ++i) {
    const char* s = packed + offsets[i];
    const size_t cap = size_t(offsets[i + 1] - offsets[i]);
    const void* nul = std::memchr(s, 0, cap);
    const size_t len = nul ? size_t(static_cast<const char*>(nul) - s) : cap;
    const int added = haystack->host->put(s, len, references[i], weights ? weights[i] : 0u);
    if (added > 0) log_put(haystack, s, len, references[i], weights ? weights[i] : 0u);
    total += added;
  }
  return total;
}

int blurrily_storage_delete(trigram_map haystack, uint32_t reference) {
  const int removed = haystack->host->del(reference);
  if (removed > 0 && log_delete(haystack, reference) < 0) return -1;
  return removed;
}

int blurrily_storage_stats(trigram_map haystack, trigram_stat_t* stats) {
  stats->references = haystack->host->total_refs();
  stats->trigrams   = haystack->host->total_trigrams();
  return 0;
}

int blurrily_storage_sync_device(trigram_map haystack) {
  DeviceScope scope(haystack->dev.device);
  haystack->host->sort_dirty_buckets();
  return ensure_device(haystack);
}

int blurrily_storage_find_batch_device(trigram_map m, const char* d_packed, size_t packed_bytes,
                                       const uint64_t* d_offsets, size_t n, uint16_t limit,
                                       trigram_match d_results, uint32_t* d_counts,
                                       uint32_t* d_nb_entries, void* stream) {
  // The needles are not visible to the host here, so every dirty bucket is
  // sorted (the reference sorts only the needle's own, storage.c:516; results
  // are identical, see DESIGN.md "Mutation and device sync").
  DeviceScope scope(m->dev.device);
  if (m->host->dirty_buckets()) m->host->sort_dirty_buckets();
  if (ensure_device(m) < 0) return -1;
  if (m->timing && !m->ev[0])
    for (auto& e : m->ev) BLURRILY_HIP_TRY(hipEventCreate(&e));
  if (wants_multi(m, n))
    return run_find_multi(m, d_packed, packed_bytes, d_offsets, n, limit, d_results, d_counts, d_nb_entries,
                          static_cast<hipStream_t>(stream));
  return run_find(m, d_packed, packed_bytes, d_offsets, n, limit, d_results, d_counts, d_nb_entries, true,
                  true, static_cast<hipStream_t>(stream));
}

}  // extern "C"

// A large host-buffer batch in chunks through three streams: while chunk k is searched (s_run), chunk k+1's
// needles travel to the device (s_in) and chunk k-1's rows travel back (s_out) -- both through pinned staging,
// which the host fills / drains meanwhile.  Two slots by turns; a slot is reused only after its rows have been
// copied out to the caller.  Each element is still exactly one blurrily_storage_find.
static int find_batch_chunked_run(trigram_map m, const char* packed, const uint64_t* offsets, size_t n, uint16_t limit,
                              trigram_match results, uint32_t* counts, bool raw, uint32_t* non_ascii, size_t chunk) {
  auto& P = m->pipe;
  if (!P.s_in) {
    for (hipStream_t* s : {&P.s_in, &P.s_run, &P.s_out}) BLURRILY_HIP_TRY(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i)
      for (hipEvent_t* e : {&P.ev_in[i], &P.ev_run[i], &P.ev_out[i]})
        BLURRILY_HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
  }
  // staging sizes: the largest chunk's bytes in (rebased offsets | needles) and out (counts | flags | rows)
  size_t max_packed = 0;
  for (size_t a = 0; a < n; a += chunk) {
    const size_t b = std::min(n, a + chunk);
    max_packed = std::max<size_t>(max_packed, size_t(offsets[b] - offsets[a]));
  }
  const size_t off_cap = align_up((chunk + 1) * sizeof(uint64_t), 256);
  const size_t cnt_cap = align_up(chunk * sizeof(uint32_t), 256);
  const size_t flag_cap = raw ? cnt_cap : 0;
  const size_t in_cap = off_cap + std::max<size_t>(max_packed, 16);
  const size_t out_cap = cnt_cap + flag_cap + std::max<size_t>(chunk * size_t(limit) * sizeof(trigram_match_t), 16);
  if (P.h_in_bytes < in_cap || P.h_out_bytes < out_cap) {
    BLURRILY_HIP_TRY(hipDeviceSynchronize());
    for (int i = 0; i < 2; ++i) {
      if (P.h_in[i]) (void)hipHostFree(P.h_in[i]);
      if (P.h_out[i]) (void)hipHostFree(P.h_out[i]);
      P.h_in[i] = P.h_out[i] = nullptr;
    }
    P.h_in_bytes = P.h_out_bytes = 0;
    for (int i = 0; i < 2; ++i) {
      BLURRILY_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&P.h_in[i]), in_cap));
      BLURRILY_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&P.h_out[i]), out_cap));
    }
    P.h_in_bytes = in_cap; P.h_out_bytes = out_cap;
  }
  for (int i = 0; i < 2; ++i)
    if (P.d_in[i].reserve(in_cap, P.s_run) < 0 || P.d_out[i].reserve(out_cap, P.s_run) < 0) return -1;

  struct Span { size_t a, b; };
  Span in_slot[2] = {{0, 0}, {0, 0}};
  // rows of the chunk slot `i` holds, from pinned staging to the caller's buffers (behind its D2H)
  auto drain = [&](int i) -> int {
    const Span sp = in_slot[i];
    if (sp.b == sp.a) return 0;
    BLURRILY_HIP_TRY(hipEventSynchronize(P.ev_out[i]));
    const size_t c = sp.b - sp.a;
    std::memcpy(counts + sp.a, P.h_out[i], c * sizeof(uint32_t));
    if (raw && non_ascii) std::memcpy(non_ascii + sp.a, P.h_out[i] + cnt_cap, c * sizeof(uint32_t));
    if (limit)
      std::memcpy(results + sp.a * size_t(limit), P.h_out[i] + cnt_cap + flag_cap, c * size_t(limit) * sizeof(trigram_match_t));
    in_slot[i] = {0, 0};
    return 0;
  };
  size_t k = 0;
  for (size_t a = 0; a < n; a += chunk, ++k) {
    const size_t b = std::min(n, a + chunk), c = b - a;
    const int i = int(k & 1);
    if (drain(i) < 0) return -1;                                  // chunk k-2: its staging and device blocks are free again
    // ---- stage chunk k: offsets rebased to the chunk, its needles; how long they can be --------------
    uint64_t* h_off = reinterpret_cast<uint64_t*>(P.h_in[i]);
    const uint64_t base = offsets[a];
    size_t max_len = 0;
    for (size_t j = 0; j <= c; ++j) h_off[j] = offsets[a + j] - base;
    const size_t bytes = size_t(offsets[b] - base);
    if (bytes) std::memcpy(P.h_in[i] + off_cap, packed + base, bytes);
    for (size_t j = 0; j < c && max_len <= 126; ++j) {            // (what the launches need to know: > 63, > 126)
      const size_t cap = size_t(h_off[j + 1] - h_off[j]);
      if (cap <= max_len) continue;
      const char* s = packed + base + h_off[j];
      const void* nul = std::memchr(s, 0, cap);
      max_len = std::max(max_len, nul ? size_t(static_cast<const char*>(nul) - s) : cap);
    }
    unsigned char* d_in = static_cast<unsigned char*>(P.d_in[i].p);
    unsigned char* d_out = static_cast<unsigned char*>(P.d_out[i].p);
    BLURRILY_HIP_TRY(hipMemcpyAsync(d_in, P.h_in[i], off_cap + std::max<size_t>(bytes, 16), hipMemcpyHostToDevice, P.s_in));
    BLURRILY_HIP_TRY(hipEventRecord(P.ev_in[i], P.s_in));
    // ---- search it ---------------------------------------------------------------------------------
    BLURRILY_HIP_TRY(hipStreamWaitEvent(P.s_run, P.ev_in[i], 0));
    const uint64_t* d_offsets = reinterpret_cast<const uint64_t*>(d_in);
    char* d_packed = reinterpret_cast<char*>(d_in + off_cap);
    uint32_t* d_counts = reinterpret_cast<uint32_t*>(d_out);
    uint32_t* d_flags = reinterpret_cast<uint32_t*>(d_out + cnt_cap);
    trigram_match d_rows = reinterpret_cast<trigram_match>(d_out + cnt_cap + flag_cap);
    if (raw && launch_normalise(d_packed, d_offsets, uint32_t(c), d_packed, d_flags, P.s_run) < 0) return -1;
    if (run_find(m, d_packed, bytes, d_offsets, c, limit, d_rows, d_counts, nullptr, max_len > 126, max_len > 63,
                 P.s_run) < 0)
      return -1;
    BLURRILY_HIP_TRY(hipEventRecord(P.ev_run[i], P.s_run));
    // ---- and send its rows home ----------------------------------------------------------------------
    BLURRILY_HIP_TRY(hipStreamWaitEvent(P.s_out, P.ev_run[i], 0));
    BLURRILY_HIP_TRY(hipMemcpyAsync(P.h_out[i], d_out, cnt_cap + flag_cap + c * size_t(limit) * sizeof(trigram_match_t),
                                    hipMemcpyDeviceToHost, P.s_out));
    BLURRILY_HIP_TRY(hipEventRecord(P.ev_out[i], P.s_out));
    in_slot[i] = {a, b};
  }
  if (drain(int(k & 1)) < 0 || drain(int((k + 1) & 1)) < 0) return -1;
  return 0;
}

static int find_batch_chunked(trigram_map m, const char* packed, const uint64_t* offsets, size_t n, uint16_t limit,
                              trigram_match results, uint32_t* counts, bool raw, uint32_t* non_ascii, size_t chunk) {
  m->class_hint = n;
  const int rc = find_batch_chunked_run(m, packed, offsets, n, limit, results, counts, raw, non_ascii, chunk);
  m->class_hint = 0;
  if (rc < 0) {                                       // chunks may still be in flight on the three streams: let them
    const int e = errno;                              // finish before anybody reuses the slots
    (void)hipDeviceSynchronize();
    errno = e;
  }
  return rc;
}

// Blurrily::Map#normalize_string (lib/blurrily/map.rb:40-47) for ONE ASCII needle on the host -- normalise_kernel's two
// passes, byte for byte (kernels/tokenise.inc: downcase; unless some line is [a-z ]+ every byte that is not a-z becomes a
// space; whitespace runs squeezed, both ends stripped, trailing NULs too): a handful of raw needles is normalised here
// and shares find_one_kernel's launch instead of paying a copy in, a normalising launch and the batch's way.  `out` holds
// at least `cap` bytes; returns the normalised length (up to the first NUL: where the tokeniser stops); *high: whether
// the needle held a byte >= 0x80 (flagged, not guessed: NFKD is the caller's).
size_t normalise_one(const char* in, size_t cap, char* out, uint32_t* high_out) {
  bool plain = false, line_ok = true;
  size_t line_len = 0;
  uint32_t high = 0;
  for (size_t k = 0; k < cap; ++k) {
    unsigned char c = static_cast<unsigned char>(in[k]);
    high |= c >> 7;
    if (c == '\n') { plain |= line_ok && line_len > 0; line_ok = true; line_len = 0; continue; }
    if (c >= 'A' && c <= 'Z') c += 'a' - 'A';
    line_ok &= (c >= 'a' && c <= 'z') || c == ' ';
    ++line_len;
  }
  plain |= line_ok && line_len > 0;
  size_t w = 0, keep = 0;
  bool gap = false;
  for (size_t k = 0; k < cap; ++k) {
    unsigned char c = static_cast<unsigned char>(in[k]);
    if (c >= 'A' && c <= 'Z') c += 'a' - 'A';
    const bool letter = c >= 'a' && c <= 'z';
    if (!plain && !letter) c = ' ';
    if (c == ' ' || (c >= '\t' && c <= '\r')) { gap = true; continue; }
    if (gap && w > 0) out[w++] = ' ';
    gap = false;
    out[w++] = static_cast<char>(c);
    if (c != 0) keep = w;
  }
  if (high_out) *high_out = high;
  const void* nul = std::memchr(out, 0, keep);
  return nul ? size_t(static_cast<const char*>(nul) - out) : keep;
}

// (a handful of needles share one launch: find_few, below)
constexpr int kOneNotTaken = -2;
static int find_few(trigram_map m, const char* const* s, const size_t* len, size_t n, uint16_t limit, trigram_match results,
                    uint32_t* counts);

// Host-buffer batch: needles in, rows out.  raw = the needles are un-normalised ASCII (see
// blurrily_storage_find_batch_raw); non_ascii (raw only, may be null) receives the per-needle flags.
static int find_batch_host(trigram_map m, const char* packed, const uint64_t* offsets, size_t n, uint16_t limit,
                           trigram_match results, uint32_t* counts, bool raw, uint32_t* non_ascii) {
  if (n == 0) return 0;
  if (n <= std::max(m->one.few_max, m->one.mid_max)) { // a handful of needles: no copies (find_few; options "few_max", "mid_max")
    const char* s[kMidMaxNeedles];
    size_t len[kMidMaxNeedles];
    std::vector<char> norm;                             // raw needles: normalised here, as normalise_kernel would
    bool fits = true;
    if (raw) {
      if (offsets[n] - offsets[0] > (1u << 16)) fits = false;      // (a handful of very long needles: the batch's way)
      else norm.resize(size_t(offsets[n] - offsets[0]) + 1);
    }
    for (size_t i = 0; fits && i < n; ++i) {
      const size_t cap = size_t(offsets[i + 1] - offsets[i]);
      if (raw) {
        char* out = norm.data() + (offsets[i] - offsets[0]);
        uint32_t high = 0;
        len[i] = normalise_one(packed + offsets[i], cap, out, &high);
        s[i] = out;
        if (non_ascii) non_ascii[i] = high;
      } else {
        s[i] = packed + offsets[i];
        const void* nul = std::memchr(s[i], 0, cap);
        len[i] = nul ? size_t(static_cast<const char*>(nul) - s[i]) : cap;
      }
    }
    if (fits) {
      const int few = find_few(m, s, len, n, limit, results, counts);
      if (few != kOneNotTaken) return few;
    }
  }
  DeviceScope scope(m->dev.device);
  // what the reference's find does first: tokenise, sort the needle's dirty buckets
  size_t max_len = 0;
  const bool any_dirty = m->host->dirty_buckets() != 0;
  std::vector<uint16_t> codes;
  for (size_t i = 0; i < n; ++i) {
    const char* s = packed + offsets[i];
    const size_t cap = size_t(offsets[i + 1] - offsets[i]);
    const void* nul = std::memchr(s, 0, cap);
    const size_t len = nul ? size_t(static_cast<const char*>(nul) - s) : cap;
    max_len = std::max(max_len, len);
    if (any_dirty && !raw) {
      codes.resize(len + 1);
      const int nt = tokenise(s, len, codes.data());
      for (int k = 0; k < nt; ++k) m->host->sort_bucket_if_dirty(codes[k]);
    }
  }
  // (raw needles are only normalised on the device: sort every dirty bucket, as the device entry does)
  if (any_dirty && raw) m->host->sort_dirty_buckets();
  if (ensure_device(m) < 0) return -1;
  if (m->timing && !m->ev[0])
    for (auto& e : m->ev) BLURRILY_HIP_TRY(hipEventCreate(&e));
  // (timing and request counters describe ONE launch sequence: those runs stay in one piece)
  const bool multi = wants_multi(m, n);   // (the batch then goes in one piece through the primary: its rows come home over ONE PCIe link)
  if (!multi && m->host_chunk && n >= 2 * size_t(m->host_chunk) && !m->timing && !m->collect_stats) {
    size_t chunk = m->host_chunk;
    const size_t row_cap = size_t(32) << 20;                      // at most 32 MiB of rows per chunk in pinned staging
    while (chunk > 1024 && chunk * size_t(limit) * sizeof(trigram_match_t) > row_cap) chunk >>= 1;
    return find_batch_chunked(m, packed, offsets, n, limit, results, counts, raw, non_ascii, chunk);
  }

  hipStream_t stream = nullptr;
  const size_t packed_bytes = size_t(offsets[n]);
  const size_t off_bytes = (n + 1) * sizeof(uint64_t);
  const size_t row_bytes = n * size_t(limit) * sizeof(trigram_match_t);
  const size_t cnt_bytes = n * sizeof(uint32_t);
  // one device block in ([offsets | needles]) and one out ([counts | rows])
  const size_t in_bytes = align_up(off_bytes, 256) + std::max<size_t>(packed_bytes, 16);
  const size_t flag_bytes = raw ? align_up(cnt_bytes, 256) : 0;          // [counts | flags | rows]
  const size_t out_bytes = align_up(cnt_bytes, 256) + flag_bytes + std::max<size_t>(row_bytes, 16);
  if (m->ws_io_in.reserve(in_bytes, stream) < 0 || m->ws_io_out.reserve(out_bytes, stream) < 0) return -1;
  unsigned char* d_in = static_cast<unsigned char*>(m->ws_io_in.p);
  unsigned char* d_out = static_cast<unsigned char*>(m->ws_io_out.p);
  const uint64_t* d_offsets = reinterpret_cast<const uint64_t*>(d_in);
  char* d_packed = reinterpret_cast<char*>(d_in + align_up(off_bytes, 256));
  uint32_t* d_counts = reinterpret_cast<uint32_t*>(d_out);
  uint32_t* d_flags = reinterpret_cast<uint32_t*>(d_out + align_up(cnt_bytes, 256));
  trigram_match d_rows = reinterpret_cast<trigram_match>(d_out + align_up(cnt_bytes, 256) + flag_bytes);

  // Small batches (the single blurrily_storage_find above all) go through pinned staging: one
  // copy in, one copy out, instead of four pageable ones.
  const bool staged = in_bytes <= kStageBytes && out_bytes <= kStageBytes;
  if (staged && !m->h_stage) BLURRILY_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->h_stage), 2 * kStageBytes));
  if (staged) {
    unsigned char* h_in = m->h_stage;
    std::memcpy(h_in, offsets, off_bytes);
    if (packed_bytes) std::memcpy(h_in + align_up(off_bytes, 256), packed, packed_bytes);
    BLURRILY_HIP_TRY(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, stream));
  } else {
    BLURRILY_HIP_TRY(hipMemcpyAsync(d_in, offsets, off_bytes, hipMemcpyHostToDevice, stream));
    if (packed_bytes)
      BLURRILY_HIP_TRY(hipMemcpyAsync(d_in + align_up(off_bytes, 256), packed, packed_bytes, hipMemcpyHostToDevice,
                                      stream));
  }
  if (raw && launch_normalise(d_packed, d_offsets, uint32_t(n), d_packed, d_flags, stream) < 0) return -1;
  if ((multi ? run_find_multi(m, d_packed, packed_bytes, d_offsets, n, limit, d_rows, d_counts, nullptr, stream)
             : run_find(m, d_packed, packed_bytes, d_offsets, n, limit, d_rows, d_counts, nullptr, max_len > 126,
                        max_len > 63, stream)) < 0)
    return -1;
  if (staged) {
    unsigned char* h_out = m->h_stage + kStageBytes;
    BLURRILY_HIP_TRY(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, stream));
    BLURRILY_HIP_TRY(hipStreamSynchronize(stream));
    std::memcpy(counts, h_out, cnt_bytes);
    if (raw && non_ascii) std::memcpy(non_ascii, h_out + align_up(cnt_bytes, 256), cnt_bytes);
    if (limit) std::memcpy(results, h_out + align_up(cnt_bytes, 256) + flag_bytes, row_bytes);
  } else {
    BLURRILY_HIP_TRY(hipMemcpyAsync(counts, d_counts, cnt_bytes, hipMemcpyDeviceToHost, stream));
    if (raw && non_ascii) BLURRILY_HIP_TRY(hipMemcpyAsync(non_ascii, d_flags, cnt_bytes, hipMemcpyDeviceToHost, stream));
    if (limit) BLURRILY_HIP_TRY(hipMemcpyAsync(results, d_rows, row_bytes, hipMemcpyDeviceToHost, stream));
    BLURRILY_HIP_TRY(hipStreamSynchronize(stream));
  }
  return 0;
}

// ---- ONE needle, the caller waiting -- or a handful: one launch, no copies (find_kernels.hip: find_one_kernel) --------
// The reference's only call shape (ext/blurrily/map_ext.c:131-162 -> storage.c:477-580), and small host-buffer batches
// (a server's coalesced FINDs under light load).  needle i = s[i][0 .. len[i]) (up to its first NUL).  Returns 0 with
// counts[] and rows filled (results + i * limit), -1 with errno, or kOneNotTaken when the finds have to go the batch's
// way: a limit of 0 or above kOneMaxKeep, more than kMidMaxNeedles needles, a needle of more than 64 distinct trigrams,
// timing or request counters switched on, option "one_launch" 0.  Mutations the base image does not hold yet are
// served: tombstones inside the select, pending puts by a second launch over the delta image.
constexpr size_t kOneRowBytes = kOneMaxKeep * sizeof(trigram_match_t);
// the pinned page per image: [kMidMaxNeedles] rows | [kMidMaxNeedles][2] count, sequence word | codes [kMidMaxNeedles][64] | T
constexpr size_t kOneWordsAt = kMidMaxNeedles * kOneRowBytes;
constexpr size_t kOneCodesAt = kOneWordsAt + kMidMaxNeedles * 8 + 64;
constexpr size_t kOneTAt = kOneCodesAt + kMidMaxNeedles * 64 * sizeof(uint16_t);
// ... | postings [kMidMaxNeedles] | start window [kMidMaxNeedles] | code offsets [kMidMaxNeedles + 1] (latency mode's needle arrays)
constexpr size_t kMidNbAt = kOneTAt + kMidMaxNeedles * sizeof(uint32_t);
constexpr size_t kMidStartAt = kMidNbAt + kMidMaxNeedles * sizeof(uint32_t);
constexpr size_t kMidOffAt = (kMidStartAt + kMidMaxNeedles * sizeof(uint32_t) + 7) & ~size_t(7);
constexpr size_t kOneHostBytes = kMidOffAt + (kMidMaxNeedles + 1) * sizeof(uint64_t) + 64;
// lists a launch may leave: up to sixteen rows of kOneMaxGrid workgroups, or more rows of fewer (find_few aims at a
// thousand workgroups in all)
constexpr size_t kOneMaxLists = size_t(kOneMaxNeedles) * kOneMaxGrid;

static int find_few(trigram_map m, const char* const* s, const size_t* len, size_t n, uint16_t limit, trigram_match results,
                    uint32_t* counts) {
  if (!m->one.enabled || limit == 0 || limit > kOneMaxKeep || n == 0 || n > kMidMaxNeedles || m->timing || m->collect_stats)
    return kOneNotTaken;
  uint16_t codes[kMidMaxNeedles * 64];
  uint32_t T[kMidMaxNeedles];
  {
    uint16_t buf[256];
    for (size_t i = 0; i < n; ++i) {
      if (len[i] > 255) return kOneNotTaken;
      const int t = tokenise(s[i], len[i], buf);            // tokeniser.c:59-119
      if (t > 64) return kOneNotTaken;
      T[i] = uint32_t(t);
      std::memcpy(codes + i * 64, buf, size_t(t) * sizeof(uint16_t));
    }
  }
  DeviceScope scope(m->dev.device);
  // what the reference's find does first: sort the needle's dirty buckets (storage.c:516), sum their sizes (:498-503)
  if (m->host->dirty_buckets())
    for (size_t i = 0; i < n; ++i)
      for (uint32_t k = 0; k < T[i]; ++k) m->host->sort_bucket_if_dirty(codes[i * 64 + k]);
  if (ensure_device(m) < 0) return -1;
  // Mutations the base image does not hold (DESIGN.md "Mutation and device sync"): deletes are tombstone bits the select
  // looks at, pending puts live in a small delta image searched by a SECOND launch; the two lists of a needle are
  // merged here (they hold disjoint references).  A log that has overflowed is folded by ensure_device above.
  const bool with_tomb = log_of(m)->n_tomb != 0, with_delta = !log_of(m)->pending.empty() && m->delta.device >= 0;
  // needles without a posting return no rows (storage.c:503) and take no row of the grid
  uint32_t row_of[kMidMaxNeedles], row_nb[kMidMaxNeedles], n_rows = 0;
  for (size_t i = 0; i < n; ++i) {
    uint64_t nb = 0;
    for (uint32_t k = 0; k < T[i]; ++k) nb += m->host->bucket(codes[i * 64 + k]).used;
    counts[i] = 0;
    if (nb == 0) continue;
    if (n_rows != i) { std::memmove(codes + n_rows * 64, codes + i * 64, 64 * sizeof(uint16_t)); T[n_rows] = T[i]; }
    row_nb[n_rows] = nb > 0xFFFFFFFFull ? 0xFFFFFFFFu : uint32_t(nb);
    row_of[n_rows++] = uint32_t(i);
  }
  if (n_rows == 0) return 0;
  auto& O = m->one;
  NameScope name_scope(&m->last_kernels);                 // (the launches below note their kernels' names in the map)
  m->last_kernels.clear();
  m->last_sweep = 0;                                      // (no sweep of a class of batches: a single launch, or latency mode)
  if (!O.h_out) {
    // stream, pinned page and its device address: built in locals and kept only when ALL of them exist (a half-made set
    // -- a stream without its page -- would have the next find skip this block and poll a null page)
    hipStream_t st = nullptr;
    unsigned char *h = nullptr, *d = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h), 2 * kOneHostBytes, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&d), h, 0);
    if (e != hipSuccess) {
      std::fprintf(stderr, "blurrily_hip: the single find's stream / pinned page: %s\n", hipGetErrorString(e));
      if (h) (void)hipHostFree(h);
      if (st) (void)hipStreamDestroy(st);
      errno = (e == hipErrorOutOfMemory) ? ENOMEM : EIO;
      return -1;
    }
    std::memset(h, 0, 2 * kOneHostBytes);
    O.stream = st; O.h_out = h; O.d_out = d;
  }
  if (with_tomb && apply_tombstones(m, O.stream) < 0) return -1;       // (deletes since the last find: their bits are set first)
  const size_t key_bytes = kOneMaxLists * kOneMaxKeep * 8, flag_bytes = kOneMaxLists * 4, ticket_bytes = (kMidMaxNeedles + 1) * 4;   // (+ latency mode's queue word)
  const size_t part_bytes = key_bytes + flag_bytes + ticket_bytes;
  if (!O.d_parts.p) {
    if (O.d_parts.reserve(2 * part_bytes, O.stream) < 0) return -1;
    BLURRILY_HIP_TRY(hipMemsetAsync(O.d_parts.p, 0, 2 * part_bytes, O.stream));
  }
  // More than few_max rows: the BASE image is searched in latency mode -- find_kernel<..., RANGED>, a needle's windows cut
  // into ranges, a task per workgroup: beyond about thirty needles its pipelined steps beat find_one_kernel's exact
  // selects -- but without the batch path's copies: the per-needle arrays its tokeniser would have left
  // on the device (trigram counts, postings, the window of the needle's own length class, where its codes start) are
  // written here, into the pinned page, and read over the link by the tasks; the merge writes rows, counts and
  // sequence words back into the page (merge_parts_pinned_kernel), where this thread polls them as it does
  // find_one_kernel's.  Two launches, no copy, no stream synchronise (the batch path: a copy in, the tokeniser, the
  // find, the merge, a copy out, a synchronise).  The delta image, a window or two, keeps find_one_kernel.
  const uint32_t mid_ranges = (n_rows > O.few_max && n_rows <= O.mid_max)
      ? latency_ranges(n_rows, limit, m->dev.n_windows, size_t(m->n_cus) * find_wgs_per_cu(), m->latency_tasks) : 1u;
  const bool mid = mid_ranges > 1;
  if (mid) {
    uint32_t* h_nb = reinterpret_cast<uint32_t*>(O.h_out + kMidNbAt);
    uint32_t* h_start = reinterpret_cast<uint32_t*>(O.h_out + kMidStartAt);
    uint64_t* h_off = reinterpret_cast<uint64_t*>(O.h_out + kMidOffAt);
    for (uint32_t r = 0; r < n_rows; ++r) {
      h_nb[r] = row_nb[r];
      h_start[r] = m->dev.h_start_win[std::min<size_t>(len[row_of[r]], 255)];   // (tokenise_kernel: start_win[len])
      h_off[r] = uint64_t(r) * 63;                        // a needle's codes start at qcodes + offsets[q] + q: [needle][64]
    }
    h_off[n_rows] = uint64_t(n_rows) * 63;
    __atomic_thread_fence(__ATOMIC_RELEASE);
  }
  // more than kOneMaxNeedles rows, or latency mode: the codes travel in the pinned page (both images' launches read the first image's copy)
  const bool far = n_rows > kOneMaxNeedles;
  if (far || mid) {
    std::memcpy(O.h_out + kOneCodesAt, codes, size_t(n_rows) * 64 * sizeof(uint16_t));
    std::memcpy(O.h_out + kOneTAt, T, size_t(n_rows) * sizeof(uint32_t));
    __atomic_thread_fence(__ATOMIC_RELEASE);
  }
  const uint32_t seq = ++O.seq ? O.seq : ++O.seq;         // (never 0: what the words hold before the first find)
  // one launch per image: [0] the base image, [1] the delta image of the pending puts (its own lists, flags and rows)
  auto launch_on = [&](const DeviceIndex& ix, const uint32_t* d_tomb, int which) -> int {
    FindArgs a{};
    a.slice_se = ix.d_slice_se; a.ent = ix.d_ent; a.ref_of_rank = ix.d_ref_of_rank;
    a.weight_of_rank = ix.d_weight_of_rank; a.n_refs = ix.n_refs; a.n_windows = ix.n_windows;
    a.win_max_tri = ix.d_win_max_tri; a.nib_windows = ix.nib_windows; a.dense_min8 = ix.dense_min8;
    a.limit = limit; a.keep = limit; a.pool_cap = 512;
    a.tomb = d_tomb;
#ifdef BLURRILY_TRACE
    if (!m->d_phase) BLURRILY_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->d_phase), kPhaseBytes));
    if (which == 0) a.phase_clocks = m->d_phase;       // (trace build: find_one_kernel's wall-clock marks, 16 per workgroup)
#endif
    unsigned char* dp = static_cast<unsigned char*>(O.d_parts.p) + which * part_bytes;
    unsigned char* d_rows = O.d_out + which * kOneHostBytes;
    if (mid && which == 0) {
      const size_t wgs = size_t(m->n_cus) * find_wgs_per_cu();
      const uint32_t tasks = n_rows * mid_ranges;           // (<= 2 wgs: far below kOneMaxLists, whose keys and flags it borrows)
      a.offsets = reinterpret_cast<const uint64_t*>(O.d_out + kMidOffAt);
      a.qcodes = reinterpret_cast<const uint16_t*>(O.d_out + kOneCodesAt);
      a.q_ntri = reinterpret_cast<const uint32_t*>(O.d_out + kOneTAt);
      a.q_nb = reinterpret_cast<const uint32_t*>(O.d_out + kMidNbAt);
      a.q_start = reinterpret_cast<const uint32_t*>(O.d_out + kMidStartAt);
      a.nm_dense = std::max((m->nm_dense + 7u) & ~7u, ix.dense_min8);
      a.nm_cmin = 0;                                        // (ranges leave nothing out of a step's count: measured, slower)
      a.n_work = tasks; a.ranges = mid_ranges; a.short_only = 1;
      a.part_keys = reinterpret_cast<unsigned long long*>(dp);
      a.part_count = reinterpret_cast<uint32_t*>(dp + key_bytes);
      a.queue = reinterpret_cast<uint32_t*>(dp + key_bytes + flag_bytes) + kMidMaxNeedles;   // (zero between launches: the merge hands it back)
      a.pool_cap = find_pool_cap(limit);
      if (launch_find(a, false, uint32_t(std::min<size_t>(tasks, wgs)), O.stream) < 0) return -1;
      uint32_t merge_cap = 1024;
      while (merge_cap < mid_ranges * limit) merge_cap <<= 1;
      a.pool_cap = merge_cap;
      return launch_merge_parts_pinned(a, n_rows, reinterpret_cast<trigram_match_t*>(d_rows),
                                       reinterpret_cast<uint32_t*>(d_rows + kOneWordsAt), seq, O.stream);
    }
    // one window per workgroup while that fills at most kOneMaxGrid of them, whole window pairs beyond; more than
    // eight needles: about a thousand workgroups in all -- two rounds of what the chip holds --, i.e. several
    // window pairs a workgroup (its later steps arrive with its own threshold: one_select's cheap way)
    uint32_t per = 1;
    if (ix.n_windows > kOneMaxGrid) { per = (ix.n_windows + kOneMaxGrid - 1) / kOneMaxGrid; per += per & 1u; }
    if (n_rows > 8) {                                     // (from nine needles on: measured, tools/mid_probe.py)
      const uint32_t rows_wgs = std::max<uint32_t>(2u, m->one.mid_workgroups / n_rows);     // workgroups per needle
      const uint32_t per_far = (ix.n_windows + rows_wgs - 1) / rows_wgs;
      per = std::max(per, per_far + (per_far > 1 ? per_far & 1u : 0u));
    }
    per = std::max(per, O.min_per);                       // (a test's way to the several-steps-per-workgroup path on a small image)
    const uint32_t grid = (ix.n_windows + per - 1) / per;
    if (size_t(grid) * n_rows > kOneMaxLists) { errno = EINVAL; return -1; }   // (cannot happen: grid <= kOneMaxGrid, far grids are small)
    return launch_find_one(a, codes, T, n_rows, per, grid, reinterpret_cast<unsigned long long*>(dp),
                           reinterpret_cast<uint32_t*>(dp + key_bytes), reinterpret_cast<trigram_match_t*>(d_rows),
                           reinterpret_cast<uint32_t*>(d_rows + kOneWordsAt), seq, O.stream, uint32_t(m->n_cus),
                           far ? reinterpret_cast<const uint16_t*>(O.d_out + kOneCodesAt) : nullptr,
                           far ? reinterpret_cast<const uint32_t*>(O.d_out + kOneTAt) : nullptr,
                           far ? reinterpret_cast<uint32_t*>(dp + key_bytes + flag_bytes) : nullptr);
  };
  if (launch_on(m->dev, with_tomb ? m->dev.d_tomb : nullptr, 0) < 0) return -1;
  if (with_delta && launch_on(m->delta, nullptr, 1) < 0) return -1;
  // A row's last store is its sequence word; the host polls the words in the pinned page instead of waiting for the
  // runtime to notice the kernel's completion signal (an interrupt or a slower poll: 10 us and more).
  uint64_t spins = 0;
  for (int which = 0; which < (with_delta ? 2 : 1); ++which) {
    volatile uint32_t* words = reinterpret_cast<volatile uint32_t*>(O.h_out + which * kOneHostBytes + kOneWordsAt);
    for (uint32_t r = 0; r < n_rows; ++r) {
      while (words[2 * r + 1] != seq) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        if ((++spins & 0xFFFFFu) == 0) {                  // every few milliseconds: is the stream still alive?
          const hipError_t q = hipStreamQuery(O.stream);
          if (q == hipSuccess && words[2 * r + 1] != seq) { std::fprintf(stderr, "blurrily_hip: find_one finished without its rows\n"); errno = EIO; return -1; }
          if (q != hipSuccess && q != hipErrorNotReady) { std::fprintf(stderr, "blurrily_hip: find_one: %s\n", hipGetErrorString(q)); errno = EIO; return -1; }
        }
      }
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  const volatile uint32_t* w0 = reinterpret_cast<const volatile uint32_t*>(O.h_out + kOneWordsAt);
  const volatile uint32_t* w1 = reinterpret_cast<const volatile uint32_t*>(O.h_out + kOneHostBytes + kOneWordsAt);
  for (uint32_t r = 0; r < n_rows; ++r) {
    const uint32_t i = row_of[r];
    const trigram_match_t* a_rows = reinterpret_cast<const trigram_match_t*>(O.h_out + r * kOneRowBytes);
    const uint32_t got_a = w0[2 * r], na = got_a < limit ? got_a : uint32_t(limit);
    trigram_match_t* out = results + size_t(i) * limit;
    if (!with_delta) {
      counts[i] = na;
      std::memcpy(out, a_rows, size_t(na) * sizeof(trigram_match_t));
      continue;
    }
    // result order: matches descending, weight ascending, reference ascending (storage.c:129-138, :566)
    const trigram_match_t* b_rows = reinterpret_cast<const trigram_match_t*>(O.h_out + kOneHostBytes + r * kOneRowBytes);
    const uint32_t got_b = w1[2 * r], nb_ = got_b < limit ? got_b : uint32_t(limit);
    uint32_t ia = 0, ib = 0, k = 0;
    while (k < limit && (ia < na || ib < nb_)) {
      bool take_a;
      if (ia >= na) take_a = false;
      else if (ib >= nb_) take_a = true;
      else {
        const trigram_match_t x = a_rows[ia], y = b_rows[ib];
        take_a = x.matches != y.matches ? x.matches > y.matches : x.weight != y.weight ? x.weight < y.weight : x.reference < y.reference;
      }
      out[k++] = take_a ? a_rows[ia++] : b_rows[ib++];
    }
    counts[i] = k;
  }
  O.taken += n_rows;
  return 0;
}

extern "C" {

int blurrily_storage_find_batch(trigram_map m, const char* packed, const uint64_t* offsets, size_t n,
                                uint16_t limit, trigram_match results, uint32_t* counts) {
  return find_batch_host(m, packed, offsets, n, limit, results, counts, false, nullptr);
}

int blurrily_storage_find_batch_raw(trigram_map m, const char* packed, const uint64_t* offsets, size_t n,
                                    uint16_t limit, trigram_match results, uint32_t* counts,
                                    uint32_t* non_ascii) {
  return find_batch_host(m, packed, offsets, n, limit, results, counts, true, non_ascii);
}

int blurrily_normalize_batch_device(const char* d_packed, const uint64_t* d_offsets, size_t n, char* d_out,
                                    uint32_t* d_non_ascii, void* stream) {
  if (n > 0xFFFFFFFFull) { errno = EINVAL; return -1; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {   // no CPU fallback, here neither
    std::fprintf(stderr, "blurrily_hip: no usable HIP device\n");
    errno = ENODEV;
    return -1;
  }
  return launch_normalise(d_packed, d_offsets, uint32_t(n), d_out, d_non_ascii, static_cast<hipStream_t>(stream));
}

int blurrily_storage_find(trigram_map haystack, const char* needle, uint16_t limit, trigram_match results) {
  {
    const size_t len = std::strlen(needle);
    uint32_t n_rows = 0;
    const int one = find_few(haystack, &needle, &len, 1, limit, results, &n_rows);
    if (one != kOneNotTaken) return one < 0 ? one : int(n_rows);
  }
  const uint64_t offsets[2] = {0, std::strlen(needle)};
  uint32_t count = 0;
  // the device writes `limit` rows per needle; go through a scratch so a short
  // caller buffer is never over-written beyond `limit` rows (it is exactly limit rows)
  if (blurrily_storage_find_batch(haystack, needle, offsets, 1, limit, results, &count) < 0) return -1;
  return int(count);
}

int blurrily_tokeniser_parse_string(const char* input, uint16_t* output) {
  return tokenise(input, std::strlen(input), output);
}

int blurrily_storage_device_info(trigram_map m, blurrily_device_info_t* info) {
  info->device_ordinal = m->dev.device;
  info->n_refs = m->dev.n_refs;
  info->n_windows = m->dev.n_windows;
  info->window_bits = kWindowBits;
  info->n_entries = m->dev.n_entries;
  info->device_bytes = m->dev.device_bytes;
  info->last_find_kernel_ms = m->last_find_ms;
  info->last_tokenise_kernel_ms = m->last_tok_ms;
  info->n_pending = uint32_t(m->pending.size());
  info->n_tombstones = uint32_t(m->n_tomb);
  info->base_builds = m->base_builds;
  info->mean_hit_slice = m->dev.mean_hit_slice;
  info->n_bitmaps = m->dev.n_bitmaps;
  info->reserved_ = 0;
  info->dense_share = m->dev.dense_share;
  info->ws_gain = m->dev.ws_gain;
  // what the replicas sit on: physical devices by PCI bus id
  info->n_replicas = uint32_t(m->replicas.size());
  info->peer_access_mask = 0; info->same_device_mask = 0;
  std::memset(info->pci_bus_id, 0, sizeof info->pci_bus_id);
  std::vector<std::string> seen;
  auto bus_of = [](int dev) { char b[32] = {0}; if (dev < 0 || hipDeviceGetPCIBusId(b, sizeof b, dev) != hipSuccess) b[0] = 0; return std::string(b); };
  if (m->dev.device >= 0) {
    const std::string mine = bus_of(m->dev.device);
    std::snprintf(info->pci_bus_id, sizeof info->pci_bus_id, "%s", mine.c_str());
    seen.push_back(mine.empty() ? "dev" + std::to_string(m->dev.device) : mine);
  }
  for (size_t k = 0; k < m->replicas.size(); ++k) {
    const Replica& r = m->replicas[k];
    if (k < 32) { info->peer_access_mask |= uint32_t(r.peer_access) << k; info->same_device_mask |= uint32_t(r.same_device) << k; }
    std::string b = bus_of(r.device);
    if (b.empty()) b = "dev" + std::to_string(r.device);
    if (std::find(seen.begin(), seen.end(), b) == seen.end()) seen.push_back(b);
  }
  info->distinct_devices = uint32_t(seen.size());
  return 0;
}

size_t blurrily_storage_device_info_sized(trigram_map m, void* info, size_t info_size) {
  blurrily_device_info_t full;
  std::memset(&full, 0, sizeof full);
  (void)blurrily_storage_device_info(m, &full);
  std::memcpy(info, &full, std::min(info_size, sizeof full));
  return sizeof full;
}

int blurrily_storage_tune(trigram_map m, const char* packed, const uint64_t* offsets, size_t n_given, size_t n,
                          uint16_t limit) {
  if (!packed || !offsets || n_given == 0 || n == 0) { errno = EINVAL; return -1; }
  // n needles, the given ones over and over: one host-buffer batch in one piece, whose class is then measured by the
  // find itself if it has not been (run_find_on); the rows are thrown away
  std::vector<uint64_t> off(n + 1);
  std::vector<char> buf;
  off[0] = 0;
  for (size_t i = 0; i < n; ++i) {
    const size_t g = i % n_given;
    buf.insert(buf.end(), packed + offsets[g], packed + offsets[g + 1]);
    off[i + 1] = buf.size();
  }
  if (buf.empty()) buf.push_back(0);
  std::vector<trigram_match_t> rows(n * size_t(limit) + 1);
  std::vector<uint32_t> counts(n);
  const uint32_t chunk = m->host_chunk;
  m->host_chunk = 0;
  const int rc = blurrily_storage_find_batch(m, buf.data(), off.data(), n, limit, rows.data(), counts.data());
  m->host_chunk = chunk;
  return rc;
}

void blurrily_storage_set_timing(trigram_map m, int enabled) { m->timing = enabled != 0; }

void blurrily_storage_set_stats(trigram_map m, int enabled) { m->collect_stats = enabled != 0; }

// Tunables: one table, so that set and get cannot drift apart.  A map's options are plain fields read by its
// next find (calls on one map are serial, as in the reference: no lock); the process-wide ones are atomics.
namespace {
struct OptionSlot { const char* key; long long lo, hi; };
constexpr OptionSlot kMapOptions[] = {
    {"wsweep", 0, 1}, {"ws_cmin", 1, 64}, {"ws_min_windows", 0, 1 << 20}, {"ws_min_needles", 0, 1ll << 32},
    {"ws_min_slice", 0, 1ll << 31}, {"dense_min", 64, 65536}, {"host_chunk", 0, 1ll << 30},
    {"ws_autotune", 0, 1}, {"ws_static_slice", 0, 1ll << 31}, {"ws_choice", 0, 0},
    {"nm_cmin", 0, 64}, {"nm_dense", 64, 65536}, {"last_sweep", 0, 0}, {"devices", 1, 64},
    {"nm_min_windows", 0, 1 << 20}, {"tuned_class", 0, 0}, {"tuned_nm_us", 0, 0}, {"tuned_ws_us", 0, 0},
    {"tuned_leave_us", 0, 0}, {"small_sweep", 0, 1}, {"small_min_needles", 0, 1ll << 32},
    {"one_launch", 0, 1}, {"one_taken", 0, 0}, {"one_windows_per_wg", 0, 1 << 20},
    {"retunes", 0, 0}, {"tune_inject", 0, 3}, {"mid_workgroups", 64, 1 << 16}, {"few_max", 1, kMidMaxNeedles}, {"mid_max", 0, kMidMaxNeedles},
    {"latency_tasks", 0, 16}};
constexpr OptionSlot kProcessOptions[] = {{"host_threads", 0, 256}, {"build_trace", 0, 1}};
int find_option(const OptionSlot* tab, size_t n, const char* key) {
  for (size_t i = 0; i < n; ++i) if (std::strcmp(tab[i].key, key) == 0) return int(i);
  return -1;
}
#define FIND_OPTION(tab, key) find_option(tab, sizeof(tab) / sizeof(tab[0]), key)
}  // namespace

int blurrily_storage_set_option(trigram_map m, const char* key, long long value) {
  if (!key) { errno = EINVAL; return -1; }
  if (!m) {
    const int i = FIND_OPTION(kProcessOptions, key);
    if (i < 0 || value < kProcessOptions[i].lo || value > kProcessOptions[i].hi) { errno = EINVAL; return -1; }
    if (i == 0) set_host_threads(unsigned(value)); else set_build_trace(value != 0);
    return 0;
  }
  const int i = FIND_OPTION(kMapOptions, key);
  if (i < 0 || value < kMapOptions[i].lo || value > kMapOptions[i].hi) { errno = EINVAL; return -1; }
  switch (i) {
    case 0: m->build_opt.ws_enabled = value != 0; break;
    case 1: m->ws_cmin = uint32_t(value); break;
    case 2: m->build_opt.ws_min_windows = uint32_t(value); break;
    case 3: m->ws_min_needles = uint32_t(std::min<long long>(value, 0xFFFFFFFFll)); break;
    case 4: m->build_opt.ws_min_slice = uint32_t(value); break;
    case 5:                                        // which slices have bitmaps: the image is rebuilt by the next find
      if (m->build_opt.dense_min != uint32_t(value) && m->dev.device >= 0) m->log_overflow = true;
      m->build_opt.dense_min = uint32_t(value);
      break;
    case 6: m->host_chunk = uint32_t(value); break;
    case 7: m->ws_autotune = value != 0; break;
    case 8: m->ws_static_slice = uint32_t(value); break;
    case 9: break;                                       // (value 0 only: forget what was measured)
    case 10: m->nm_cmin = uint32_t(value); break;
    case 11: m->nm_dense = uint32_t(value); break;
    case 12: m->last_sweep = 0; return 0;                // (value 0 only; nothing to measure again)
    case 13:                                             // (replicas are made by the next large batch; dropped at once)
      m->n_devices = uint32_t(value);
      while (m->replicas.size() + 1 > m->n_devices) { free_replica(m->replicas.back()); m->replicas.pop_back(); }
      return 0;
    case 14: m->nm_min_windows = uint32_t(value); break;
    case 15: case 16: case 17: case 18: return 0;        // (read-only: what the last measurement saw)
    case 19: m->small_sweep = value != 0; break;
    case 20: m->small_min_needles = uint32_t(std::min<long long>(value, 0xFFFFFFFFll)); break;
    case 21: m->one.enabled = value != 0; return 0;      // (the single find's own launch; nothing to measure again)
    case 22: return 0;                                   // (read-only)
    case 23: m->one.min_per = uint32_t(value); return 0;
    case 24: return 0;                                   // (read-only)
    case 25: m->tune_inject = int(value); return 0;      // (tests: the next measurement's bad sample)
    case 26: m->one.mid_workgroups = uint32_t(value); return 0;
    case 27: m->one.few_max = uint32_t(value); return 0;
    case 28: m->one.mid_max = uint32_t(value); return 0;
    case 29: m->latency_tasks = uint32_t(value); return 0;
  }
  if (i != 6) std::fill(std::begin(m->ws_choice), std::end(m->ws_choice), 0);   // the sweep's choice is measured again
  return 0;
}

int blurrily_storage_get_option(trigram_map m, const char* key, long long* value) {
  if (!key || !value) { errno = EINVAL; return -1; }
  if (!m) {
    const int i = FIND_OPTION(kProcessOptions, key);
    if (i < 0) { errno = EINVAL; return -1; }
    *value = i == 0 ? (long long)host_threads() : (long long)build_trace();
    return 0;
  }
  switch (FIND_OPTION(kMapOptions, key)) {
    case 0: *value = m->build_opt.ws_enabled; return 0;
    case 1: *value = m->ws_cmin; return 0;
    case 2: *value = m->build_opt.ws_min_windows; return 0;
    case 3: *value = m->ws_min_needles; return 0;
    case 4: *value = m->build_opt.ws_min_slice; return 0;
    case 5: *value = m->build_opt.dense_min; return 0;
    case 6: *value = m->host_chunk; return 0;
    case 7: *value = m->ws_autotune; return 0;
    case 8: *value = m->ws_static_slice; return 0;
    case 9: {                                            // what was measured so far: class c's choice in bits 2c+1:2c
      long long v = 0;                                   // (1 needle-major, 2 window-major, 3 needle-major with slices left out)
      for (int c = 0; c < 8; ++c) v |= (long long)(m->ws_choice[c]) << (2 * c);
      *value = v;
      return 0;
    }
    case 10: *value = m->nm_cmin; return 0;
    case 11: *value = m->nm_dense; return 0;
    case 12: *value = m->last_sweep; return 0;
    case 13: *value = m->n_devices; return 0;
    case 14: *value = m->nm_min_windows; return 0;
    case 15: *value = m->last_tuned; return 0;             // -1: nothing measured yet
    case 16: case 17: case 18:                             // microseconds of the sweep in that measurement (0: it could not run)
      *value = m->last_tuned < 0 ? 0 : (long long)(1000.0 * m->ws_tuned_ms[m->last_tuned][FIND_OPTION(kMapOptions, key) - 16]);
      return 0;
    case 19: *value = m->small_sweep; return 0;
    case 20: *value = m->small_min_needles; return 0;
    case 21: *value = m->one.enabled; return 0;
    case 22: *value = (long long)m->one.taken; return 0;
    case 23: *value = m->one.min_per; return 0;
    case 24: *value = (long long)m->retunes; return 0;
    case 25: *value = m->tune_inject; return 0;
    case 26: *value = m->one.mid_workgroups; return 0;
    case 27: *value = m->one.few_max; return 0;
    case 28: *value = m->one.mid_max; return 0;
    case 29: *value = m->latency_tasks; return 0;
    default: errno = EINVAL; return -1;
  }
}

// debugging aid (tools/ws_probe.py): all counter slots, phase clocks of the window-major sweep included
int blurrily_debug_find_stats16(trigram_map m, uint64_t* out16) {
  std::memset(out16, 0, kStatAllSlots * 8);
  if (!m->d_stats) return 0;
  DeviceScope scope(m->dev.device);
  BLURRILY_HIP_TRY(hipDeviceSynchronize());
  BLURRILY_HIP_TRY(hipMemcpy(out16, m->d_stats, kStatAllSlots * 8, hipMemcpyDeviceToHost));
  return 0;
}

int blurrily_storage_find_stats(trigram_map m, uint64_t* out8) {
  std::memset(out8, 0, kStatSlots * 8);
  if (!m->d_stats) return 0;
  DeviceScope scope(m->dev.device);
  BLURRILY_HIP_TRY(hipDeviceSynchronize());
  BLURRILY_HIP_TRY(hipMemcpy(out8, m->d_stats, kStatSlots * 8, hipMemcpyDeviceToHost));
  return 0;
}

size_t blurrily_storage_last_kernels(trigram_map m, char* out, size_t cap) {
  const std::string& s = m->last_kernels;
  if (out && cap) {
    const size_t k = std::min(s.size(), cap - 1);
    std::memcpy(out, s.data(), k);
    out[k] = '\0';
  }
  return s.size();
}

int blurrily_storage_find_path_flags(trigram_map m, uint32_t* out, size_t n) {
  if (!m->ws_flags.p || n > m->n_flags) { errno = EINVAL; return -1; }
  DeviceScope scope(m->dev.device);
  BLURRILY_HIP_TRY(hipDeviceSynchronize());
  BLURRILY_HIP_TRY(hipMemcpy(out, m->ws_flags.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return 0;
}

// debugging aid (tools/phase_profile.py): per-workgroup phase clocks of the last find launched while
// blurrily_storage_set_stats was on
int blurrily_debug_phase_clocks(trigram_map m, unsigned long long* out, size_t n_workgroups) {
  if (!m->d_phase || n_workgroups > kPhaseWorkgroups) { errno = EINVAL; return -1; }
  DeviceScope scope(m->dev.device);
  BLURRILY_HIP_TRY(hipDeviceSynchronize());
  BLURRILY_HIP_TRY(hipMemcpy(out, m->d_phase, n_workgroups * 16 * 8, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
