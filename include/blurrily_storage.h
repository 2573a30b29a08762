/*
 * blurrily_storage.h -- C ABI of libblurrily_hip.so, the MI355X-native drop-in
 * for the trigram index behind Blurrily::Map (mezis/blurrily v1.0.2).
 *
 * Part 1 declares the nine entry points of the reference's
 * ext/blurrily/storage.h:36-117 with the same names, argument meaning and
 * error behaviour, so the reference's own Ruby glue (ext/blurrily/map_ext.c)
 * links against this library unchanged.  Part 2 adds the
 * batched / device-resident entry points the reference has no counterpart for;
 * each batched element is defined as exactly one blurrily_storage_find.
 *
 * `find` runs on the GPU (hand-written HIP kernels for gfx950).  There is no
 * CPU fallback: without a usable device every find entry point returns -1 with
 * errno = ENODEV and says so on stderr.
 */
#ifndef BLURRILY_AMD_STORAGE_H
#define BLURRILY_AMD_STORAGE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- types ---- */

/* When the reference's own ext/blurrily/storage.h has been included first (a translation unit of the gem's
 * glue, ruby/ext/blurrily/map_ext_batch.c), its types are the types: this header then only RE-DECLARES the nine
 * functions -- which the compiler checks for compatibility, tests/test_header_compat.py -- and adds part 2. */
#ifndef __STORAGE_H__

struct trigram_map_t;                               /* opaque; storage.h:15-16 */
typedef struct trigram_map_t* trigram_map;

/* storage.h:18-24 -- packed 12-byte result row. */
struct __attribute__((__packed__)) trigram_match_t {
  uint32_t reference;
  uint32_t matches;
  uint32_t weight;
};
typedef struct trigram_match_t  trigram_match_t;
typedef struct trigram_match_t* trigram_match;

/* storage.h:26-30 */
typedef struct trigram_stat_t {
  uint32_t references;
  uint32_t trigrams;
} trigram_stat_t;

#endif /* __STORAGE_H__ */

/* ------------------------------------------------- part 1: reference ABI ---- */

/* storage.h:36 / storage.c:178-206.  New empty map.  0 on success, <0 + errno. */
int blurrily_storage_new(trigram_map* haystack);

/* storage.h:41 / storage.c:210-266.  Load a `.trigrams` file written by this
 * library or by the reference (same bytes).  <0 + errno on failure: ENOENT
 * etc. from open(2); EPROTO for a short file, bad magic, wrong endianness or
 * pointer size (storage.c:226-230,245-250) and for out-of-bounds bucket
 * descriptors. */
int blurrily_storage_load(trigram_map* haystack, const char* path);

/* storage.h:46 / storage.c:270-295.  Release host and device memory; sets
 * *haystack = NULL. */
int blurrily_storage_close(trigram_map* haystack);

/* storage.h:51 / storage.c:625-629.  GC mark hook of the Ruby glue.  The ref
 * set here is native (no Ruby objects) so this is a no-op. */
void blurrily_storage_mark(trigram_map haystack);

/* storage.h:58 / storage.c:299-377.  Write the map to `path` atomically
 * (temp file + rename), byte-identical to what the reference writes for the
 * same sequence of operations. */
int blurrily_storage_save(trigram_map haystack, const char* path);

/* storage.h:70 / storage.c:398-473.  Returns the number of trigrams added,
 * 0 if `reference` is already present.  weight == 0 -> strlen(needle).
 * Out of memory: -1 with errno ENOMEM and no entry added (the reference
 * asserts in smalloc, storage.c:93-98; a bucket grown before the failing
 * allocation keeps its new capacity, which a later save writes out). */
int blurrily_storage_put(trigram_map haystack, const char* needle,
                         uint32_t reference, uint32_t weight);

/* storage.h:96 / storage.c:584-612.  Returns the number of entries removed. */
int blurrily_storage_delete(trigram_map haystack, uint32_t reference);

/* storage.h:110 / storage.c:477-580.  At most `limit` rows into the
 * caller-allocated `results`, ordered by matches descending, weight ascending,
 * reference ascending.  Returns the row count, or -1 (errno ENODEV) when no
 * GPU is usable.  ONE kernel launch and no copy for a needle of at most 64
 * distinct trigrams at a limit of 1..120 (option "one_launch"; a second launch
 * over the delta image while puts are pending); otherwise one element of
 * blurrily_storage_find_batch -- which takes the same launch for up to "few_max"
 * (24; the kernel: up to 128) such needles, a row of the grid each, and still no
 * copy for up to "mid_max" (128). */
int blurrily_storage_find(trigram_map haystack, const char* needle,
                          uint16_t limit, trigram_match results);

/* storage.h:117 / storage.c:616-621. */
int blurrily_storage_stats(trigram_map haystack, trigram_stat_t* stats);

/* ------------------------------------------- part 2: batched extensions ---- */

/* Same as `n` calls of blurrily_storage_put, in order (no reference
 * counterpart; avoids one FFI crossing per string for bulk imports).
 * Needle i is the bytes packed[offsets[i] .. offsets[i+1]) cut at the first
 * NUL.  weights may be NULL (all 0).  Returns the total number of trigrams
 * added, or -1. */
long blurrily_storage_put_many(trigram_map haystack, const char* packed,
                               const uint64_t* offsets, const uint32_t* references,
                               const uint32_t* weights, size_t n);

/* n independent finds on the GPU in one launch sequence.  Host buffers in,
 * host buffers out.  results holds n*limit rows (query i owns rows
 * [i*limit, i*limit+counts[i])); counts holds n row counts.  Returns 0, or -1
 * with errno (ENODEV: no GPU). */
int blurrily_storage_find_batch(trigram_map haystack, const char* packed,
                                const uint64_t* offsets, size_t n, uint16_t limit,
                                trigram_match results, uint32_t* counts);

/* Device-resident variant: every pointer is a device pointer on the map's GPU;
 * work is enqueued on `stream` (a hipStream_t; NULL = default stream) and the
 * call returns without waiting for it -- with these exceptions, each of which
 * blocks the calling thread until the work enqueued so far on `stream` is done:
 *   - the first call after a mutation builds / refreshes the device image
 *     (as blurrily_storage_sync_device would; call that first to keep it out);
 *   - the first call after a blurrily_storage_delete uploads the deleted ranks and
 *     sets their tombstone bits (a synchronous copy and a stream synchronise);
 *   - with "ws_autotune" 1 (the default), the FIRST batch of a class -- limit up
 *     to / above 32 x 129.. / 16 384.. / 65 536.. / 262 144.. needles -- on an image runs
 *     every sweep that can serve it and waits for them once, to note the fastest
 *     (blurrily_storage_tune does that ahead of time; "ws_autotune" 0 never does);
 *   - blurrily_storage_set_timing(1) and blurrily_storage_set_stats(1).
 * With "devices" > 1 the batch is sharded over the replicas; `stream` then waits
 * (on the device, not the host) for every replica's rows.
 * d_nb_entries (optional, may be NULL) receives per query the reference's
 * nb_entries (storage.c:498-502), the unit of the matched-entries/s metric. */
int blurrily_storage_find_batch_device(trigram_map haystack, const char* d_packed,
                                       size_t packed_bytes, /* == offsets[n] */
                                       const uint64_t* d_offsets, size_t n, uint16_t limit,
                                       trigram_match d_results, uint32_t* d_counts,
                                       uint32_t* d_nb_entries, void* stream);

/* Needle normalisation on the device -- Blurrily::Map#normalize_string
 * (lib/blurrily/map.rb:40-47) for ASCII needles: A-Z -> a-z, every other byte
 * that is not a-z -> ' ', runs of spaces squeezed, ends stripped.  The result is
 * written to `d_out` at the needle's own offset (never longer than the input;
 * NUL-terminated when shorter), so `d_out` may be `d_packed` and the buffer goes
 * straight into blurrily_storage_find_batch_device.  d_non_ascii[i] (optional)
 * is set to 1 for a needle holding a byte >= 0x80: its NFKD decomposition is host
 * work (the reference uses ActiveSupport's tables) and the caller must normalise
 * that needle itself.  Asynchronous on `stream`.  0, or -1 with errno. */
int blurrily_normalize_batch_device(const char* d_packed, const uint64_t* d_offsets, size_t n,
                                    char* d_out, uint32_t* d_non_ascii, void* stream);

/* blurrily_storage_find_batch over un-normalised ASCII needles: normalised on the
 * device, then found (a handful -- up to "few_max" / "mid_max" -- by the library on the host, byte
 * for byte the same, so that they share the single find's launch or its pinned page).  non_ascii (optional,
 * n slots) as above; rows of a flagged needle are those of its bytes >= 0x80 read as
 * non-letters. */
int blurrily_storage_find_batch_raw(trigram_map haystack, const char* packed, const uint64_t* offsets,
                                    size_t n, uint16_t limit, trigram_match results, uint32_t* counts,
                                    uint32_t* non_ascii);

/* Build / refresh the device-resident index now (it is otherwise built lazily
 * by the first find after a mutation).  0, or -1 with errno. */
int blurrily_storage_sync_device(trigram_map haystack);

/* Measure NOW which sweep serves batches of `n` needles at `limit` on this map's
 * image (what the first such batch would otherwise do inside its find call): runs
 * `n` needles sampled from the host-side needles given -- packed / offsets as for
 * blurrily_storage_find_batch, at least one needle, repeated as needed -- through
 * every sweep the class can take and notes the fastest.  Synchronous.  After it,
 * blurrily_storage_find_batch_device on that class never waits for a measurement.
 * 0, or -1 with errno. */
int blurrily_storage_tune(trigram_map haystack, const char* packed, const uint64_t* offsets,
                          size_t n_given, size_t n, uint16_t limit);

/* Tokeniser (ext/blurrily/tokeniser.h:34, tokeniser.c:59-119): `output` needs
 * strlen(input)+1 slots; returns the number of distinct codes, ascending. */
int blurrily_tokeniser_parse_string(const char* input, uint16_t* output);

/* Introspection for tests / bench (no reference counterpart). */
typedef struct blurrily_device_info_t {
  int32_t  device_ordinal;      /* -1 if no usable GPU                           */
  uint32_t n_refs;              /* distinct references in the device index       */
  uint32_t n_windows;           /* reference-rank windows                        */
  uint32_t window_bits;         /* log2(ranks per window)                        */
  uint64_t n_entries;           /* (trigram, ref) entries resident               */
  uint64_t device_bytes;        /* HBM bytes held by the index                   */
  double   last_find_kernel_ms; /* HIP-event time of the last find kernel launch */
  double   last_tokenise_kernel_ms;
  uint32_t n_pending;           /* puts since the base image was built (served by a delta image) */
  uint32_t n_tombstones;        /* base references deleted since the build                       */
  uint64_t base_builds;         /* full device-image builds so far                               */
  double   mean_hit_slice;      /* postings a needle trigram finds per window, on average: what the
                                   choice between the two sweeps is gated on ("ws_min_slice")     */
  uint32_t n_bitmaps;           /* dense slices that also exist as bitmaps (0: the window-major
                                   sweep cannot run on this image)                               */
  uint32_t reserved_;
  double   dense_share;         /* share of those postings in slices dense enough to have a bitmap  */
  double   ws_gain;             /* the four biggest buckets' part of mean_hit_slice x postings per
                                   reference: postings a needle can expect to leave out per window */
  /* option "devices" (round 5): what the replicas of the image actually sit on */
  uint32_t n_replicas;          /* copies of the image beside the primary's (0: "devices" 1, or none made yet)   */
  uint32_t distinct_devices;    /* PHYSICAL devices holding a copy, the primary's included (by PCI bus id): more
                                   replicas than devices share devices, and a batch is then no faster for them   */
  uint32_t peer_access_mask;    /* bit k: replica k's device and the primary's reach each other's memory directly
                                   (hipDeviceCanAccessPeer both ways, enabled): rows travel point to point over
                                   xGMI; bit clear: the peer copies of that replica are staged by the runtime     */
  uint32_t same_device_mask;    /* bit k: replica k sits on the primary's own device (no link involved)          */
  char     pci_bus_id[16];      /* the primary's device, "0000:c1:00.0"                                          */
} blurrily_device_info_t;
int blurrily_storage_device_info(trigram_map haystack, blurrily_device_info_t* info);
/* The same for a caller compiled against another version of this header: at most
 * `info_size` bytes are written (pass sizeof(blurrily_device_info_t) as the caller
 * knows it); returns the size of the structure as the LIBRARY knows it, so that a
 * caller can tell which trailing fields it got.  The structure only ever grows at
 * its end. */
size_t blurrily_storage_device_info_sized(trigram_map haystack, void* info, size_t info_size);

/* When non-zero, find_batch_device brackets its kernels with hipEvents on the
 * launch stream and synchronises to fill last_*_kernel_ms (bench/profiling). */
void blurrily_storage_set_timing(trigram_map haystack, int enabled);

/* Request counters of the find kernels (bench/profiling; no reference counterpart).  While
 * enabled, every find launch sequence counts -- exactly, from wave-uniform values -- what it
 * asks of the memory system and of the LDS; blurrily_storage_find_stats synchronises the
 * device and copies the counters of the LAST find call into out8[8]:
 *   [0] 16-bit postings loaded (x 2 = bytes; each is also one LDS-atomic lane)
 *   [1] sweep steps   [2] slice-table words loaded (x 4 = bytes)   [3] needles (or ranges) swept
 *   [4] candidate-pool compactions   [5] windows swept again after a pool overflow
 *   [6] wave-loads of postings issued by the window-major sweep   [7] bitmap words read for its candidates
 * Collecting costs a few scalar instructions per wave-load; leave it off when timing. */
void blurrily_storage_set_stats(trigram_map haystack, int enabled);
int  blurrily_storage_find_stats(trigram_map haystack, uint64_t* out8);
/* While the counters are on, every needle of a find call also gets a word saying which paths of the
 * kernels its find went through (4-bit / byte / 16-bit counters, cold-start bisection, pool overflow
 * and re-sweep, windows stepped over, latency-mode ranges, the window-major sweep's left-out slices,
 * robust scan, overflows ...: the kPath* bits of csrc/find_kernels.h, mirrored in blurrily_amd/map.py).
 * Copies the words of the first n needles of the LAST such call.  The parity tests use it to compare
 * needles of every class row for row.  0, or -1 with errno EINVAL (no such call, n too large). */
int  blurrily_storage_find_path_flags(trigram_map haystack, uint32_t* out, size_t n);
/* The find kernels the LAST batched find on this map launched for its needles, in launch order, distinct names joined by
 * '+' (e.g. "find_kernel<uint8_t,1024,false,true,true>", "find_small_kernel+find_kernel<uint8_t,1024,false,true,false>",
 * "find_kernel<uint8_t,1024,false,true,false>+wsweep_kernel"): what a measured sweep choice actually ran, for a bench
 * line or a profile to name.  NUL-terminated into out[cap] (truncated to cap - 1); returns the untruncated length. */
size_t blurrily_storage_last_kernels(trigram_map haystack, char* out, size_t cap);

/* Tunables (no reference counterpart; nothing on the find path reads the environment).
 * Per map -- read by the map's next find; calls on one map are serial, as in the reference:
 *   "wsweep"          1 (default) / 0: whether the window-major sweep may be taken at all
 *   "ws_min_windows"  (8)     fewest windows of an image it is taken on
 *   "ws_min_slice"    (1550)  least mean postings a needle trigram finds per window for the sweep to be possible on
 *                             an image at all (such an image carries bitmaps of its dense slices)
 *   "ws_autotune"     (1)     which sweep serves a class of batches (limit up to / above 32; 129.. / 16 384.. / 65 536.. /
 *                             262 144.. needles) is MEASURED: the first such batch on an image runs every sweep it can
 *                             take -- needle-major, window-major, needle-major with dense slices left out of the count
 *                             -- TWICE, the better run counting (same rows; that one call waits for them, see
 *                             blurrily_storage_tune) and the fastest serves the class until the image is rebuilt or an
 *                             option changes; leaving slices out has to win by 3 %, the window-major sweep by 5 %.  The
 *                             choice is watched: two batches of the class in a row that run over 10 % slower per needle
 *                             than the measurement saw have the class measured again, at most once in sixteen batches
 *                             ("retunes", get: how often that happened).  0: the static rules below
 *   "ws_static_slice" (2200)  the static rule: window-major iff mean postings per window >= this, x1.7 for batches
 *                             under 65 536 needles, x1.7 for limits above 32, x4 for both (measured table)
 *   "ws_choice"       get: what has been measured (class c in bits 2c+1:2c: 0 not yet, 1 needle-major,
 *                             2 window-major, 3 needle-major with slices left out; classes 0..2: limits up to 32 by
 *                             batch size 16 384.. / 65 536.. / 262 144.., 3..5: the same for limits above 32, 6 / 7:
 *                             batches of 129 .. 16 383 needles, limits up to / above 32); set 0: forget it
 *   "tuned_class", "tuned_nm_us", "tuned_ws_us", "tuned_leave_us"   get: the class measured most recently (-1: none)
 *                             and what its three sweeps took, in microseconds (0: that sweep could not run)
 *   "last_sweep"      get: which sweep the last large batch took (1 / 2 / 3 as above, 4: the small-haystack sweep; 0: latency mode)
 *   "nm_cmin"         (3)     the needle-major sweep may leave the largest dense slices of a (needle, window) out of
 *                             the count -- at most need - nm_cmin of them, eight at most -- and settle the candidates
 *                             that leaves pending through the slices' bitmaps; 0: never.  Limits up to 149 (the candidate pool's tail
 *                             has to hold the settled candidates beside what a glance at the pool lets pass)
 *   "nm_dense"        (3072)  ... slices of at least this many postings only (not below "dense_min")
 *   "nm_min_windows"  (256)   ... and, where the choice is not measured, on images of at least this many windows
 *   "small_sweep"     (1)     an image of at most eight windows serves batches of at least "small_min_needles" (4096)
 *                             needles at limits up to 64 with four waves and one window's counters per needle -- four
 *                             needles per CU at a time instead of two ("last_sweep" 4); 0: never
 *   "devices"         (1)     replicate the device image on the first n visible devices (replica k on device (primary
 *                             + k) mod visible) and shard every batch of at least 1 024 x n needles contiguously over
 *                             them: blurrily_storage_find_batch and _find_batch_device alike -- the rows land in the
 *                             caller's buffers, the answer does not depend on n.  Replicas follow puts and deletes
 *   "ws_min_needles"  (16384) smallest batch it is taken for
 *   "ws_cmin"         (3)     counted matches a left-out slice must leave
 *   "dense_min"       (1024)  postings from which a (window, trigram) slice also exists as a bitmap; changing
 *                             it rebuilds the device image at the next find.  Bitmaps are part of the postings
 *                             array (2^32 slots at most): a value that would overflow it is doubled for that build
 *                             until the image fits (a line on stderr says so)
 *   "one_launch"      (1)     blurrily_storage_find as ONE launch without copies where it can be (see there); 0: always the
 *                             batch's way.  "one_taken" (get): finds served that way so far.  "one_windows_per_wg" (0): at
 *                             least this many windows per workgroup of that launch (0: as few as 256 workgroups allow)
 *   "few_max"         (24)    blurrily_storage_find_batch / _raw: batches of up to this many needles (128 at most) share
 *                             the single find's launch, a row of the grid per needle; "mid_workgroups" (1024): the
 *                             workgroups such a launch aims at from nine needles on (a workgroup then takes several
 *                             window pairs)
 *   "mid_max"         (128)   ... and batches of more than "few_max" and up to this many needles (128 at most; 0: none)
 *                             are searched in latency mode without copies: tokenised on the host, the needle arrays
 *                             read from a pinned page, the merged rows written back into it and polled there (two
 *                             launches; larger batches, limits above 120 and needles of more than 64 distinct
 *                             trigrams take the staged copies, the device's tokeniser and a stream synchronise)
 *   "latency_tasks"   (0)     latency mode (batches too small to give every resident workgroup a needle): the tasks a
 *                             needle's windows are cut into, aimed at per resident workgroup; 0: one up to 60 needles,
 *                             two beyond (measured at Geonames scale)
 *   "host_chunk"      (131072) blurrily_storage_find_batch / _raw: a batch of at least twice as many needles goes
 *                             in chunks of this many through a three-stream pipeline (needles in, search, rows
 *                             out overlap); 0 = always one piece
 * Process-wide -- `haystack` NULL:
 *   "host_threads"    (0 = hardware threads, at most 64) threads of put_many and of the device-image build
 *   "build_trace"     (0) wall time of the build stages on stderr
 * set: 0, or -1 with errno EINVAL (unknown key, value out of range).  get: the value in effect. */
int blurrily_storage_set_option(trigram_map haystack, const char* key, long long value);
int blurrily_storage_get_option(trigram_map haystack, const char* key, long long* value);

#ifdef __cplusplus
}
#endif
#endif /* BLURRILY_AMD_STORAGE_H */
