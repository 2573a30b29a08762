// host_index.h -- the mutable, host-resident trigram index.
//
// This is the write side of the map: put / delete / stats / save / load.  It
// feeds the device-resident index (device_index.h) that `find` scans.  Its
// observable behaviour follows the reference's ext/blurrily/storage.c:
//   put     storage.c:398-473   (dup refs rejected, weight 0 -> strlen,
//                                bucket capacity 512 then x4/3, 0xAA fill)
//   delete  storage.c:584-612   (swap-with-last, 0xFF scribble)
//   save    storage.c:299-377   (sort dirty buckets, packed 548 832-byte
//                                header, 0xFF padding, page-aligned blocks,
//                                temp file + rename)
//   load    storage.c:210-266   (EPROTO on short file / bad magic / endianness
//                                / pointer size)
// so that files are byte-identical to the reference's for the same operations.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "tokeniser.h"

namespace blurrily {

struct Entry {            // storage.c:36-40 (packed {u32 reference; u32 weight})
  uint32_t ref;
  uint32_t weight;
};
static_assert(sizeof(Entry) == 8, "entry is 8 bytes on disk and in HBM");

// One trigram's postings.  `slots` is the reference's `buckets` (capacity).
// The raw storage is kept byte-for-byte like the reference's array (unused
// slots 0xAA, deleted tail slots 0xFF) because save() dumps all `slots`.
struct Bucket {
  uint32_t slots = 0;
  uint32_t used  = 0;
  Entry*   e     = nullptr;
  uint8_t  dirty = 0;     // appended since last sort (storage.c:55,464)
};

// Native set of live references (replaces search_tree.c's Ruby Hash).
class RefSet {
 public:
  ~RefSet();
  bool test(uint32_t ref) const;
  // Room for one more reference: grows the table when it is half full.  false = growth failed (out of
  // memory) AND the table is 7/8 full -- one more add could leave no empty slot for test() to stop at.
  bool ensure_room();
  void add(uint32_t ref);      // caller guarantees !test(ref) and a successful ensure_room()
  void remove(uint32_t ref);
  void clear();
  void reserve(uint64_t n);
  uint64_t size() const { return live_; }
 private:
  bool rehash(uint64_t want);
  uint32_t* key_ = nullptr;
  uint8_t*  tag_ = nullptr;    // 0 empty, 1 full, 2 tombstone
  uint64_t  cap_ = 0, live_ = 0, filled_ = 0;
};

// Process-wide options (blurrily_storage_set_option with a NULL map).
//   host_threads  threads for bulk work (put_many, the device-image transform); 0 = the hardware threads,
//                 at most 64 (several ranks on one host share its cores: bench.py sets cores / world)
//   build_trace   wall time of every stage of put_many / the device-image build on stderr
void     set_host_threads(unsigned n);
unsigned host_threads();
void     set_build_trace(bool on);
bool     build_trace();

class HostIndex {
 public:
  HostIndex();
  ~HostIndex();
  HostIndex(const HostIndex&) = delete;
  HostIndex& operator=(const HostIndex&) = delete;

  int  put(const char* needle, size_t len, uint32_t ref, uint32_t weight);
  // n puts in string order with the result (buckets byte for byte, totals) of n calls of put();
  // tokenising and filling the buckets run on all host cores.  Strings are the C strings at
  // packed + offsets[i] (at most offsets[i+1] - offsets[i] bytes); weights may be null (all 0).
  // Returns the trigrams added.
  long put_many(const char* packed, const uint64_t* offsets, const uint32_t* refs, const uint32_t* weights,
                size_t n);
  int  del(uint32_t ref);
  int  save(const char* path);                 // 0 / -1+errno
  static HostIndex* load(const char* path);    // nullptr+errno on failure

  uint32_t total_refs() const     { return total_refs_; }
  uint32_t total_trigrams() const { return total_trigrams_; }

  // Sort every dirty bucket by reference (storage.c:142-150, :310-312).
  void sort_dirty_buckets();
  // What the reference's find does for each of the needle's trigrams
  // (storage.c:516): sorting happens in place and clears the dirty flag.
  void sort_bucket_if_dirty(uint32_t code);
  uint32_t dirty_buckets() const;

  // Monotone counter bumped by every mutation of postings; the device index
  // remembers the value it was built from.
  uint64_t generation() const { return generation_; }

  const Bucket& bucket(uint32_t code) const { return b_[code]; }

 private:
  void ensure_refset();                        // storage.c:404-407, :381-394
  Bucket*  b_;                                 // kNumCodes buckets
  uint32_t total_refs_ = 0, total_trigrams_ = 0;
  bool     refset_built_ = false;
  RefSet   refs_;
  uint64_t generation_ = 1;
};

}  // namespace blurrily
