// host_index.cpp -- see host_index.h.
#include "host_index.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cerrno>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace blurrily {

namespace {
std::atomic<unsigned> g_host_threads{0};
std::atomic<bool>     g_build_trace{false};
}  // namespace

void set_host_threads(unsigned n) { g_host_threads.store(std::min(n, 256u)); }
unsigned host_threads() {
  const unsigned n = g_host_threads.load();
  return n ? n : std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
}
void set_build_trace(bool on) { g_build_trace.store(on); }
bool build_trace() { return g_build_trace.load(); }

namespace {

constexpr size_t   kPage          = 4096;               // storage.c:29
constexpr uint32_t kFirstSlots    = kPage / sizeof(Entry);  // storage.c:31 (512)
constexpr size_t   kDescBytes     = 25;                 // packed trigram_entries_t, storage.c:47-56
constexpr size_t   kHeaderFixed   = 32;                 // magic..refs, storage.c:64-71
constexpr size_t   kHeaderBytes   = kHeaderFixed + kDescBytes * kNumCodes;  // 548 832
constexpr uint8_t  kFillFresh     = 0xAA;               // storage.c:93-98
constexpr uint8_t  kFillDead      = 0xFF;               // storage.c:338,598

inline size_t round_to_page(size_t v) { return (v + kPage - 1) / kPage * kPage; }

inline uint8_t endian_byte() {                          // storage.c:103-109: 1 = little endian
  const uint32_t probe = 0xAA0000BBu;
  return (*reinterpret_cast<const uint8_t*>(&probe) == 0xBB) ? 1 : 2;
}

Entry* alloc_slots(uint32_t slots) {
  Entry* p = static_cast<Entry*>(std::malloc(size_t(slots) * sizeof(Entry)));
  if (p) std::memset(p, kFillFresh, size_t(slots) * sizeof(Entry));
  return p;
}

inline uint64_t mix(uint32_t x) {
  uint64_t z = (uint64_t(x) + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
  return z ^ (z >> 31);
}

template <typename T> void put_le(uint8_t* p, T v) { std::memcpy(p, &v, sizeof(T)); }
template <typename T> T    get_le(const uint8_t* p) { T v; std::memcpy(&v, p, sizeof(T)); return v; }

}  // namespace

// ------------------------------------------------------------------ RefSet ---

RefSet::~RefSet() { clear(); }

void RefSet::clear() {
  std::free(key_); std::free(tag_);
  key_ = nullptr; tag_ = nullptr; cap_ = live_ = filled_ = 0;
}

bool RefSet::rehash(uint64_t want) {
  uint64_t cap = 1024;
  while (cap < want * 2) cap <<= 1;
  uint32_t* nk = static_cast<uint32_t*>(std::malloc(cap * sizeof(uint32_t)));
  uint8_t*  nt = static_cast<uint8_t*>(std::calloc(cap, 1));
  if (!nk || !nt) {                                               // out of memory: the table stays as it is;
    std::free(nk); std::free(nt);                                 // ensure_room() tells the caller when it is too full
    return false;
  }
  uint64_t live = 0;
  for (uint64_t i = 0; i < cap_; ++i) {
    if (tag_[i] != 1) continue;
    uint64_t h = mix(key_[i]) & (cap - 1);
    while (nt[h]) h = (h + 1) & (cap - 1);
    nt[h] = 1; nk[h] = key_[i]; ++live;
  }
  std::free(key_); std::free(tag_);
  key_ = nk; tag_ = nt; cap_ = cap; live_ = filled_ = live;
  return true;
}

void RefSet::reserve(uint64_t n) { if (n * 2 > cap_) (void)rehash(n); }

bool RefSet::ensure_room() {
  if ((filled_ + 1) * 2 <= cap_) return true;
  if (rehash(std::max<uint64_t>(live_ + 1, 512))) return true;   // (also drops the tombstones of removed references)
  return cap_ != 0 && (filled_ + 1) * 8 <= cap_ * 7;
}

bool RefSet::test(uint32_t ref) const {
  if (!cap_) return false;
  uint64_t h = mix(ref) & (cap_ - 1);
  while (tag_[h]) {
    if (tag_[h] == 1 && key_[h] == ref) return true;
    h = (h + 1) & (cap_ - 1);
  }
  return false;
}

void RefSet::add(uint32_t ref) {
  uint64_t h = mix(ref) & (cap_ - 1);
  while (tag_[h] == 1) h = (h + 1) & (cap_ - 1);
  if (tag_[h] == 0) ++filled_;
  tag_[h] = 1; key_[h] = ref; ++live_;
}

void RefSet::remove(uint32_t ref) {
  if (!cap_) return;
  uint64_t h = mix(ref) & (cap_ - 1);
  while (tag_[h]) {
    if (tag_[h] == 1 && key_[h] == ref) { tag_[h] = 2; --live_; return; }
    h = (h + 1) & (cap_ - 1);
  }
}

// --------------------------------------------------------------- HostIndex ---

HostIndex::HostIndex() { b_ = new Bucket[kNumCodes]; }

HostIndex::~HostIndex() {
  for (uint32_t t = 0; t < kNumCodes; ++t) std::free(b_[t].e);
  delete[] b_;
}

// The reference builds its ref set lazily on the first put (storage.c:404-407)
// by walking every entry (:381-394).  Every string contributes exactly one
// "leading" trigram "**c" (frame position 0, code 784*sym(c)), and a reference
// occurs at most once per bucket, so the 27 leading buckets enumerate each
// live reference once; we fall back to the full walk if the totals disagree
// (possible only for files not produced by put).
void HostIndex::ensure_refset() {
  if (refset_built_) return;
  refs_.clear();
  uint64_t lead = 0;
  for (uint32_t s = 0; s < uint32_t(kBase) - 1; ++s) lead += b_[s * kBase * kBase].used;
  refs_.reserve(std::max<uint64_t>(lead, total_refs_));
  auto add_bucket = [&](const Bucket& bk) {
    for (uint32_t j = 0; j < bk.used; ++j)
      if (!refs_.test(bk.e[j].ref)) {
        if (!refs_.ensure_room()) { std::fprintf(stderr, "blurrily_hip: out of memory (reference set)\n"); std::abort(); }
        refs_.add(bk.e[j].ref);
      }
  };
  if (lead == total_refs_) {
    for (uint32_t s = 0; s < uint32_t(kBase) - 1; ++s) add_bucket(b_[s * kBase * kBase]);
  }
  if (refs_.size() != total_refs_ || lead != total_refs_) {
    for (uint32_t t = 0; t < kNumCodes; ++t) add_bucket(b_[t]);
  }
  refset_built_ = true;
}

int HostIndex::put(const char* needle, size_t len, uint32_t ref, uint32_t weight) {
  ensure_refset();
  if (refs_.test(ref)) return 0;                                  // storage.c:408
  if (!refs_.ensure_room()) { errno = ENOMEM; return -1; }        // (before anything is touched)
  if (weight == 0) weight = uint32_t(len);                        // storage.c:409

  uint16_t  small[256];
  uint16_t* codes = (len + 1 <= 256) ? small
                                     : static_cast<uint16_t*>(std::malloc((len + 1) * sizeof(uint16_t)));
  if (!codes) { errno = ENOMEM; return -1; }
  const int n = tokenise(needle, len, codes);                     // storage.c:412

  // Room first, entries after: an allocation that fails adds no entry (the reference's smalloc asserts
  // instead, storage.c:93-98; buckets grown before the failure keep their new capacity).  A bucket grows exactly when the reference's would --
  // when it is full and about to take an entry -- so files stay byte-identical.
  for (int k = 0; k < n; ++k) {                                   // storage.c:424-458
    Bucket& bk = b_[codes[k]];
    if (bk.slots == 0) {                                          // :424-429
      Entry* ne = alloc_slots(kFirstSlots);
      if (!ne) { if (codes != small) std::free(codes); errno = ENOMEM; return -1; }
      bk.e = ne; bk.slots = kFirstSlots;
    } else if (bk.used == bk.slots) {                             // :430-458
      uint32_t grown = uint32_t(uint64_t(bk.slots) * 4 / 3);
      if (grown <= bk.slots) grown = bk.slots + 1;                // only for hand-made files with < 3 slots
      Entry* ne = alloc_slots(grown);
      if (!ne) { if (codes != small) std::free(codes); errno = ENOMEM; return -1; }
      std::memcpy(ne, bk.e, size_t(bk.slots) * sizeof(Entry));
      std::free(bk.e);
      bk.e = ne; bk.slots = grown;
    }
  }
  for (int k = 0; k < n; ++k) {                                   // storage.c:462-465
    Bucket& bk = b_[codes[k]];
    bk.e[bk.used].ref = ref;                                      // :462-464
    bk.e[bk.used].weight = weight;
    bk.used += 1;
    bk.dirty = 1;
  }
  total_trigrams_ += uint32_t(n);                                 // :466-467
  total_refs_ += 1;
  refs_.add(ref);                                                 // :469
  ++generation_;
  if (codes != small) std::free(codes);
  return n;
}

long HostIndex::put_many(const char* packed, const uint64_t* offsets, const uint32_t* refs,
                         const uint32_t* weights, size_t n) {
  if (n == 0) return 0;
  const bool trace = build_trace();
  auto t_last = std::chrono::steady_clock::now();
  auto stage = [&](const char* what) {
    if (!trace) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "blurrily_hip: put_many: %-28s %8.1f ms\n", what,
                 std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  ensure_refset();
  // 1. which strings are stored: a reference already known -- or seen earlier in this batch -- is
  //    skipped (storage.c:408); string order decides, so this pass is sequential
  std::vector<uint8_t> take(n);
  size_t accepted = 0;
  for (size_t i = 0; i < n; ++i) {
    take[i] = !refs_.test(refs[i]);
    if (!take[i]) continue;
    if (!refs_.ensure_room()) {                                   // out of memory: nothing of this batch is stored
      for (size_t j = 0; j < i; ++j) if (take[j]) refs_.remove(refs[j]);
      errno = ENOMEM;
      return -1;
    }
    refs_.add(refs[i]); ++accepted;
  }
  if (!accepted) return 0;
  stage("reference set");

  // 2. tokenise in parallel, chunk by chunk: codes of every stored string, and how many entries
  //    each chunk adds to each bucket
  const size_t n_chunks = std::max<size_t>(1, std::min<size_t>(size_t(host_threads()), n / 4096 + 1));
  const size_t per = (n + n_chunks - 1) / n_chunks;
  struct Chunk { std::vector<uint16_t> codes; std::vector<uint32_t> ncodes, lens; std::vector<uint32_t> hist; long added = 0; };
  std::vector<Chunk> chunks(n_chunks);
  auto for_chunks = [&](auto&& body) {
    std::vector<std::thread> th;
    for (size_t c = 1; c < n_chunks; ++c) th.emplace_back([&, c] { body(c); });
    body(0);
    for (auto& t : th) t.join();
  };
  for_chunks([&](size_t c) {
    const size_t lo = c * per, hi = std::min(n, lo + per);
    const size_t cnt = hi > lo ? hi - lo : 0;
    // (locals, handed over at the end: the chunks sit next to each other in memory)
    std::vector<uint16_t> codes;
    std::vector<uint32_t> ncodes(cnt, 0), lens(cnt, 0), hist(kNumCodes, 0);
    codes.reserve(cnt * 16);
    long added = 0;
    uint16_t small[256];
    std::vector<uint16_t> big;
    for (size_t i = lo; i < hi; ++i) {
      if (!take[i]) continue;
      const char* s = packed + offsets[i];
      const size_t cap = size_t(offsets[i + 1] - offsets[i]);
      const void* nul = std::memchr(s, 0, cap);
      const size_t len = nul ? size_t(static_cast<const char*>(nul) - s) : cap;
      uint16_t* tmp = small;
      if (len + 1 > 256) { big.resize(len + 1); tmp = big.data(); }
      const int k = tokenise(s, len, tmp);
      codes.insert(codes.end(), tmp, tmp + k);
      ncodes[i - lo] = uint32_t(k);
      lens[i - lo] = uint32_t(len);
      for (int j = 0; j < k; ++j) ++hist[tmp[j]];
      added += k;
    }
    Chunk& ck = chunks[c];
    ck.codes.swap(codes); ck.ncodes.swap(ncodes); ck.lens.swap(lens); ck.hist.swap(hist); ck.added = added;
  });
  stage("tokenise");
  // 3. per bucket: where each chunk writes (chunks in string order: the order of n puts), and the
  //    array the reference's growth rule ends with (512 slots, then x 4/3 whenever full; a grown
  //    array is the old one copied whole, the rest 0xAA -- storage.c:424-458)
  std::vector<uint32_t> start(size_t(n_chunks) * kNumCodes);
  {
    std::vector<std::thread> th;
    const uint32_t n_thr = uint32_t(std::max<size_t>(1, std::min<size_t>(n_chunks, 32)));
    auto grow = [&](uint32_t t0, uint32_t t1) {
      for (uint32_t t = t0; t < t1; ++t) {
        Bucket& bk = b_[t];
        uint32_t at = bk.used;
        for (size_t c = 0; c < n_chunks; ++c) { start[c * kNumCodes + t] = at; at += chunks[c].hist[t]; }
        if (at == bk.used) continue;
        uint32_t slots = bk.slots ? bk.slots : kFirstSlots;
        while (slots < at) {
          uint32_t grown = uint32_t(uint64_t(slots) * 4 / 3);
          if (grown <= slots) grown = slots + 1;
          slots = grown;
        }
        if (slots != bk.slots) {
          Entry* ne = alloc_slots(slots);
          if (!ne) {                                              // mid-way through a bulk import on all cores:
            std::fprintf(stderr, "blurrily_hip: out of memory growing a bucket\n");   // nothing sane to return to
            std::abort();
          }
          if (bk.slots) std::memcpy(ne, bk.e, size_t(bk.slots) * sizeof(Entry));
          std::free(bk.e);
          bk.e = ne; bk.slots = slots;
        }
        bk.used = at;
        bk.dirty = 1;
      }
    };
    const uint32_t span = (kNumCodes + n_thr - 1) / n_thr;
    for (uint32_t k = 1; k < n_thr; ++k) th.emplace_back(grow, k * span, std::min(kNumCodes, (k + 1) * span));
    grow(0, std::min(kNumCodes, span));
    for (auto& t : th) t.join();
  }

  stage("grow buckets");
  // 4. fill: every chunk writes its own slots
  for_chunks([&](size_t c) {
    const Chunk& ck = chunks[c];
    const size_t lo = c * per, hi = std::min(n, lo + per);
    uint32_t* at = &start[c * kNumCodes];
    size_t pos = 0;
    for (size_t i = lo; i < hi; ++i) {
      if (!take[i]) continue;
      const uint32_t w0 = weights ? weights[i] : 0u;
      const Entry e{refs[i], w0 ? w0 : ck.lens[i - lo]};              // storage.c:409
      for (uint32_t j = 0; j < ck.ncodes[i - lo]; ++j) b_[ck.codes[pos + j]].e[at[ck.codes[pos + j]]++] = e;
      pos += ck.ncodes[i - lo];
    }
  });

  stage("fill");
  long added = 0;
  for (const Chunk& ck : chunks) added += ck.added;
  total_trigrams_ += uint32_t(added);                                 // storage.c:466-467
  total_refs_ += uint32_t(accepted);
  ++generation_;
  return added;
}

int HostIndex::del(uint32_t ref) {                                // storage.c:584-612
  int removed = 0;
  for (uint32_t t = 0; t < kNumCodes; ++t) {
    Bucket& bk = b_[t];
    for (uint32_t j = 0; j < bk.used; ++j) {
      if (bk.e[j].ref != ref) continue;
      bk.e[j] = bk.e[bk.used - 1];                                // swap with last (:597)
      std::memset(&bk.e[bk.used - 1], kFillDead, sizeof(Entry));  // :598
      bk.used -= 1;
      ++removed;
      --j;                                                        // re-examine the moved entry (:603)
    }
  }
  total_trigrams_ -= uint32_t(removed);                           // :606-607
  if (removed > 0) { total_refs_ -= 1; ++generation_; }
  if (refset_built_) refs_.remove(ref);                           // :609
  return removed;
}

void HostIndex::sort_dirty_buckets() {
  for (uint32_t t = 0; t < kNumCodes; ++t) {
    Bucket& bk = b_[t];
    if (!bk.dirty) continue;                                      // storage.c:145
    std::sort(bk.e, bk.e + bk.used,
              [](const Entry& l, const Entry& r) { return l.ref < r.ref; });
    bk.dirty = 0;
  }
}

void HostIndex::sort_bucket_if_dirty(uint32_t code) {
  Bucket& bk = b_[code];
  if (!bk.dirty) return;
  std::sort(bk.e, bk.e + bk.used, [](const Entry& l, const Entry& r) { return l.ref < r.ref; });
  bk.dirty = 0;
}

uint32_t HostIndex::dirty_buckets() const {
  uint32_t n = 0;
  for (uint32_t t = 0; t < kNumCodes; ++t) n += b_[t].dirty ? 1u : 0u;
  return n;
}

// -------------------------------------------------------------------- save ---

namespace {
struct Writer {
  int fd; std::vector<uint8_t> buf; size_t fill = 0; bool ok = true;
  explicit Writer(int f) : fd(f), buf(size_t(1) << 22) {}
  void flush() {
    size_t off = 0;
    while (ok && off < fill) {
      ssize_t w = ::write(fd, buf.data() + off, fill - off);
      if (w < 0) { if (errno == EINTR) continue; ok = false; break; }
      off += size_t(w);
    }
    fill = 0;
  }
  void bytes(const void* p, size_t n) {
    const uint8_t* s = static_cast<const uint8_t*>(p);
    while (n) {
      const size_t c = std::min(n, buf.size() - fill);
      std::memcpy(buf.data() + fill, s, c);
      fill += c; s += c; n -= c;
      if (fill == buf.size()) flush();
    }
  }
  void pad(uint8_t v, size_t n) {
    while (n) {
      const size_t c = std::min(n, buf.size() - fill);
      std::memset(buf.data() + fill, v, c);
      fill += c; n -= c;
      if (fill == buf.size()) flush();
    }
  }
};
}  // namespace

int HostIndex::save(const char* path) {
  sort_dirty_buckets();                                           // storage.c:310-312

  char tmp[PATH_MAX];
  std::snprintf(tmp, sizeof(tmp), "%s.tmp.%ld", path, random());  // storage.c:315
  const int fd = ::open(tmp, O_RDWR | O_CREAT | O_TRUNC, 0644);   // storage.c:325
  if (fd < 0) return -1;

  // header (storage.c:340-362): packed map with pointers cleared, per-bucket
  // offsets of the page-aligned blocks that follow.
  std::vector<uint8_t> hdr(kHeaderBytes, 0);
  std::memcpy(hdr.data(), "trigra", 6);
  hdr[6] = endian_byte();
  hdr[7] = uint8_t(sizeof(void*));
  put_le<uint32_t>(&hdr[8], total_refs_);
  put_le<uint32_t>(&hdr[12], total_trigrams_);
  // mapped_size (8 bytes) and refs (8 bytes) are written as zero (:345-346)
  uint64_t offset = round_to_page(kHeaderBytes);
  for (uint32_t t = 0; t < kNumCodes; ++t) {
    uint8_t* d = &hdr[kHeaderFixed + kDescBytes * t];
    const Bucket& bk = b_[t];
    put_le<uint32_t>(d + 0, bk.slots);
    put_le<uint32_t>(d + 4, bk.used);
    // entries pointer (8 bytes) stays NULL (:355,360)
    const uint64_t block = uint64_t(bk.slots) * sizeof(Entry);
    put_le<uint64_t>(d + 16, block ? offset : 0);                 // entries_offset (:356,361)
    d[24] = bk.dirty;
    offset += round_to_page(block);
  }

  Writer w(fd);
  w.bytes(hdr.data(), hdr.size());
  w.pad(kFillDead, round_to_page(kHeaderBytes) - kHeaderBytes);   // file pre-filled with 0xFF (:338)
  for (uint32_t t = 0; t < kNumCodes; ++t) {
    const Bucket& bk = b_[t];
    const size_t block = size_t(bk.slots) * sizeof(Entry);
    if (!block) continue;
    w.bytes(bk.e, block);                                         // whole capacity, fill bytes included (:353)
    w.pad(kFillDead, round_to_page(block) - block);
  }
  w.flush();
  int res = w.ok ? 0 : -1;
  const int saved = errno;
  if (::close(fd) < 0 && res == 0) res = -1;
  if (res < 0) { if (w.ok == false) errno = saved; ::unlink(tmp); return -1; }
  return ::rename(tmp, path);                                     // storage.c:371-374
}

// -------------------------------------------------------------------- load ---

HostIndex* HostIndex::load(const char* path) {
  const int fd = ::open(path, O_RDONLY);                          // storage.c:219
  if (fd < 0) return nullptr;
  struct stat st;
  if (::fstat(fd, &st) < 0) { const int e = errno; ::close(fd); errno = e; return nullptr; }
  if (st.st_size < off_t(kHeaderBytes)) {                         // storage.c:226-230
    ::close(fd); errno = EPROTO; return nullptr;
  }
  const size_t size = size_t(st.st_size);
  void* mem = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
  const int map_errno = errno;
  ::close(fd);
  if (mem == MAP_FAILED) { errno = map_errno; return nullptr; }
  const uint8_t* base = static_cast<const uint8_t*>(mem);

  auto fail = [&](int e) -> HostIndex* { ::munmap(mem, size); errno = e; return nullptr; };

  if (std::memcmp(base, "trigra", 6) != 0 || base[6] != endian_byte() ||
      base[7] != uint8_t(sizeof(void*)))                          // storage.c:245-250
    return fail(EPROTO);

  HostIndex* ix = new HostIndex();
  ix->total_refs_     = get_le<uint32_t>(base + 8);
  ix->total_trigrams_ = get_le<uint32_t>(base + 12);
  for (uint32_t t = 0; t < kNumCodes; ++t) {
    const uint8_t* d = base + kHeaderFixed + kDescBytes * t;
    const uint32_t slots = get_le<uint32_t>(d + 0);
    const uint32_t used  = get_le<uint32_t>(d + 4);
    const uint64_t off   = get_le<uint64_t>(d + 16);
    if (off == 0) {                                               // storage.c:257
      if (used != 0) { delete ix; return fail(EPROTO); }
      continue;                                                   // nothing mapped for this trigram
    }
    const uint64_t block = uint64_t(slots) * sizeof(Entry);
    if (used > slots || off > size || block > size - off) { delete ix; return fail(EPROTO); }
    Bucket& bk = ix->b_[t];
    bk.slots = slots; bk.used = used; bk.dirty = d[24];
    bk.e = static_cast<Entry*>(std::malloc(std::max<size_t>(block, 1)));
    if (!bk.e) { bk.slots = bk.used = 0; delete ix; ::munmap(mem, size); errno = ENOMEM; return nullptr; }
    std::memcpy(bk.e, base + off, block);
  }
  ::munmap(mem, size);
  return ix;
}

}  // namespace blurrily
