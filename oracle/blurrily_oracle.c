/*
 * blurrily_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the reference's trigram
 * put/find algorithm (mezis/blurrily v1.0.2).  It exists so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the
 * HIP path against the reference's *results*.  Nothing under blurrily_amd/
 * may include, link, dlopen or call this file.
 *
 * Parity status: PINNED.  tests/test_oracle_pinning.py checks this file
 *   (a) against every known-answer vector the reference's specs hold for the
 *       find path (tests/golden/spec_vectors.json, transcribed from
 *       spec/blurrily/map_spec.rb, command_processor_spec.rb,
 *       integration_spec.rb), and
 *   (b) against the reference's own C compiled in place
 *       (oracle/_ref/libblurrily_ref.so, see oracle/Makefile) on seeded random
 *       haystacks, and against fixtures that build emitted
 *       (tests/golden/ref_find_*.json).
 *
 * Each function cites the reference lines it restates
 * (paths relative to /root/reference/ext/blurrily/).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORA_BASE     28                /* tokeniser.h:22  TRIGRAM_BASE         */
#define ORA_NCODES   (28 * 28 * 28)    /* storage.c:30    TRIGRAM_COUNT = 21952 */

typedef struct { uint32_t reference, weight; } ora_entry_t;           /* storage.c:36-40 */
typedef struct __attribute__((packed)) {                               /* storage.h:18-22 */
  uint32_t reference, matches, weight;
} ora_match_t;

typedef struct {
  ora_entry_t* e;
  uint32_t     used, cap;
} ora_bucket_t;

typedef struct {
  ora_bucket_t b[ORA_NCODES];
  uint32_t     total_refs, total_trigrams;          /* storage.c:68-69 */
  /* open-addressing set of live refs (stands in for search_tree.c's Hash) */
  uint32_t*    set;        /* slot = ref+1 stored as u64 would waste; we keep refs and a tag array */
  uint8_t*     tag;        /* 0 empty, 1 full, 2 tombstone */
  uint64_t     set_cap, set_live, set_used;
} ora_map_t;

/* ------------------------------------------------------------------ tokeniser */

/* symbol value of one byte: 'a'..'z' -> 1..26, everything else (incl. '*',
 * space, bytes >= 0x80 which are negative chars in the reference) -> 0.
 * tokeniser.c:21-31 */
static inline unsigned ora_sym(unsigned char c) {
  return (c >= 'a' && c <= 'z') ? (unsigned)(c - 'a' + 1) : 0u;
}

static int ora_cmp_u16(const void* a, const void* b) {
  return (int)*(const uint16_t*)a - (int)*(const uint16_t*)b;   /* tokeniser.c:50-55 */
}

/* tokeniser.c:59-119: p = "**" + s + "*"; for k in 0..len: code(p[k..k+2]),
 * little-endian base 28 (first char is the least significant digit);
 * sort ascending, drop duplicates.  Returns the number of distinct codes
 * (always >= 1: the empty string gives the single code 0).
 * `out` needs strlen(input)+1 slots. */
int oracle_tokenise(const char* input, uint16_t* out) {
  size_t len = strlen(input);
  size_t n = len + 1;
  for (size_t k = 0; k < n; ++k) {
    /* padded string index k+j maps to input index k+j-2 */
    unsigned code = 0, mul = 1;
    for (int j = 0; j < 3; ++j, mul *= ORA_BASE) {
      size_t p = k + (size_t)j;                 /* position in "**s*" */
      unsigned s = 0;
      if (p >= 2 && p < len + 2) s = ora_sym((unsigned char)input[p - 2]);
      code += mul * s;
    }
    out[k] = (uint16_t)code;
  }
  qsort(out, n, sizeof(uint16_t), ora_cmp_u16);                 /* tokeniser.c:93 */
  size_t m = 0;
  for (size_t k = 0; k < n; ++k)                                /* :96-107 (dedup) */
    if (m == 0 || out[m - 1] != out[k]) out[m++] = out[k];
  return (int)m;
}

/* ------------------------------------------------------------------ ref set   */

static uint64_t ora_hash(uint32_t x) {
  uint64_t z = (uint64_t)x * 0x9E3779B97F4A7C15ull;
  return z ^ (z >> 29);
}

static void ora_set_grow(ora_map_t* m, uint64_t want) {
  uint64_t cap = 1024;
  while (cap < want * 2) cap <<= 1;
  uint32_t* os = m->set; uint8_t* ot = m->tag; uint64_t oc = m->set_cap;
  m->set = (uint32_t*)calloc(cap, sizeof(uint32_t));
  m->tag = (uint8_t*)calloc(cap, 1);
  m->set_cap = cap; m->set_used = 0; m->set_live = 0;
  for (uint64_t i = 0; i < oc; ++i) {
    if (ot[i] != 1) continue;
    uint64_t h = ora_hash(os[i]) & (cap - 1);
    while (m->tag[h]) h = (h + 1) & (cap - 1);
    m->tag[h] = 1; m->set[h] = os[i]; m->set_used++; m->set_live++;
  }
  free(os); free(ot);
}

static int ora_set_test(const ora_map_t* m, uint32_t ref) {
  if (!m->set_cap) return 0;
  uint64_t h = ora_hash(ref) & (m->set_cap - 1);
  while (m->tag[h]) {
    if (m->tag[h] == 1 && m->set[h] == ref) return 1;
    h = (h + 1) & (m->set_cap - 1);
  }
  return 0;
}

static void ora_set_add(ora_map_t* m, uint32_t ref) {
  if ((m->set_used + 1) * 2 > m->set_cap) ora_set_grow(m, m->set_live + 1);
  uint64_t h = ora_hash(ref) & (m->set_cap - 1);
  while (m->tag[h] == 1) h = (h + 1) & (m->set_cap - 1);
  if (m->tag[h] == 0) m->set_used++;
  m->tag[h] = 1; m->set[h] = ref; m->set_live++;
}

static void ora_set_remove(ora_map_t* m, uint32_t ref) {
  if (!m->set_cap) return;
  uint64_t h = ora_hash(ref) & (m->set_cap - 1);
  while (m->tag[h]) {
    if (m->tag[h] == 1 && m->set[h] == ref) { m->tag[h] = 2; m->set_live--; return; }
    h = (h + 1) & (m->set_cap - 1);
  }
}

/* ------------------------------------------------------------------ map       */

void* oracle_new(void) { return calloc(1, sizeof(ora_map_t)); }   /* storage.c:178-206 */

void oracle_free(void* h) {                                        /* storage.c:270-295 */
  ora_map_t* m = (ora_map_t*)h;
  if (!m) return;
  for (int t = 0; t < ORA_NCODES; ++t) free(m->b[t].e);
  free(m->set); free(m->tag); free(m);
}

/* storage.c:398-473.  Returns number of trigrams added, 0 for a duplicate ref.
 * weight 0 -> strlen(needle) (:409). */
int oracle_put(void* h, const char* needle, uint32_t ref, uint32_t weight) {
  ora_map_t* m = (ora_map_t*)h;
  if (ora_set_test(m, ref)) return 0;                              /* :408 */
  size_t len = strlen(needle);
  if (weight == 0) weight = (uint32_t)len;                         /* :409 */
  uint16_t* codes = (uint16_t*)malloc((len + 1) * sizeof(uint16_t));
  int n = oracle_tokenise(needle, codes);                          /* :412 */
  for (int k = 0; k < n; ++k) {                                    /* :415-465 */
    ora_bucket_t* b = &m->b[codes[k]];
    if (b->used == b->cap) {
      b->cap = b->cap ? b->cap * 2 : 16;
      b->e = (ora_entry_t*)realloc(b->e, (size_t)b->cap * sizeof(ora_entry_t));
    }
    b->e[b->used].reference = ref;
    b->e[b->used].weight = weight;
    b->used++;
  }
  m->total_trigrams += (uint32_t)n;                                /* :466-467 */
  m->total_refs += 1;
  ora_set_add(m, ref);                                             /* :469 */
  free(codes);
  return n;
}

/* n puts in one call (refs 1..n when `refs` is NULL); needles are
 * packed[offsets[i] .. offsets[i+1]).  Test/bench convenience only. */
long oracle_put_many(void* h, const char* packed, const uint64_t* offsets, const uint32_t* refs, size_t n) {
  long total = 0;
  size_t cap = 256;
  char* tmp = (char*)malloc(cap);
  for (size_t i = 0; i < n; ++i) {
    size_t len = (size_t)(offsets[i + 1] - offsets[i]);
    if (len + 1 > cap) { cap = (len + 1) * 2; tmp = (char*)realloc(tmp, cap); }
    memcpy(tmp, packed + offsets[i], len);
    tmp[len] = 0;
    total += oracle_put(h, tmp, refs ? refs[i] : (uint32_t)(i + 1), 0);
  }
  free(tmp);
  return total;
}

/* storage.c:584-612: drop every entry of `ref` (swap with last). */
int oracle_delete(void* h, uint32_t ref) {
  ora_map_t* m = (ora_map_t*)h;
  int removed = 0;
  for (int t = 0; t < ORA_NCODES; ++t) {
    ora_bucket_t* b = &m->b[t];
    for (uint32_t j = 0; j < b->used; ) {
      if (b->e[j].reference == ref) { b->e[j] = b->e[--b->used]; ++removed; }
      else ++j;
    }
  }
  m->total_trigrams -= (uint32_t)removed;                          /* :606-607 */
  if (removed > 0) m->total_refs -= 1;
  ora_set_remove(m, ref);                                          /* :609 */
  return removed;
}

int oracle_stats(void* h, uint32_t* refs, uint32_t* trigrams) {   /* storage.c:616-621 */
  ora_map_t* m = (ora_map_t*)h;
  *refs = m->total_refs; *trigrams = m->total_trigrams;
  return 0;
}

/* Σ_t used[t] over the needle's trigrams: the reference's nb_entries
 * (storage.c:498-502) -- the unit of the "matched trigram-entries/s" metric. */
uint64_t oracle_nb_entries(void* h, const char* needle) {
  ora_map_t* m = (ora_map_t*)h;
  size_t len = strlen(needle);
  uint16_t* codes = (uint16_t*)malloc((len + 1) * sizeof(uint16_t));
  int n = oracle_tokenise(needle, codes);
  uint64_t s = 0;
  for (int k = 0; k < n; ++k) s += m->b[codes[k]].used;
  free(codes);
  return s;
}

/* The reference compares with signed int differences (storage.c:121-138);
 * inside its validated ranges (refs, weights < 2^31, lib/blurrily/defaults.rb:7-9)
 * that equals plain integer order.  The reference obtains ref-ascending order
 * among (matches, weight) ties only because glibc's qsort is a stable merge
 * sort running over ref-sorted input (storage.c:523,566; pinned by
 * spec/integration_spec.rb:37-42); here the third key is explicit so the
 * result does not depend on libc. */
static int ora_cmp_entry(const void* a, const void* b) {           /* storage.c:121-126 */
  uint32_t x = ((const ora_entry_t*)a)->reference, y = ((const ora_entry_t*)b)->reference;
  return (x > y) - (x < y);
}

static int ora_cmp_match(const void* a, const void* b) {           /* storage.c:129-138 + tie */
  const ora_match_t* l = (const ora_match_t*)a; const ora_match_t* r = (const ora_match_t*)b;
  if (l->matches != r->matches) return (l->matches < r->matches) ? 1 : -1;   /* desc */
  if (l->weight  != r->weight)  return (l->weight  > r->weight)  ? 1 : -1;   /* asc  */
  return (l->reference > r->reference) - (l->reference < r->reference);      /* asc  */
}

/* storage.c:477-580: gather the needle's buckets, sort by ref, run-length
 * count, rank by (matches desc, weight asc, ref asc), copy the first `limit`. */
int oracle_find(void* h, const char* needle, uint16_t limit, ora_match_t* results) {
  ora_map_t* m = (ora_map_t*)h;
  size_t len = strlen(needle);
  uint16_t* codes = (uint16_t*)malloc((len + 1) * sizeof(uint16_t));
  int n = oracle_tokenise(needle, codes);                          /* :492 */
  size_t nb = 0;
  for (int k = 0; k < n; ++k) nb += m->b[codes[k]].used;           /* :498-503 */
  int out = 0;
  if (nb == 0) { free(codes); return 0; }

  ora_entry_t* all = (ora_entry_t*)malloc(nb * sizeof(ora_entry_t));
  size_t p = 0;
  for (int k = 0; k < n; ++k) {                                    /* :512-519 */
    const ora_bucket_t* b = &m->b[codes[k]];
    if (b->used) memcpy(all + p, b->e, (size_t)b->used * sizeof(ora_entry_t));
    p += b->used;
  }
  /* :523 -- entries of one ref share one weight (put writes the same weight
   * into every bucket), so the order inside a run does not matter. */
  qsort(all, nb, sizeof(ora_entry_t), ora_cmp_entry);

  size_t nm = 0;                                                   /* :527-536 */
  for (size_t i = 0; i < nb; ++i)
    if (i == 0 || all[i].reference != all[i - 1].reference) ++nm;
  ora_match_t* mt = (ora_match_t*)malloc(nm * sizeof(ora_match_t));
  size_t j = 0;                                                    /* :545-563 */
  for (size_t i = 0; i < nb; ++i) {
    if (i == 0 || all[i].reference != all[i - 1].reference) {
      mt[j].reference = all[i].reference;
      mt[j].weight = all[i].weight;        /* weight of the first entry of the run (:548-554) */
      mt[j].matches = 1;
      ++j;
    } else {
      mt[j - 1].matches += 1;
    }
  }
  qsort(mt, nm, sizeof(ora_match_t), ora_cmp_match);               /* :566 */
  out = (int)((limit < nm) ? limit : nm);                          /* :569 */
  if (out) memcpy(results, mt, (size_t)out * sizeof(ora_match_t)); /* :570-573 */
  free(mt); free(all); free(codes);
  return out;
}
