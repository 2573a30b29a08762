/*
 * oracle_batch.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Batch drivers over blurrily_oracle.c for the full-size parity checks (tests/,
 * bench.py's parity leg): the oracle's find is read-only once every string is
 * put, so a sample of needles is checked on all host cores.  Each element is
 * exactly one oracle_find / oracle_nb_entries / oracle_tokenise call.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct __attribute__((packed)) { uint32_t reference, matches, weight; } ora_match_t;
int      oracle_find(void* h, const char* needle, uint16_t limit, ora_match_t* results);
uint64_t oracle_nb_entries(void* h, const char* needle);
int      oracle_tokenise(const char* input, uint16_t* out);

typedef struct {
  void* h; const char* packed; const uint64_t* offsets; const uint32_t* idx; size_t n;
  uint16_t limit; ora_match_t* rows; uint32_t* counts; uint64_t* nb; uint32_t* ntri;
  size_t next; pthread_mutex_t mu;
} job_t;

static char* needle_cstr(const job_t* j, size_t q, char** buf, size_t* cap) {
  size_t a = (size_t)j->offsets[q], b = (size_t)j->offsets[q + 1];
  if (b - a + 1 > *cap) { *cap = (b - a + 1) * 2; *buf = (char*)realloc(*buf, *cap); }
  memcpy(*buf, j->packed + a, b - a);
  (*buf)[b - a] = 0;                                  /* a needle is a C string (storage.c:480) */
  return *buf;
}

static void* worker(void* arg) {
  job_t* j = (job_t*)arg;
  size_t cap = 256;
  char* buf = (char*)malloc(cap);
  uint16_t* codes = NULL; size_t ccap = 0;
  for (;;) {
    pthread_mutex_lock(&j->mu);
    size_t k = j->next; j->next += 8;
    pthread_mutex_unlock(&j->mu);
    if (k >= j->n) break;
    size_t end = k + 8 < j->n ? k + 8 : j->n;
    for (; k < end; ++k) {
      size_t q = j->idx ? j->idx[k] : k;
      const char* s = needle_cstr(j, q, &buf, &cap);
      if (j->rows) j->counts[k] = (uint32_t)oracle_find(j->h, s, j->limit, j->rows + k * (size_t)j->limit);
      if (j->nb) j->nb[k] = oracle_nb_entries(j->h, s);
      if (j->ntri) {
        size_t len = strlen(s);
        if (len + 1 > ccap) { ccap = (len + 1) * 2; codes = (uint16_t*)realloc(codes, ccap * sizeof(uint16_t)); }
        j->ntri[k] = (uint32_t)oracle_tokenise(s, codes);
      }
    }
  }
  free(buf); free(codes);
  return NULL;
}

/* For k in [0, n): needle q = idx ? idx[k] : k of the packed batch;
 *   rows[k*limit ..] / counts[k] = oracle_find (skipped when rows is NULL),
 *   nb[k]   = oracle_nb_entries (storage.c:498-502; skipped when NULL),
 *   ntri[k] = distinct trigrams of the needle (tokeniser.c:59-119; skipped when NULL).
 * `threads` <= 1 runs inline. */
int oracle_batch(void* h, const char* packed, const uint64_t* offsets, const uint32_t* idx, size_t n,
                 uint16_t limit, ora_match_t* rows, uint32_t* counts, uint64_t* nb, uint32_t* ntri, int threads) {
  job_t j = {h, packed, offsets, idx, n, limit, rows, counts, nb, ntri, 0, PTHREAD_MUTEX_INITIALIZER};
  if (threads <= 1) { worker(&j); return 0; }
  if (threads > 256) threads = 256;
  pthread_t th[256];
  int started = 0;
  for (int i = 0; i < threads; ++i)
    if (pthread_create(&th[started], NULL, worker, &j) == 0) ++started;
  if (!started) worker(&j);
  for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
  return 0;
}
