/* A plain C99 client of include/blurrily_storage.h -- what ext/blurrily/map_ext.c does with the
 * reference's storage.h (map_ext.c:39-197), without Ruby.  Built by tests/test_c_client.py with gcc and
 * linked against libblurrily_hip.so.
 *
 *   abi_client <scratch.trigrams> host   put / stats / delete / save / load / close; find must fail with ENODEV
 *   abi_client <scratch.trigrams> gpu    the same, and find answers (spec/blurrily/map_spec.rb:118-210 vectors)
 */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "blurrily_storage.h"

#define CHECK(cond)                                                         \
  do {                                                                      \
    if (!(cond)) {                                                          \
      fprintf(stderr, "%s:%d: %s failed (errno %d)\n", __FILE__, __LINE__, #cond, errno); \
      return 1;                                                             \
    }                                                                       \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const char* path = argv[1];
  const int gpu = strcmp(argv[2], "gpu") == 0;
  trigram_map map = NULL;
  trigram_stat_t stats;
  trigram_match_t rows[10];

  CHECK(blurrily_storage_new(&map) == 0);
  CHECK(blurrily_storage_put(map, "london", 10, 0) == 7);            /* trigrams added */
  CHECK(blurrily_storage_put(map, "londres", 11, 0) == 8);
  CHECK(blurrily_storage_put(map, "paris", 12, 3) == 6);
  CHECK(blurrily_storage_put(map, "anything", 10, 0) == 0);          /* duplicate reference (storage.c:408) */
  CHECK(blurrily_storage_stats(map, &stats) == 0 && stats.references == 3 && stats.trigrams == 21);

  errno = 0;
  {
    const int n = blurrily_storage_find(map, "london", 10, rows);
    if (gpu) {
      CHECK(n == 2);
      CHECK(rows[0].reference == 10 && rows[0].matches == 7 && rows[0].weight == 6);
      CHECK(rows[1].reference == 11 && rows[1].matches == 4 && rows[1].weight == 7);
      CHECK(blurrily_storage_find(map, "paris", 10, rows) == 1 && rows[0].weight == 3);
      CHECK(blurrily_storage_find(map, "", 10, rows) == 0);
    } else {
      CHECK(n == -1 && errno == ENODEV);                               /* no CPU fallback */
    }
  }

  /* tunables go through the ABI (nothing reads the environment): per map, and process-wide with a NULL map */
  {
    long long v = -1;
    CHECK(blurrily_storage_get_option(map, "ws_min_slice", &v) == 0 && v == 1550);
    CHECK(blurrily_storage_set_option(map, "wsweep", 0) == 0);
    CHECK(blurrily_storage_get_option(map, "wsweep", &v) == 0 && v == 0);
    errno = 0;
    CHECK(blurrily_storage_set_option(map, "no_such_option", 1) == -1 && errno == EINVAL);
    CHECK(blurrily_storage_set_option(NULL, "host_threads", 2) == 0);
    CHECK(blurrily_storage_get_option(NULL, "host_threads", &v) == 0 && v == 2);
    CHECK(blurrily_storage_set_option(NULL, "host_threads", 0) == 0);
    if (gpu) CHECK(blurrily_storage_find(map, "london", 10, rows) == 2);      /* same rows whatever the options */
  }

  CHECK(blurrily_storage_delete(map, 11) == 8);
  CHECK(blurrily_storage_save(map, path) == 0);
  CHECK(blurrily_storage_close(&map) == 0 && map == NULL);

  CHECK(blurrily_storage_load(&map, path) == 0);
  CHECK(blurrily_storage_stats(map, &stats) == 0 && stats.references == 2 && stats.trigrams == 13);
  if (gpu) {
    CHECK(blurrily_storage_find(map, "londres", 10, rows) == 1 && rows[0].reference == 10 && rows[0].matches == 4);
  }
  CHECK(blurrily_storage_close(&map) == 0);

  errno = 0;
  CHECK(blurrily_storage_load(&map, "/nonexistent/x.trigrams") < 0 && errno == ENOENT);
  puts("ok");
  return 0;
}
