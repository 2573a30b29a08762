/* header_compat.c -- the reference's ext/blurrily/storage.h and include/blurrily_storage.h in ONE translation
 * unit, compiled with -std=c99 -Wall -Wextra -Werror by tests/test_header_compat.py (only where /root/reference
 * exists: the header is never copied).  The nine functions of storage.h:36-117 are declared by both; C accepts
 * the second declaration only if it is compatible with the first -- same return type, same parameter types --
 * so a drift between the two headers is a compile error here.  The types are the reference's own (our header
 * sees __STORAGE_H__ and leaves them alone); their layout is pinned below. */
#include "storage.h"             /* the reference's: -I/root/reference/ext/blurrily */
#include "blurrily_storage.h"    /* this repository's */

#include <stddef.h>

#define STATIC_ASSERT(cond, name) typedef char static_assert_##name[(cond) ? 1 : -1]
STATIC_ASSERT(sizeof(trigram_match_t) == 12, match_is_twelve_packed_bytes);          /* storage.h:18-24 */
STATIC_ASSERT(offsetof(trigram_match_t, reference) == 0, match_reference_first);
STATIC_ASSERT(offsetof(trigram_match_t, matches) == 4, match_matches_second);
STATIC_ASSERT(offsetof(trigram_match_t, weight) == 8, match_weight_third);
STATIC_ASSERT(sizeof(trigram_stat_t) == 8, stat_is_two_words);                       /* storage.h:26-30 */

/* the glue's calls (map_ext.c:30,39,49,65,91,107,123,149,177,196), type-checked against BOTH declarations */
int header_compat_calls(trigram_map* h, const char* path, trigram_match rows, trigram_stat_t* st);
int header_compat_calls(trigram_map* h, const char* path, trigram_match rows, trigram_stat_t* st)
{
  int (*f_new)(trigram_map*) = blurrily_storage_new;
  int (*f_load)(trigram_map*, const char*) = blurrily_storage_load;
  int (*f_close)(trigram_map*) = blurrily_storage_close;
  void (*f_mark)(trigram_map) = blurrily_storage_mark;
  int (*f_save)(trigram_map, const char*) = blurrily_storage_save;
  int (*f_put)(trigram_map, const char*, uint32_t, uint32_t) = blurrily_storage_put;
  int (*f_delete)(trigram_map, uint32_t) = blurrily_storage_delete;
  int (*f_find)(trigram_map, const char*, uint16_t, trigram_match) = blurrily_storage_find;
  int (*f_stats)(trigram_map, trigram_stat_t*) = blurrily_storage_stats;
  /* part 2 names exist beside them */
  int (*f_batch)(trigram_map, const char*, const uint64_t*, size_t, uint16_t, trigram_match, uint32_t*) =
      blurrily_storage_find_batch;
  (void)f_new; (void)f_load; (void)f_close; (void)f_mark; (void)f_save; (void)f_put; (void)f_delete;
  (void)f_find; (void)f_stats; (void)f_batch; (void)h; (void)path; (void)rows; (void)st;
  return 0;
}
