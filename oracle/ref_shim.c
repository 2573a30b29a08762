/*
 * ref_shim.c -- TEST INFRASTRUCTURE.  Lazy loader for the compiled reference.
 *
 * oracle/_ref/libblurrily_ref.so is the reference's own storage.c + tokeniser.c
 * compiled in place (oracle/Makefile), with NO stand-in for search_tree.c (it
 * needs ruby.h, which this image lacks).  The six blurrily_refs_* symbols are
 * therefore left undefined in that .so.  They are only reached from
 * put / delete / mark / close-after-put, so everything the find path needs
 * (load, find, save, stats, close of a never-put map, the tokeniser) runs as
 * long as the library is bound lazily.  Python's ctypes always adds RTLD_NOW,
 * hence this shim: it dlopen()s the reference with RTLD_LAZY and forwards.
 *
 * Haystacks reach the reference as .trigrams files (written by the product's
 * blurrily_storage_save) through the reference's own blurrily_storage_load.
 */
#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

typedef int (*fn_tok)(const char*, uint16_t*);
typedef int (*fn_load)(void**, const char*);
typedef int (*fn_new)(void**);
typedef int (*fn_close)(void**);
typedef int (*fn_save)(void*, const char*);
typedef int (*fn_find)(void*, const char*, uint16_t, void*);
typedef int (*fn_stats)(void*, void*);

static void* g_lib;
static fn_tok g_tok; static fn_load g_load; static fn_new g_new; static fn_close g_close;
static fn_save g_save; static fn_find g_find; static fn_stats g_stats;

int ref_open(const char* so_path) {
  if (g_lib) return 0;
  g_lib = dlopen(so_path, RTLD_LAZY | RTLD_LOCAL);
  if (!g_lib) { fprintf(stderr, "ref_shim: %s\n", dlerror()); return -1; }
  g_tok   = (fn_tok)  dlsym(g_lib, "blurrily_tokeniser_parse_string");
  g_load  = (fn_load) dlsym(g_lib, "blurrily_storage_load");
  g_new   = (fn_new)  dlsym(g_lib, "blurrily_storage_new");
  g_close = (fn_close)dlsym(g_lib, "blurrily_storage_close");
  g_save  = (fn_save) dlsym(g_lib, "blurrily_storage_save");
  g_find  = (fn_find) dlsym(g_lib, "blurrily_storage_find");
  g_stats = (fn_stats)dlsym(g_lib, "blurrily_storage_stats");
  return (g_tok && g_load && g_new && g_close && g_save && g_find && g_stats) ? 0 : -2;
}

int ref_tokenise(const char* s, uint16_t* out)            { return g_tok(s, out); }
int ref_new(void** map)                                   { return g_new(map); }
int ref_load(void** map, const char* path)                { return g_load(map, path); }
int ref_close(void** map)                                 { return g_close(map); }
int ref_save(void* map, const char* path)                 { return g_save(map, path); }
int ref_find(void* map, const char* needle, uint16_t limit, void* results) {
  return g_find(map, needle, limit, results);
}
int ref_stats(void* map, uint32_t* out2)                  { return g_stats(map, out2); }

/* Loop for bench.py's cpu_baseline leg (kind "reference"): runs `n` finds over
 * NUL-separated needles, KEEPING every needle's rows (rows[i*limit ..], counts[i])
 * so that the caller can compare them with the GPU's rows for the same needles.
 * Returns the summed result count.  Timing is done by the caller. */
long ref_find_many(void* map, const char* packed, const uint32_t* offsets, int n,
                   uint16_t limit, void* rows, uint32_t* counts) {
  long total = 0;
  for (int i = 0; i < n; ++i) {
    int c = g_find(map, packed + offsets[i], limit, (char*)rows + (size_t)i * limit * 12);
    counts[i] = (uint32_t)(c < 0 ? 0 : c);
    total += c;
  }
  return total;
}
