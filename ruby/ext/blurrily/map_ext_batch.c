/*
 * map_ext_batch.c -- batched methods for Blurrily::RawMap over libblurrily_hip.so.
 *
 * Compiled BESIDE the blurrily gem's own glue, ext/blurrily/map_ext.c (mezis/blurrily v1.0.2), by
 * ruby/ext/blurrily/extconf.rb of this repository.  The gem's glue is taken as it is: it defines
 * Blurrily::RawMap with new/load/put/delete/save/find/stats/close (map_ext.c:206-229) over the nine
 * blurrily_storage_* functions of storage.h:36-117, which libblurrily_hip.so exports with the same
 * signatures (include/blurrily_storage.h, part 1).  This file adds what the reference has no
 * counterpart for (include/blurrily_storage.h, part 2):
 *
 *   RawMap#find_batch(needles, limit)       -> [[[ref, matches, weight], ...], ...]   one list per needle
 *   RawMap#find_batch_raw(needles, limit)   -> [lists, non_ascii_flags]   needles BEFORE normalize_string
 *   RawMap#put_many(needles, refs, weights) -> trigrams added
 *   RawMap#sync_device                      -> nil        build the device image now
 *   RawMap#set_option(key, value) / #get_option(key)      tunables, "devices" among them (multi-GPU)
 *
 * Each element of a batch is defined as exactly one RawMap#find (map_ext.c:131-162): limit <= 0 means
 * LIMIT_DEFAULT, rows are [reference, matches, weight].
 *
 * THREADING.  The reference holds the GVL for every call ("Threading"), which
 * is what makes its lock-free map safe: calls on one map are serial.  A batch of half a second should
 * not stop every other Ruby thread, so the two batched finds run WITHOUT the GVL
 * (rb_thread_call_without_gvl) -- and the map, which has no lock of its own (its scratch buffers, its
 * mutation log and its replicas are mutated by a find), is guarded HERE instead: a batch marks its
 * RawMap busy (@blurrily_busy) for as long as the library call runs, and EVERY method that reaches the
 * library on that object -- the batched ones below and the gem's own put / delete / save / find /
 * stats / close, which Init_map_ext re-defines as guarded wrappers around the gem's methods -- raises
 * Blurrily::RawMap::BusyError while it is set.  A concurrent close can therefore not free a map a
 * batch is searching, and a concurrent put cannot race the find's bucket sorts.  Other maps and other
 * Ruby threads go on.
 *
 * extconf.rb compiles the gem's map_ext.c with -DInit_map_ext=Init_map_ext_reference (that source
 * only: this one is compiled without the define), so the gem's initialiser gets that name; the
 * Init_map_ext Ruby calls on `require 'blurrily/map_ext'` is the one at the end of this file, which
 * runs the gem's first.
 *
 * No Ruby toolchain exists in the image this repository is built in: this file is written against the
 * documented C API of Ruby >= 2.0.  tests/test_ruby_glue_syntax.py runs the compiler's front end over
 * it and over the gem's map_ext.c (gcc -fsyntax-only, prototypes-only headers under
 * tests/c/mock_ruby/: no object code is produced, nothing is linked or run).
 */
#include <ruby.h>
#include <ruby/thread.h>
#include <errno.h>
#include <stdint.h>
#include <string.h>

#include "storage.h"            /* the gem's own header: trigram_map, trigram_match_t, the nine functions */
#include "blurrily_storage.h"   /* this repository's: re-declares those nine (compatibly) and adds part 2 */

void Init_map_ext_reference(void);          /* the gem's Init_map_ext under the name extconf.rb gave it */
void Init_map_ext(void);

static VALUE mBlurrily = Qnil, cRawMap = Qnil, eBusyError = Qnil;

/* ---- the per-map guard ------------------------------------------------------------------------- */

static void raise_if_busy(VALUE self)
{
  if (rb_ivar_get(self, rb_intern("@blurrily_busy")) == Qtrue)
    rb_raise(eBusyError, "a batch is in flight on this map");
}

static trigram_map map_of(VALUE self)
{
  trigram_map haystack = NULL;
  if (rb_ivar_get(self, rb_intern("@closed")) == Qtrue)          /* map_ext.c:11-16 */
    rb_raise(rb_const_get(cRawMap, rb_intern("ClosedError")), "Map was freed");
  raise_if_busy(self);
  Data_Get_Struct(self, struct trigram_map_t, haystack);
  return haystack;
}

/* The gem's own methods, guarded: `name` is re-defined to check the busy flag and then call the gem's
 * method, kept under the alias `name`_unguarded (the GVL is held here, so check-then-call is atomic
 * with respect to other Ruby threads). */
static VALUE guarded_call(int argc, VALUE* argv, VALUE self, const char* unguarded)
{
  raise_if_busy(self);
  return rb_funcallv(self, rb_intern(unguarded), argc, argv);
}
#define GUARDED(name_)                                                                   \
  static VALUE guarded_##name_(int argc, VALUE* argv, VALUE self)                        \
  { return guarded_call(argc, argv, self, #name_ "_unguarded"); }
GUARDED(put) GUARDED(delete) GUARDED(save) GUARDED(find) GUARDED(stats) GUARDED(close)
#undef GUARDED

static void guard_method(const char* name, const char* unguarded, VALUE (*fn)(int, VALUE*, VALUE))
{
  rb_define_alias(cRawMap, unguarded, name);
  rb_define_method(cRawMap, name, fn, -1);
}

/* ---- arguments ---------------------------------------------------------------------------------- */

static uint16_t limit_of(VALUE rb_limit)
{
  int limit = NUM2UINT(rb_limit);                                  /* map_ext.c:135,142-146 */
  if (limit <= 0) limit = NUM2UINT(rb_const_get(mBlurrily, rb_intern("LIMIT_DEFAULT")));
  return (uint16_t)limit;                                          /* storage.h:110: uint16_t */
}

/* needles -> one packed buffer + n+1 offsets (what the library's batched entry points take) */
typedef struct {
  char*     packed;
  uint64_t* offsets;
  size_t    n;
} packed_needles;

/* Everything a batched call allocates hangs off ONE structure that an rb_ensure handler frees, so an
 * exception raised by a conversion (StringValue, NUM2UINT) half-way through leaks nothing. */
typedef struct {
  VALUE           self;
  VALUE           rb_needles, rb_limit, rb_refs, rb_weights;
  int             raw;
  VALUE           strings;        /* the needles as Strings (converted copies: kept alive for the packing) */
  trigram_map     map;
  packed_needles  in;
  uint16_t        limit;
  trigram_match   rows;
  uint32_t*       counts;
  uint32_t*       non_ascii;      /* find_batch_raw only */
  uint32_t*       refs;           /* put_many only */
  uint32_t*       weights;
  int             marked_busy;
  int             res, err;
} batch_call;

static void pack_needles(batch_call* c)
{
  long   n, i;
  size_t bytes = 0;
  Check_Type(c->rb_needles, T_ARRAY);                              /* before any RARRAY_ macro touches it */
  n = RARRAY_LEN(c->rb_needles);
  /* the converted strings are collected first: an element that only responds to to_str is converted
   * ONCE, and what is measured is what is copied */
  c->strings = rb_ary_new2(n);
  for (i = 0; i < n; ++i) {
    VALUE s = rb_ary_entry(c->rb_needles, i);
    StringValue(s);
    rb_ary_push(c->strings, s);
    bytes += (size_t)RSTRING_LEN(s);
  }
  c->in.n = (size_t)n;
  c->in.offsets = ALLOC_N(uint64_t, n + 1);
  c->in.packed  = ALLOC_N(char, bytes + 1);
  c->in.offsets[0] = 0;
  for (i = 0; i < n; ++i) {
    VALUE s = rb_ary_entry(c->strings, i);
    memcpy(c->in.packed + c->in.offsets[i], RSTRING_PTR(s), (size_t)RSTRING_LEN(s));
    c->in.offsets[i + 1] = c->in.offsets[i] + (uint64_t)RSTRING_LEN(s);
  }
}

static VALUE batch_call_cleanup(VALUE p)
{
  batch_call* c = (batch_call*)p;
  if (c->marked_busy) rb_ivar_set(c->self, rb_intern("@blurrily_busy"), Qfalse);
  if (c->in.packed)  xfree(c->in.packed);
  if (c->in.offsets) xfree(c->in.offsets);
  if (c->rows)       xfree(c->rows);
  if (c->counts)     xfree(c->counts);
  if (c->non_ascii)  xfree(c->non_ascii);
  if (c->refs)       xfree(c->refs);
  if (c->weights)    xfree(c->weights);
  return Qnil;
}

/* ---- find_batch / find_batch_raw ---------------------------------------------------------------- */

static void* batch_call_run(void* p)
{
  batch_call* c = (batch_call*)p;
  c->res = c->non_ascii
    ? blurrily_storage_find_batch_raw(c->map, c->in.packed, c->in.offsets, c->in.n, c->limit, c->rows, c->counts, c->non_ascii)
    : blurrily_storage_find_batch(c->map, c->in.packed, c->in.offsets, c->in.n, c->limit, c->rows, c->counts);
  c->err = errno;
  return NULL;
}

static VALUE rows_to_ruby(const batch_call* c)
{
  VALUE  out = rb_ary_new2((long)c->in.n);
  size_t i;
  for (i = 0; i < c->in.n; ++i) {
    VALUE    per = rb_ary_new2(c->counts[i]);
    uint32_t k;
    for (k = 0; k < c->counts[i]; ++k) {                           /* map_ext.c:153-160 */
      const trigram_match_t* m = c->rows + i * c->limit + k;
      rb_ary_push(per, rb_ary_new3(3, rb_uint_new(m->reference), rb_uint_new(m->matches), rb_uint_new(m->weight)));
    }
    rb_ary_push(out, per);
  }
  return out;
}

static VALUE find_batch_body(VALUE p)
{
  batch_call* c = (batch_call*)p;
  VALUE       out = Qnil;
  c->map   = map_of(c->self);                                      /* raises when closed or busy */
  c->limit = limit_of(c->rb_limit);
  pack_needles(c);
  c->rows   = ALLOC_N(trigram_match_t, c->in.n * c->limit + 1);
  c->counts = ALLOC_N(uint32_t, c->in.n + 1);
  if (c->raw) c->non_ascii = ALLOC_N(uint32_t, c->in.n + 1);
  /* busy from here until the cleanup: the library call runs without the GVL */
  rb_ivar_set(c->self, rb_intern("@blurrily_busy"), Qtrue);
  c->marked_busy = 1;
  rb_thread_call_without_gvl(batch_call_run, c, RUBY_UBF_IO, NULL);
  if (c->res < 0) { errno = c->err; rb_sys_fail("blurrily_storage_find_batch"); }   /* ENODEV: no GPU, no CPU fallback */
  out = rows_to_ruby(c);
  if (c->raw) {
    size_t i;
    VALUE  flags = rb_ary_new2((long)c->in.n);
    for (i = 0; i < c->in.n; ++i) rb_ary_push(flags, c->non_ascii[i] ? Qtrue : Qfalse);
    out = rb_ary_new3(2, out, flags);
  }
  return out;
}

static VALUE find_batch_common(VALUE self, VALUE rb_needles, VALUE rb_limit, int raw)
{
  batch_call c;
  memset(&c, 0, sizeof c);
  c.self = self; c.rb_needles = rb_needles; c.rb_limit = rb_limit; c.raw = raw;
  c.strings = Qnil; c.rb_refs = Qnil; c.rb_weights = Qnil;
  return rb_ensure(find_batch_body, (VALUE)&c, batch_call_cleanup, (VALUE)&c);
}

static VALUE blurrily_find_batch(VALUE self, VALUE rb_needles, VALUE rb_limit)
{
  return find_batch_common(self, rb_needles, rb_limit, 0);
}

static VALUE blurrily_find_batch_raw(VALUE self, VALUE rb_needles, VALUE rb_limit)
{
  return find_batch_common(self, rb_needles, rb_limit, 1);
}

/* ---- put_many (holds the GVL: host work, as the gem's put) -------------------------------------- */

static VALUE put_many_body(VALUE p)
{
  batch_call* c = (batch_call*)p;
  long        i, added;
  c->map = map_of(c->self);
  Check_Type(c->rb_refs, T_ARRAY);
  if (!NIL_P(c->rb_weights)) Check_Type(c->rb_weights, T_ARRAY);
  pack_needles(c);
  if ((size_t)RARRAY_LEN(c->rb_refs) != c->in.n || (!NIL_P(c->rb_weights) && (size_t)RARRAY_LEN(c->rb_weights) != c->in.n))
    rb_raise(rb_eArgError, "needles, references and weights differ in length");
  c->refs = ALLOC_N(uint32_t, c->in.n + 1);
  if (!NIL_P(c->rb_weights)) c->weights = ALLOC_N(uint32_t, c->in.n + 1);
  for (i = 0; i < (long)c->in.n; ++i) {
    c->refs[i] = NUM2UINT(rb_ary_entry(c->rb_refs, i));            /* (may raise: the cleanup frees) */
    if (c->weights) c->weights[i] = NUM2UINT(rb_ary_entry(c->rb_weights, i));
  }
  added = blurrily_storage_put_many(c->map, c->in.packed, c->in.offsets, c->refs, c->weights, c->in.n);
  if (added < 0) rb_sys_fail("blurrily_storage_put_many");
  return LONG2NUM(added);
}

static VALUE blurrily_put_many(VALUE self, VALUE rb_needles, VALUE rb_refs, VALUE rb_weights)
{
  batch_call c;
  memset(&c, 0, sizeof c);
  c.self = self; c.rb_needles = rb_needles; c.rb_refs = rb_refs; c.rb_weights = rb_weights;
  c.strings = Qnil; c.rb_limit = Qnil;
  return rb_ensure(put_many_body, (VALUE)&c, batch_call_cleanup, (VALUE)&c);
}

/* ---- the small ones ----------------------------------------------------------------------------- */

static VALUE blurrily_sync_device(VALUE self)
{
  if (blurrily_storage_sync_device(map_of(self)) < 0) rb_sys_fail("blurrily_storage_sync_device");
  return Qnil;
}

static VALUE blurrily_set_option(VALUE self, VALUE rb_key, VALUE rb_value)
{
  if (blurrily_storage_set_option(map_of(self), StringValueCStr(rb_key), NUM2LL(rb_value)) < 0)
    rb_sys_fail("blurrily_storage_set_option");
  return rb_value;
}

static VALUE blurrily_get_option(VALUE self, VALUE rb_key)
{
  long long v = 0;
  if (blurrily_storage_get_option(map_of(self), StringValueCStr(rb_key), &v) < 0)
    rb_sys_fail("blurrily_storage_get_option");
  return LL2NUM(v);
}

void Init_map_ext(void)
{
  Init_map_ext_reference();                                        /* Blurrily::RawMap as the gem defines it */
  mBlurrily  = rb_define_module("Blurrily");
  cRawMap    = rb_const_get(mBlurrily, rb_intern("RawMap"));
  eBusyError = rb_define_class_under(cRawMap, "BusyError", rb_eRuntimeError);
  /* the gem's methods that reach the library, behind the busy check (THREADING above) */
  guard_method("put",    "put_unguarded",    guarded_put);
  guard_method("delete", "delete_unguarded", guarded_delete);
  guard_method("save",   "save_unguarded",   guarded_save);
  guard_method("find",   "find_unguarded",   guarded_find);
  guard_method("stats",  "stats_unguarded",  guarded_stats);
  guard_method("close",  "close_unguarded",  guarded_close);
  rb_define_method(cRawMap, "find_batch",     blurrily_find_batch,     2);
  rb_define_method(cRawMap, "find_batch_raw", blurrily_find_batch_raw, 2);
  rb_define_method(cRawMap, "put_many",       blurrily_put_many,       3);
  rb_define_method(cRawMap, "sync_device",    blurrily_sync_device,    0);
  rb_define_method(cRawMap, "set_option",     blurrily_set_option,     2);
  rb_define_method(cRawMap, "get_option",     blurrily_get_option,     1);
}
